"""Flow factory helpers: the configuration schema of ``src/nessai/flows/utils.py`` (names, argument meaning
and error types kept - it is the B1 surface), restated as tables over what the gfx950 kernels implement."""

import copy
import logging

import torch

from . import modules as M

logger = logging.getLogger(__name__)

# What the kernels have, and what the reference knows by name but the kernels do not (refused loudly).
_NO_KERNEL = "has no gfx950 kernel"
_LINEAR = {
    "permutation": lambda d: M.RandomPermutation(features=d),
    "lu": lambda d: M.CompositeTransform([M.RandomPermutation(features=d), M.LULinear(d, identity_init=True)]),
    "svd": lambda d: M.CompositeTransform([M.RandomPermutation(features=d),
                                           M.SVDLinear(d, num_householder=10, identity_init=True)]),
}
_LINEAR_REFUSED = ()
_PRE = {"batch_norm": lambda d, **kw: M.BatchNorm(features=d, **kw)}
_PRE_REFUSED = ("logit",)
_WIDTH_RULES = {"auto": (2, 1), "double": (2, 1), "equal": (1, 1), "half": (1, 2)}  # n_inputs * a // b
_FLOW_NAMES = {"realnvp": "RealNVP", "frealnvp": "RealNVP", "spline": "NeuralSplineFlow", "nsf": "NeuralSplineFlow"}
_BASE_DISTRIBUTIONS = ("mvn", "normal")
_POSITIONAL = ("n_inputs", "n_neurons", "n_blocks", "n_layers")  # FlowClass(*these, **rest)


def create_linear_transform(linear_transform, features):
    """The layer between couplings, by name (``flows/utils.py:295-329``)."""
    key = linear_transform.lower()
    if key in _LINEAR_REFUSED:
        raise NotImplementedError(f"The {key} linear transform {_NO_KERNEL}")
    if key not in _LINEAR:
        raise ValueError(f"Unknown linear transform: {linear_transform}. Choose from: {{permutation, lu, svd}}.")
    return _LINEAR[key](features)


def create_pre_transform(pre_transform, features, **kwargs):
    """The layer in front of the first coupling, by name (``flows/utils.py:332-349``)."""
    if pre_transform in _PRE_REFUSED:
        raise NotImplementedError(f"The {pre_transform} pre-transform {_NO_KERNEL}")
    if pre_transform not in _PRE:
        raise ValueError(f"Unknown pre-transform: {pre_transform}")
    return _PRE[pre_transform](features, **kwargs)


def get_n_neurons(n_neurons=None, n_inputs=None, default=8):
    """Width of the conditioner (``flows/utils.py:112-165``): a number, or a rule on the number of inputs
    ('auto' / 'double', also the meaning of ``None`` once the inputs are known; 'equal'; 'half')."""
    width = n_neurons
    if n_inputs is None:
        width = default if n_neurons is None else n_neurons
    elif n_neurons is None or n_neurons in _WIDTH_RULES:
        a, b = _WIDTH_RULES["auto" if n_neurons is None else n_neurons]
        width = n_inputs * a // b
    try:
        return int(width)
    except ValueError:
        raise ValueError(
            f"Could not get number of neurons. `n_neurons` was set to `{n_neurons}` which could not be "
            "translated to a valid int."
        )


def get_flow_class(name):
    """Flow class of an ``ftype`` (``flows/utils.py:168-196``)."""
    from . import nsf, realnvp

    key = name.lower()
    if key == "maf":
        raise NotImplementedError("MaskedAutoregressiveFlow is outside the MI355X hot-path scope (SURVEY.md §2)")
    if key not in _FLOW_NAMES:
        raise ValueError(f"Unknown flow: {name}")
    cls = _FLOW_NAMES[key]
    return getattr(realnvp if cls == "RealNVP" else nsf, cls)


def get_activation_function(name):
    """Validate an activation name (the kernels select it by name)."""
    return M.activation_name(name)


def _base_distribution(spec, n_inputs, kwargs):
    if not isinstance(spec, str):
        return spec
    if spec.lower() not in _BASE_DISTRIBUTIONS:
        raise NotImplementedError(f"Base distribution {spec!r} {_NO_KERNEL}")
    dist = M.StandardNormal([n_inputs])
    dist.var = float(kwargs.get("var", 1.0))
    return dist


def configure_model(config):
    """A flow from the ``flow_config`` dictionary (``flows/utils.py:208-246``): class from ``flow`` or ``ftype``,
    the four sizes positionally, everything else by keyword."""
    cfg = copy.deepcopy(config)
    if not isinstance(cfg["n_inputs"], int):
        raise TypeError("Number of inputs (n_inputs) must be an int")
    flow_class, ftype = cfg.pop("flow", None), cfg.pop("ftype", None)
    if flow_class is None:
        if ftype is None:
            raise RuntimeError("Must specify either 'flow' or 'ftype'.")
        flow_class = get_flow_class(ftype)
    if "activation" in cfg:
        cfg["activation"] = get_activation_function(cfg["activation"])
    dist = _base_distribution(cfg.pop("distribution", None), cfg["n_inputs"], cfg.pop("distribution_kwargs", None) or {})
    if dist is not None:
        cfg["distribution"] = dist
    sizes = [cfg.pop(k) for k in _POSITIONAL]
    return flow_class(*sizes, **cfg)


def reset_weights(module):
    """Re-initialise a module in place (``flows/utils.py:249-274``): whatever has a
    ``reset_parameters`` (the Linear layers; batch norm, whose constants the reference writes by
    hand).  As in the reference, LU-linear layers and act norm have none - they are re-initialised by
    ``reset_permutations`` / never."""
    if hasattr(module, "reset_parameters"):
        module.reset_parameters()


def reset_permutations(module):
    """Re-draw permutations / re-initialise LU layers (``flows/utils.py:277-292``)."""
    if isinstance(module, M.LULinear):
        module._initialize(identity_init=True)
    elif isinstance(module, M.RandomPermutation):
        module._permutation.copy_(torch.randperm(len(module._permutation)))
