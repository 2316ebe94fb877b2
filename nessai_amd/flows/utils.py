"""Flow factory helpers (counterparts of ``src/nessai/flows/utils.py``)."""

import copy
import logging
import math

import torch

from . import modules as M

logger = logging.getLogger(__name__)


def create_linear_transform(linear_transform, features):
    """permutation | lu  (``flows/utils.py:295-329``; svd has no kernel)."""
    name = linear_transform.lower()
    if name == "permutation":
        return M.RandomPermutation(features=features)
    elif name == "lu":
        return M.CompositeTransform(
            [M.RandomPermutation(features=features), M.LULinear(features, identity_init=True)]
        )
    elif name == "svd":
        raise NotImplementedError("The svd linear transform has no gfx950 kernel")
    raise ValueError(
        f"Unknown linear transform: {linear_transform}. "
        "Choose from: {permutation, lu, svd}."
    )


def create_pre_transform(pre_transform, features, **kwargs):
    """batch_norm  (``flows/utils.py:332-349``; logit has no kernel)."""
    if pre_transform == "batch_norm":
        return M.BatchNorm(features=features, **kwargs)
    elif pre_transform == "logit":
        raise NotImplementedError("The logit pre-transform has no gfx950 kernel")
    raise ValueError(f"Unknown pre-transform: {pre_transform}")


def get_n_neurons(n_neurons=None, n_inputs=None, default=8):
    """``flows/utils.py:112-165``: 'auto'/'double' = 2*n_inputs, 'equal', 'half'."""
    if n_inputs is None:
        n = default if n_neurons is None else n_neurons
    elif n_neurons is None or n_neurons in {"auto", "double"}:
        n = 2 * n_inputs
    elif n_neurons == "equal":
        n = n_inputs
    elif n_neurons == "half":
        n = n_inputs // 2
    else:
        n = n_neurons
    try:
        return int(n)
    except ValueError:
        raise ValueError(
            "Could not get number of neurons. `n_neurons` was set to "
            f"`{n_neurons}` which could not be translated to a valid int."
        )


def get_flow_class(name):
    """'realnvp' | 'frealnvp' | 'nsf' | 'spline'  (``flows/utils.py:168-196``)."""
    from .nsf import NeuralSplineFlow
    from .realnvp import RealNVP

    flows = {
        "realnvp": RealNVP,
        "frealnvp": RealNVP,
        "spline": NeuralSplineFlow,
        "nsf": NeuralSplineFlow,
    }
    name = name.lower()
    if name == "maf":
        raise NotImplementedError(
            "MaskedAutoregressiveFlow is outside the MI355X hot-path scope (SURVEY.md §2)"
        )
    if name not in flows:
        raise ValueError(f"Unknown flow: {name}")
    return flows[name]


def get_activation_function(name):
    """Validate an activation name (the kernels select it by name)."""
    return M.activation_name(name)


def configure_model(config):
    """dict -> flow, with the reference's argument order
    ``FlowClass(n_inputs, n_neurons, n_blocks, n_layers, **rest)``
    (``flows/utils.py:208-246``)."""
    config = copy.deepcopy(config)
    if not isinstance(config["n_inputs"], int):
        raise TypeError("Number of inputs (n_inputs) must be an int")
    if "activation" in config:
        config["activation"] = get_activation_function(config["activation"])
    dist_kwargs = config.pop("distribution_kwargs", None) or {}
    distribution = config.pop("distribution", None)
    if isinstance(distribution, str):
        if distribution.lower() not in ("mvn", "normal"):
            raise NotImplementedError(
                f"Base distribution {distribution!r} has no gfx950 kernel"
            )
        dist = M.StandardNormal([config["n_inputs"]])
        dist.var = float(dist_kwargs.get("var", 1.0))
        config["distribution"] = dist
    elif distribution is not None:
        config["distribution"] = distribution
    FlowClass = config.pop("flow", None)
    ftype = config.pop("ftype", None)
    if FlowClass is None and ftype is None:
        raise RuntimeError("Must specify either 'flow' or 'ftype'.")
    if FlowClass is None:
        FlowClass = get_flow_class(ftype)
    return FlowClass(
        config.pop("n_inputs"),
        config.pop("n_neurons"),
        config.pop("n_blocks"),
        config.pop("n_layers"),
        **config,
    )


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
