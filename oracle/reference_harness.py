"""Import the reference (``/root/reference``) in the BUILD CONTAINER ONLY.

Test infrastructure only (see ``oracle/__init__.py``).  Used by
``tests/golden/make_golden.py`` to generate fixtures; never imported by
``-m gpu`` tests, ``smoke()`` or ``bench.py`` (``/root/reference`` does not
exist on the GPU box).

The reference's flow arithmetic lives in the absent third-party package
``glasflow``.  ``install()`` registers
``oracle.nflows_restated`` under the ``glasflow.*`` module names the
reference imports, plus an empty ``seaborn``, so that the reference's own
*assembly and glue* code — ``nessai.flows.RealNVP`` / ``NeuralSplineFlow``,
``nessai.flowmodel.FlowModel``, ``nessai.proposal.FlowProposal``, the
truncation rules and the reparameterisations — runs unmodified from
``/root/reference/src`` and can be captured as golden vectors.  Nothing is
written under ``/root/reference`` (``sys.dont_write_bytecode``).
"""

import importlib.machinery
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def available():
    return os.path.isdir(os.path.join(REFERENCE_SRC, "nessai"))


def _module(name, **attrs):
    mod = types.ModuleType(name)
    mod.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def install():
    """Make ``import nessai`` work here, with glasflow → oracle restatement."""
    if not available():
        raise RuntimeError("reference tree not present; fixtures cannot be rebuilt")
    sys.dont_write_bytecode = True
    if "glasflow" in sys.modules and getattr(
        sys.modules["glasflow"], "_is_oracle_stub", False
    ):
        return

    import torch

    from .nflows_restated import distributions as D
    from .nflows_restated import nets as N
    from .nflows_restated import transforms as T
    from .nflows_restated import utils as U

    class _Unavailable:
        """Placeholder for classes outside the restated scope."""

        def __init__(self, *args, **kwargs):
            raise NotImplementedError(
                f"{type(self).__name__} is outside the oracle's restated scope"
            )

    def _unavailable(name):
        return type(name, (_Unavailable,), {})

    class BoxUniform(D.Distribution):
        def __init__(self, low, high):
            super().__init__()
            self.low = torch.as_tensor(low)
            self.high = torch.as_tensor(high)

    glasflow = _module("glasflow", __version__="oracle-restatement", _is_oracle_stub=True)
    glasflow.__path__ = []

    gdist = _module(
        "glasflow.distributions",
        MultivariateUniform=_unavailable("MultivariateUniform"),
        ResampledGaussian=type(
            "ResampledGaussian",
            (D.Distribution,),
            {"estimate_normalisation_constant": lambda self, *a, **k: None},
        ),
    )
    glasflow.distributions = gdist

    nflows = _module("glasflow.nflows")
    nflows.__path__ = []
    glasflow.nflows = nflows

    tmod = _module("glasflow.nflows.transforms")
    tmod.__path__ = []
    for name in (
        "Transform",
        "CompositeTransform",
        "Permutation",
        "RandomPermutation",
        "ReversePermutation",
        "LULinear",
        "SVDLinear",
        "BatchNorm",
        "ActNorm",
        "AffineCouplingTransform",
        "AdditiveCouplingTransform",
        "PiecewiseRationalQuadraticCouplingTransform",
        "InverseNotAvailable",
        "InputOutsideDomain",
        "normalization",
    ):
        setattr(tmod, name, getattr(T, name))
    tmod.Logit = _unavailable("Logit")
    nflows.transforms = tmod

    _module("glasflow.nflows.transforms.base", CompositeTransform=T.CompositeTransform, Transform=T.Transform)
    _module("glasflow.nflows.transforms.normalization", BatchNorm=T.BatchNorm, ActNorm=T.ActNorm)
    _module(
        "glasflow.nflows.transforms.permutations",
        RandomPermutation=T.RandomPermutation,
        ReversePermutation=T.ReversePermutation,
        Permutation=T.Permutation,
    )
    _module(
        "glasflow.nflows.transforms.autoregressive",
        MaskedAffineAutoregressiveTransform=_unavailable(
            "MaskedAffineAutoregressiveTransform"
        ),
    )

    dmod = _module(
        "glasflow.nflows.distributions",
        Distribution=D.Distribution,
        StandardNormal=D.StandardNormal,
    )
    dmod.__path__ = []
    nflows.distributions = dmod
    _module("glasflow.nflows.distributions.normal", StandardNormal=D.StandardNormal)
    _module("glasflow.nflows.distributions.uniform", BoxUniform=BoxUniform)

    nn_mod = _module("glasflow.nflows.nn")
    nn_mod.__path__ = []
    nflows.nn = nn_mod
    nets_mod = _module("glasflow.nflows.nn.nets", ResidualNet=N.ResidualNet)
    nn_mod.nets = nets_mod

    umod = _module(
        "glasflow.nflows.utils",
        create_alternating_binary_mask=U.create_alternating_binary_mask,
        torchutils=U.torchutils,
    )
    nflows.utils = umod

    if "seaborn" not in sys.modules:
        try:
            import seaborn  # noqa: F401
        except ImportError:
            _module(
                "seaborn",
                set=lambda *a, **k: None,
                set_style=lambda *a, **k: None,
                plotting_context=lambda *a, **k: {},
                color_palette=lambda *a, **k: [],
            )

    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
