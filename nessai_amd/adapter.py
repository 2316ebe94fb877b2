"""Plug ``nessai_amd.FlowProposal`` into the reference sampler as a ``flow_proposal_class``.

The reference accepts a proposal class only if it subclasses its own ``FlowProposal``
(``src/nessai/proposal/utils.py:148``; same rule for the ``nessai.proposals`` entry-point group,
``docs/plugins.rst:11-22``).  ``as_reference_proposal(nessai.proposal.FlowProposal)`` returns such a
subclass whose every operation is carried out by the gfx950 implementation:

    from nessai.flowsampler import FlowSampler
    from nessai.proposal import FlowProposal as ReferenceFlowProposal
    from nessai_amd.adapter import as_reference_proposal

    FlowSampler(model, flow_proposal_class=as_reference_proposal(ReferenceFlowProposal), ...).run()

The subclass never runs the reference constructor (it would build the CPU flow); it holds a
``nessai_amd.FlowProposal`` and forwards the sampler-facing protocol to it - ``initialise``,
``train``, ``draw``, ``populate``, ``resume``, ``update_output``, ``reset_model_weights`` - together
with every attribute the sampler reads or writes (``populated``, ``indices``, ``samples``,
``training_count``, ``population_acceptance``, ``ns_acceptance``, ``r`` ...).  The constructor
carries the signature of ``nessai_amd.FlowProposal`` so that the reference's keyword check
(``proposal/utils.py:14-79``, a walk over ``inspect.signature`` of the MRO) accepts the device
options (``device_seed``, ``prior``, ``likelihood``, ``shard``, ``precision`` ...).
"""

import inspect

from .proposal import FlowProposal


_CLASSES = {}


def _rebuild(reference_base, implementation, impl):
    obj = object.__new__(as_reference_proposal(reference_base, implementation))
    object.__setattr__(obj, "_impl", impl)
    return obj


def as_reference_proposal(reference_base, implementation=FlowProposal):
    """A subclass of ``reference_base`` that delegates to ``implementation`` (see module docstring).
    One class per (base, implementation) pair: repeated calls return the same object."""
    if (reference_base, implementation) in _CLASSES:
        return _CLASSES[(reference_base, implementation)]

    class HipFlowProposal(reference_base):
        _implementation = implementation

        def __init__(self, model, **kwargs):  # noqa: D107 - the reference constructor is NOT run
            object.__setattr__(self, "_impl", implementation(model, **kwargs))

        def __getattribute__(self, name):
            # whatever the implementation has wins - also over class attributes and properties of
            # the reference base (``initialised`` is a property there); the rest resolves normally
            if name.startswith("__") or name in ("_impl", "_implementation"):
                return object.__getattribute__(self, name)
            try:
                return getattr(object.__getattribute__(self, "_impl"), name)
            except AttributeError:
                return object.__getattribute__(self, name)

        def __setattr__(self, name, value):
            setattr(object.__getattribute__(self, "_impl"), name, value)

        def __reduce__(self):
            # the class is made at run time: pickle it as (reference base, implementation state)
            return (_rebuild, (reference_base, implementation, object.__getattribute__(self, "_impl")))

    def forward(name):
        def method(self, *args, **kwargs):
            return getattr(object.__getattribute__(self, "_impl"), name)(*args, **kwargs)

        method.__name__ = name
        method.__doc__ = getattr(implementation, name).__doc__
        return method

    # everything the implementation defines wins over what the reference base class defines
    for name, member in inspect.getmembers(implementation):
        if name.startswith("__") or name in ("_implementation",):
            continue
        if isinstance(inspect.getattr_static(implementation, name), property):
            setattr(HipFlowProposal, name, property(
                (lambda n: lambda self: getattr(object.__getattribute__(self, "_impl"), n))(name),
                (lambda n: lambda self, v: setattr(object.__getattribute__(self, "_impl"), n, v))(name),
            ))
        elif callable(member):
            setattr(HipFlowProposal, name, forward(name))
    HipFlowProposal.__init__.__signature__ = inspect.signature(implementation.__init__)
    HipFlowProposal.__name__ = HipFlowProposal.__qualname__ = "HipFlowProposal"
    _CLASSES[(reference_base, implementation)] = HipFlowProposal
    return HipFlowProposal
