"""``nessai_amd.adapter``: the subclass the reference's ``flow_proposal_class`` check demands
(``src/nessai/proposal/utils.py:148``) delegates everything to ``nessai_amd.FlowProposal``."""
import inspect
import os
import pickle

import numpy as np
import pytest

from nessai_amd.adapter import as_reference_proposal
from nessai_amd.model import RosenbrockModel
from nessai_amd.proposal import FlowProposal


class StandInBase:
    """What matters of the reference base for the check: a class with its own constructor, state
    and methods of the same names."""

    def __init__(self, model, poolsize=None, **kwargs):
        raise AssertionError("the reference constructor must not run")

    populated = "class attribute of the base"

    def populate(self, *a, **k):
        raise AssertionError("the reference populate must not run")

    def helper_only_in_base(self):
        return "base"


def test_subclass_delegates_to_the_device_implementation(tmp_path):
    cls = as_reference_proposal(StandInBase)
    assert issubclass(cls, StandInBase) and cls.__name__ == "HipFlowProposal"
    params = inspect.signature(cls.__init__).parameters
    assert {"device_seed", "shard", "precision", "truncation_method", "poolsize"} <= set(params)
    prop = cls(RosenbrockModel(4), poolsize=50, rng=np.random.default_rng(1), output=str(tmp_path),
               radius_mode="fixed", fixed_radius=2.0)
    impl = object.__getattribute__(prop, "_impl")
    assert isinstance(impl, FlowProposal)
    assert prop.populated is False and prop.poolsize == 50 and prop.fixed_radius == 2.0
    prop.ns_acceptance = 0.25  # the sampler writes attributes (nestedsampler.py:854)
    prop.populated = True
    assert impl.ns_acceptance == 0.25 and impl.populated is True
    assert prop.helper_only_in_base() == "base"  # untouched names still resolve on the base
    assert prop.populate.__doc__ == FlowProposal.populate.__doc__
    with pytest.raises(RuntimeError):  # delegated: the implementation's own error, not the base's
        prop.populate(None, n_samples=10)
    again = pickle.loads(pickle.dumps(prop))
    assert isinstance(again, cls) and again.ns_acceptance == 0.25 and again.model is None


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/nessai"), reason="reference tree not present")
def test_reference_accepts_the_subclass():
    """In the build container: the reference's own class check and keyword check pass."""
    import sys

    from oracle import reference_harness

    reference_harness.install()
    sys.path.insert(0, reference_harness.REFERENCE_SRC)
    from nessai.proposal import FlowProposal as ReferenceFlowProposal
    from nessai.proposal.utils import check_proposal_kwargs, get_flow_proposal_class

    cls = as_reference_proposal(ReferenceFlowProposal)
    assert get_flow_proposal_class(cls) is cls
    kwargs = dict(poolsize=100, drawsize=1000, device_seed=3, precision="f32", volume_fraction=0.9)
    assert check_proposal_kwargs(cls, kwargs) == kwargs
    with pytest.raises(RuntimeError):
        check_proposal_kwargs(cls, dict(no_such_option=1))
    prop = cls(RosenbrockModel(3), **kwargs)
    assert prop.drawsize == 1000 and prop.volume_fraction == 0.9 and isinstance(prop, ReferenceFlowProposal)
