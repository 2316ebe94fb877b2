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
    # the module-level export the `nessai.proposals` entry point names
    from nessai_amd.plugin import MI355XFlowProposal

    assert MI355XFlowProposal is cls and get_flow_proposal_class(MI355XFlowProposal) is cls


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/nessai"), reason="reference tree not present")
def test_reference_reset_visitors_reach_the_hip_flow():
    """``FlowModel.reset_model`` of the reference applies ITS ``reset_weights`` / ``reset_permutations``
    to the flow (``flowmodel/base.py:748-772``, ``flows/utils.py:249-292``); they select modules with
    ``isinstance(..., glasflow class)``.  Given to ``HipFlow.apply`` they are swapped for the counterparts of
    this package, so batch norm returns to its constants, Linear layers are re-drawn, LU-linear layers go
    back to the identity and permutations are re-drawn - through the reference's own entry points."""
    import sys

    import torch

    from oracle import reference_harness

    reference_harness.install()
    sys.path.insert(0, reference_harness.REFERENCE_SRC)
    from nessai.flows.utils import reset_permutations, reset_weights

    from nessai_amd.flows import RealNVP
    from nessai_amd.flows import modules as M

    torch.manual_seed(4)
    flow = RealNVP(6, 16, 2, 2)  # the reference's defaults: LU-linear + batch norm between the couplings
    bn = [m for m in flow.modules() if isinstance(m, M.BatchNorm)]
    lu = [m for m in flow.modules() if isinstance(m, M.LULinear)]
    perms = [m for m in flow.modules() if isinstance(m, M.RandomPermutation)]
    linear = [m for m in flow.modules() if isinstance(m, torch.nn.Linear)]
    assert bn and lu and perms and linear
    with torch.no_grad():
        for m in bn:
            m.unconstrained_weight.add_(0.3)
            m.bias.add_(1.0)
            m.running_mean.add_(2.0)
            m.running_var.mul_(3.0)
        for m in lu:
            for p in m.parameters():
                p.add_(0.2)
    before = [m.weight.detach().clone() for m in linear]
    lu_before = [torch.cat([p.detach().reshape(-1) for p in m.parameters()]) for m in lu]
    flow.apply(reset_weights)
    for m in bn:
        constant = np.log(np.exp(1 - m.eps) - 1)
        assert torch.allclose(m.unconstrained_weight, torch.full_like(m.unconstrained_weight, constant))
        assert float(m.bias.abs().max()) == 0.0 and float(m.running_mean.abs().max()) == 0.0
        assert torch.equal(m.running_var, torch.ones_like(m.running_var))
    assert all(not torch.equal(a, m.weight) for a, m in zip(before, linear))
    # LU-linear layers have no reset_parameters in nflows either: untouched by reset_weights ...
    assert all(torch.equal(a, torch.cat([p.detach().reshape(-1) for p in m.parameters()])) for a, m in zip(lu_before, lu))
    old_perms = [m._permutation.clone() for m in perms]
    torch.manual_seed(9)
    expected = [torch.randperm(len(m._permutation)) for m in perms]
    torch.manual_seed(9)
    flow.apply(reset_permutations)
    # ... and re-initialised to the identity by reset_permutations, which also re-draws the permutations
    for m in lu:
        w = m.weight() if callable(getattr(m, "weight", None)) else None
        if w is not None:
            assert torch.allclose(w, torch.eye(w.shape[0]), atol=1e-6)
    assert all(torch.equal(m._permutation.cpu(), e) for m, e in zip(perms, expected))
    assert any(not torch.equal(a, m._permutation) for a, m in zip(old_perms, perms))
    # any other visitor goes through untouched
    seen = []
    flow.apply(lambda m: seen.append(type(m).__name__))
    assert "BatchNorm" in seen and "LULinear" in seen


def _egg_info(tmp_path):
    """The package's metadata as an installer would write it (no install: `setup.py egg_info` into a scratch
    directory), read back through importlib.metadata."""
    import importlib.metadata
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, "setup.py", "-q", "egg_info", "--egg-base", str(tmp_path)], cwd=root, check=True,
                   env=dict(os.environ, NESSAI_AMD_SKIP_HIP_BUILD="1"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    info = [p for p in os.listdir(tmp_path) if p.endswith(".egg-info")]
    assert len(info) == 1
    return importlib.metadata.Distribution.at(os.path.join(tmp_path, info[0]))


def test_the_package_registers_the_proposal_under_the_reference_s_plugin_group(tmp_path):
    """pyproject.toml: `mi355x = nessai_amd.plugin:MI355XFlowProposal` in the `nessai.proposals` group - the name
    the reference's `get_flow_proposal_class("mi355x")` looks up (src/nessai/proposal/utils.py:97-107)."""
    dist = _egg_info(tmp_path)
    assert dist.metadata["Name"] == "nessai-amd"
    eps = [e for e in dist.entry_points if e.group == "nessai.proposals"]
    assert [(e.name, e.value) for e in eps] == [("mi355x", "nessai_amd.plugin:MI355XFlowProposal")]


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/nessai"), reason="reference tree not present")
def test_the_reference_resolves_the_registered_name(tmp_path, monkeypatch):
    """With the metadata on the path, the reference's own lookup turns `flow_proposal_class="mi355x"` into the
    delegating subclass (the reference reads the group through `nessai.utils.entry_points.get_entry_points`)."""
    import sys

    from oracle import reference_harness

    reference_harness.install()
    sys.path.insert(0, reference_harness.REFERENCE_SRC)
    dist = _egg_info(tmp_path)
    monkeypatch.syspath_prepend(str(tmp_path))
    import importlib
    import importlib.metadata

    importlib.invalidate_caches()
    found = [e for e in importlib.metadata.entry_points(group="nessai.proposals") if e.name == "mi355x"]
    assert found and found[0].value == "nessai_amd.plugin:MI355XFlowProposal", dist.entry_points
    import nessai.proposal.utils as ref_utils
    from nessai.proposal import FlowProposal as ReferenceFlowProposal

    assert "mi355x" in ref_utils.available_external_flow_proposal_classes()
    cls = ref_utils.get_flow_proposal_class("mi355x")
    from nessai_amd.plugin import MI355XFlowProposal

    assert cls is MI355XFlowProposal and issubclass(cls, ReferenceFlowProposal)
