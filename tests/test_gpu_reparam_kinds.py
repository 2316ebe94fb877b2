"""Every reparameterisation kind of the reference's registry mixed at random in one proposal (runs on a real MI355X):
``scripts/fuzz_reparam_kinds.py`` - the fused draw kernel's ``backward_pass`` on injected latents against the float64
host table (pinned bit for bit to the reference's ``FlowProposal`` by the golden tests of ``tests/test_reparam.py``)
applied to the flow's own x': kept rows, every column of x with its auxiliary radii, log q with the row-dependent
Jacobian terms - random group orders, reversed application, angle pairs split by other parameters
(``src/nessai/reparameterisations/__init__.py:28-202``, ``combined.py:58-72``)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_mixes_of_every_registry_name(seed):
    import fuzz_reparam_kinds as F

    summary, bad = F.run(25, seed, verbose=False)
    assert not bad, bad[:2]
    assert summary["worst_x_over_tol"] <= 1.0 and summary["worst_log_q_rel"] <= 1e-5, summary
