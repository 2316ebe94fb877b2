"""The device path against golden vectors captured from the reference's own code.

Same fixtures as ``tests/test_golden.py`` (``tests/golden/reference_goldens.npz``):
outputs of the reference's ``RealNVP`` / ``NeuralSplineFlow`` / ``FlowProposal``
on fixed weights and inputs.  Tolerances: float32 flow outputs to float32
round-off (different summation order on the MFMA), float64 glue exact given the
flow outputs, accepted sets identical.
"""
import os

import numpy as np
import pytest
import torch

from helpers import DEV, rel_err
from nessai_amd.flows import NeuralSplineFlow, RealNVP
from nessai_amd.livepoint import numpy_array_to_live_points
from nessai_amd.model import RosenbrockModel
from nessai_amd.proposal import FlowProposal

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_goldens.npz"))

FLOW_CASES = {
    "nsf_d4": (NeuralSplineFlow, (4, 8, 2, 2), {}),
    "nsf_d6_k10": (NeuralSplineFlow, (6, 16, 3, 1), dict(num_bins=10)),
    "rnvp_d4": (RealNVP, (4, 8, 2, 2), {}),
    "rnvp_d5_mlp": (RealNVP, (5, 16, 3, 2), dict(net="mlp", linear_transform="permutation")),
}


def state_of(prefix):
    pre = prefix + "/state/"
    return {k[len(pre):]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith(pre)}


@pytest.mark.parametrize("name", sorted(FLOW_CASES))
def test_flows_match_reference_outputs(name):
    cls, args, kw = FLOW_CASES[name]
    flow = cls(*args, **kw)
    flow.load_state_dict(state_of(f"flows/{name}"))
    flow.to(DEV)
    flow.eval()
    x = torch.from_numpy(GOLD[f"flows/{name}/x"]).to(DEV)
    z, ld = flow.forward(x)
    xi, ldi = flow.inverse(x)
    lp = flow.log_prob(x)
    assert rel_err(z.cpu(), GOLD[f"flows/{name}/z"]) < 1e-4
    assert rel_err(ld.cpu(), GOLD[f"flows/{name}/logdet"]) < 5e-4
    assert rel_err(xi.cpu(), GOLD[f"flows/{name}/x_inv"]) < 2e-4
    assert rel_err(ldi.cpu(), GOLD[f"flows/{name}/logdet_inv"]) < 5e-4
    assert rel_err(lp.cpu(), GOLD[f"flows/{name}/log_prob"]) < 1e-5  # north_star tolerance


def make_proposal(accumulate=False):
    dims = 4
    model = RosenbrockModel(dims)
    prop = FlowProposal(model, rng=np.random.default_rng(0), poolsize=100, drawsize=1000,
                        flow_config=dict(ftype="nsf", n_blocks=2, n_neurons=8, n_layers=2),
                        accumulate_weights=accumulate, device_seed=1)
    prop.initialise()
    prop.flow.model.load_state_dict(state_of("proposal"))
    prop.flow.model.eval()
    live = numpy_array_to_live_points(GOLD["proposal/live_points"], model.names)
    prop.check_state(live)
    return prop, live


def test_rescaling_matches_reference():
    prop, live = make_proposal()
    x_prime, log_j = prop.rescale(live)
    assert np.array_equal(np.stack([x_prime[p] for p in prop.prime_parameters], axis=1), GOLD["proposal/rescaled"])
    assert np.array_equal(log_j, GOLD["proposal/rescale_log_j"])
    x_back, log_j_inv = prop.inverse_rescale(x_prime)
    assert np.array_equal(np.stack([x_back[p] for p in prop.names], axis=1), GOLD["proposal/inverse_rescaled"])
    assert np.array_equal(log_j_inv, GOLD["proposal/inverse_rescale_log_j"])


def test_forward_pass_matches_reference():
    prop, live = make_proposal()
    z, lp = prop.forward_pass(live)
    np.testing.assert_allclose(z, GOLD["proposal/forward_z"], rtol=0, atol=5e-5)
    assert rel_err(lp, GOLD["proposal/forward_log_prob"]) < 1e-5


def test_backward_pass_matches_reference():
    prop, _ = make_proposal()
    x, log_q, z = prop.backward_pass(GOLD["proposal/bp_z"], return_z=True)
    assert np.array_equal(z, GOLD["proposal/bp_z_kept"])
    np.testing.assert_allclose(np.stack([x[n] for n in prop.names], axis=1), GOLD["proposal/bp_x"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(log_q, GOLD["proposal/bp_log_q"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("tag", ["std", "acc"])
def test_populate_matches_reference(tag):
    prop, _ = make_proposal(tag == "acc")
    zs, us = GOLD[f"populate_{tag}/z"], GOLD[f"populate_{tag}/u"]
    prop.populate(None, n_samples=100, r=float(GOLD["proposal/radius"]), injected=iter(zip(zs, us)))
    want = GOLD[f"populate_{tag}/x"]
    got = np.stack([prop.x[n] for n in prop.names], axis=1)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)
    np.testing.assert_allclose(prop.samples["logP"], GOLD[f"populate_{tag}/logP"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(prop.samples["logL"], GOLD[f"populate_{tag}/logL"], rtol=1e-4)
    assert abs(prop.population_acceptance - float(GOLD[f"populate_{tag}/population_acceptance"])) < 1e-12
    assert prop.n_proposed == 1000 * int(GOLD[f"populate_{tag}/n_passes"])
