"""Discrete parameters on the device path (runs on a real MI355X): ``dequantise`` / ``dequantise-logit``
(``src/nessai/reparameterisations/discrete.py:9-78``, registry ``reparameterisations/__init__.py:180-189``) - uniform noise
on the way to x' (host, the proposal's generator), rescale-to-bounds on [min, max + 1], and ``floor`` in the draw
kernel's epilogue on the way back (kinds ``NSA_RP_DEQUANTISE*``).

Goldens: ``tests/golden/dequantise_goldens.npz`` - the reference's own ``FlowProposal`` (``make_dequantise_goldens.py``);
the float64 glue on the host is bit-exact (``tests/test_reparam.py``), the fused kernel is held to float32 here: a whole
number can differ where the float32 flow output puts the continuous value within rounding of one.
"""
import os

import numpy as np
import pytest
import torch

from nessai_amd.livepoint import numpy_array_to_live_points
from nessai_amd.model import UniformBoxModel
from nessai_amd.proposal import FlowProposal

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dequantise_goldens.npz"))
NAMES = ["a", "k", "c", "m"]
BOUNDS = {"a": [-3.0, 7.0], "k": [0.0, 7.0], "c": [0.5, 2.5], "m": [2.0, 5.0]}
CONFIG = {"a": "rescaletobounds", "k": "dequantise", "c": "z-score", "m": "dequantise-logit"}


class DiscreteBoxModel(UniformBoxModel):
    def __init__(self):
        self.names = list(NAMES)
        self.bounds = {n: list(BOUNDS[n]) for n in NAMES}

    def new_point(self, N=1, rng=None):
        x = super().new_point(N, rng=rng)
        for p in ("k", "m"):
            x[p] = np.floor(x[p])
        return x

    def log_likelihood(self, x):
        return -0.5 * ((x["a"] - 1.0) ** 2 + (x["k"] - 3.0) ** 2 + (x["c"] - 1.5) ** 2 + (x["m"] - 4.0) ** 2)


def stack(x, names):
    return np.stack([np.asarray(x[n], dtype=np.float64) for n in names], axis=1)


@pytest.mark.parametrize("prior", ["device", "host"])
def test_dequantise_proposal_matches_the_reference(prior):
    prop = FlowProposal(DiscreteBoxModel(), rng=np.random.default_rng(4242), poolsize=50, drawsize=400, device_seed=1,
                        flow_config=dict(ftype="realnvp", n_blocks=2, n_neurons=8, n_layers=1,
                                         batch_norm_between_layers=False, linear_transform="permutation"),
                        reparameterisations=CONFIG, prior=prior)
    prop.initialise()
    g = GOLD
    state = {k[len("state/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
    prop.flow.model.load_state_dict(state)
    prop.flow.model.eval()
    live = numpy_array_to_live_points(g["live"], NAMES)
    prop.check_state(live)
    assert prop.prime_parameters == [str(p) for p in g["prime_parameters"]] and prop.dims == 4
    prop.rng.bit_generator.state = np.random.default_rng(4242).bit_generator.state
    x_prime, log_j = prop.rescale(live)
    assert np.array_equal(stack(x_prime, prop.prime_parameters), g["x_prime"])
    assert np.array_equal(log_j, g["log_j"])
    x, log_q, z_kept = prop.backward_pass(g["bp_z"], return_z=True)
    assert np.array_equal(z_kept, g["bp_z_kept"])  # the same rows survive the bounds
    names = [str(n) for n in g["bp_names"]]
    got, want = stack(x, names), g["bp_x"]
    np.testing.assert_allclose(got[:, [0, 2]], want[:, [0, 2]], rtol=0, atol=3e-5)
    for col in (1, 3):  # whole numbers, the same ones (a float32 flow output next to a step may land on its other side)
        assert np.array_equal(got[:, col], np.floor(got[:, col]))
        assert np.mean(got[:, col] == want[:, col]) >= 0.99 and np.abs(got[:, col] - want[:, col]).max() <= 1.0
    np.testing.assert_allclose(log_q, g["bp_log_q"], rtol=1e-5, atol=5e-5)
    with np.errstate(divide="ignore"):
        log_w = prop.compute_weights(x, log_q)
    np.testing.assert_allclose(log_w, g["bp_log_w"], rtol=1e-5, atol=5e-5)


def test_populate_with_discrete_parameters(tmp_path):
    """Train on live points of a model with two discrete parameters, populate: every loop of ``populate`` (lean
    passes + replay = full passes bit for bit), the pool holds whole numbers inside their bounds - all of them - with
    the model's prior, and the flow has learnt which ones the likelihood favours."""
    model = DiscreteBoxModel()

    def run(lean):
        torch.manual_seed(3)
        prop = FlowProposal(model, rng=np.random.default_rng(5), poolsize=400, drawsize=30_000, device_seed=11,
                            flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2),
                            training_config=dict(max_epochs=40, patience=40, batch_size=500),
                            reparameterisations=CONFIG, lean=lean, output=str(tmp_path / ("lean" if lean else "full")))
        prop.initialise()
        live = model.new_point(4000, rng=np.random.default_rng(1))
        live["logL"] = model.log_likelihood(live)
        live = np.sort(live, order="logL")[-1000:]  # the 1000 best: k near 3, m near 4
        prop.train(live)
        prop.populate(live[0], n_samples=400, max_samples=10**8)
        return prop

    a, b = run(True), run(False)
    assert a.n_proposed == b.n_proposed and np.array_equal(stack(a.samples, NAMES), stack(b.samples, NAMES))
    s = a.samples
    assert len(s) == 400 and model.in_bounds(s).all()
    assert np.array_equal(s["k"], np.floor(s["k"])) and np.array_equal(s["m"], np.floor(s["m"]))
    assert set(np.unique(s["m"])) <= {2.0, 3.0, 4.0, 5.0} and set(np.unique(s["k"])) <= set(np.arange(8.0))
    assert np.array_equal(s["logP"], model.log_prior(s))
    x, _ = a.backward_pass(np.random.default_rng(4).normal(size=(4000, a.dims)))
    assert abs(np.mean(x["k"]) - 3.0) < 1.0 and abs(np.mean(x["m"]) - 4.0) < 1.0
