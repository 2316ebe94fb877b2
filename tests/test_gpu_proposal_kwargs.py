"""The rest of ``BaseFlowProposal``'s keyword surface on the device path (runs on a real MI355X):
``map_to_unit_hypercube``, ``reverse_reparameterisations``, ``use_default_reparameterisations``, ``save_training_data``
(``src/nessai/proposal/flowproposal/base.py:116-121, 140-158, 277-292, 589, 602-603, 744-797, 887-900, 1053-1128``).

Goldens: ``tests/golden/proposal_kwargs_goldens.npz`` - outputs of the reference's own ``FlowProposal`` built with these
arguments (``tests/golden/make_proposal_kwargs_goldens.py``).  The float64 glue is bit-exact; the flow is float32.
"""
import json
import os

import numpy as np
import pytest
import torch

from nessai_amd.livepoint import numpy_array_to_live_points
from nessai_amd.model import GaussianModel, UniformBoxModel
from nessai_amd.proposal import FlowProposal

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "proposal_kwargs_goldens.npz"))
NAMES = ["a", "b", "c", "d", "e"]
BOUNDS = {"a": [-3.0, 7.0], "b": [0.5, 2.5], "c": [-1.0, 1.0], "d": [10.0, 20.0], "e": [-4.0, 4.0]}
CONFIG = {"a": "rescaletobounds", "b": "logit", "c": "z-score", "d": "rescaletobounds"}


class BoxModel(UniformBoxModel):
    """The generator's model: a uniform box with its own bounds per parameter, Gaussian likelihood."""

    def __init__(self):
        self.names = list(NAMES)
        self.bounds = {n: list(BOUNDS[n]) for n in NAMES}

    def log_likelihood(self, x):
        return -0.5 * np.sum(self.unstructured_view(x) ** 2, axis=-1)


def stack(x, names):
    return np.stack([np.asarray(x[n], dtype=np.float64) for n in names], axis=1)


def make(tag, **extra):
    kwargs = dict(map_to_unit_hypercube=tag.startswith("cube"), reverse_reparameterisations=tag.endswith("reversed"))
    kwargs.update(extra)
    prop = FlowProposal(BoxModel(), rng=np.random.default_rng(0), poolsize=50, drawsize=400, device_seed=1,
                        flow_config=dict(ftype="realnvp", n_blocks=2, n_neurons=8, n_layers=1,
                                         batch_norm_between_layers=False, linear_transform="permutation"),
                        reparameterisations=CONFIG, **kwargs)
    prop.initialise()
    if f"{tag}/state/_transform._transforms.0._permutation" in GOLD.files or any(
            k.startswith(f"{tag}/state/") for k in GOLD.files):
        state = {k[len(f"{tag}/state/"):]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith(f"{tag}/state/")}
        prop.flow.model.load_state_dict(state)
        prop.flow.model.eval()
    live = numpy_array_to_live_points(GOLD["live"], NAMES)
    prop.check_state(live)
    return prop, live


@pytest.mark.parametrize("tag", ["plain", "reversed", "cube", "cube_reversed"])
def test_rescale_and_inverse_rescale_match_the_reference_proposal(tag):
    prop, live = make(tag)
    x_prime, log_j = prop.rescale(live)
    primes = [str(p) for p in GOLD[f"{tag}/prime_parameters"]]
    assert np.array_equal(stack(x_prime, primes), GOLD[f"{tag}/x_prime"])
    assert np.array_equal(log_j, GOLD[f"{tag}/log_j"])
    xp = numpy_array_to_live_points(GOLD["xp_in"], [n + "_prime" for n in NAMES])
    x, log_j_inv = prop.inverse_rescale(xp)
    assert np.array_equal(stack(x, NAMES), GOLD[f"{tag}/x_back"], equal_nan=True)
    assert np.array_equal(log_j_inv, GOLD[f"{tag}/log_j_inv"], equal_nan=True)
    if tag.startswith("cube"):
        assert np.array_equal(np.array([prop.prior_bounds[n] for n in NAMES]), GOLD[f"{tag}/prior_bounds"])
        xc, lj = prop.inverse_rescale(xp, return_unit_hypercube=True)
        assert np.array_equal(stack(xc, NAMES), GOLD[f"{tag}/x_back_cube"], equal_nan=True)
        assert np.array_equal(lj, GOLD[f"{tag}/log_j_inv_cube"], equal_nan=True)
    else:
        assert prop.prior_bounds is prop.model.bounds


@pytest.mark.parametrize("tag", ["cube", "cube_reversed"])
def test_unit_hypercube_weights_samples_and_backward_pass_match_the_reference(tag):
    prop, _ = make(tag)
    cube = numpy_array_to_live_points(GOLD[f"{tag}/cube_points"], NAMES)
    with np.errstate(divide="ignore", invalid="ignore"):
        log_w = prop.compute_weights(cube, GOLD[f"{tag}/cube_log_q"])
    assert np.array_equal(log_w, GOLD[f"{tag}/cube_log_w"])  # -inf outside [0, 1): the upper face included
    edge = numpy_array_to_live_points(np.array([[0.5, 0.5, 0.5, 0.5, 1.0], [0.0, 0.5, 0.5, 0.5, 0.5]]), NAMES)
    with np.errstate(divide="ignore"):
        w_edge = prop.compute_weights(edge, np.zeros(2))
    assert np.isneginf(w_edge[0]) and w_edge[1] == 0.0  # [0, 1): the upper face is outside, the lower one inside
    samples = prop.convert_to_samples(cube)
    assert np.array_equal(stack(samples, NAMES), GOLD[f"{tag}/cube_samples"])
    assert np.array_equal(samples["logP"], GOLD[f"{tag}/cube_samples_logP"])
    # the fused draw kernel with the cube's bounds, in the cube and mapped on to the physical space
    z = GOLD[f"{tag}/bp_z"]
    xc, log_q_c, z_c = prop.backward_pass(z, return_z=True, return_unit_hypercube=True)
    # (the reference tests no bounds in the cube - rows outside get weight -inf from the cube prior later; the kernel
    # drops them here: the rows INSIDE are the same rows)
    want = GOLD[f"{tag}/bp_x_cube"]
    inside = np.all((want >= 0.0) & (want < 1.0), axis=1)
    np.testing.assert_allclose(stack(xc, NAMES), want[inside], rtol=0, atol=2e-6)
    np.testing.assert_allclose(log_q_c, GOLD[f"{tag}/bp_log_q_cube"][inside], rtol=1e-5, atol=1e-5)
    x, log_q, z_kept = prop.backward_pass(z, return_z=True)
    assert np.array_equal(z_kept, GOLD[f"{tag}/bp_z_kept"])  # in the cube <=> inside the prior bounds
    np.testing.assert_allclose(stack(x, NAMES), GOLD[f"{tag}/bp_x"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(log_q, GOLD[f"{tag}/bp_log_q"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("lean", [True, False])
def test_populate_in_the_unit_hypercube(lean):
    """A population kept in the unit hypercube (``flowproposal.py:441-446``): every loop of ``populate`` runs with the
    cube's bounds and prior (log 1), the pool is mapped back by ``convert_to_samples`` - in bounds, with the model's
    log-prior and likelihood - and equals, ROW FOR ROW up to the last bits of the affine maps, the pool of the same
    proposal working in the physical space with the equivalent reparameterisation (z-score of u = (x - lo) / (hi - lo)
    against z-score of x: the same x' for the same flow, the constant Jacobians cancel in the weights)."""
    def run(**kw):
        torch.manual_seed(3)
        model = GaussianModel(6)
        prop = FlowProposal(model, rng=np.random.default_rng(5), poolsize=400, drawsize=40_000, device_seed=11,
                            flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2), lean=lean, **kw)
        prop.initialise()
        live = model.new_point(500, rng=np.random.default_rng(1))
        for n in prop.names:
            live[n] *= 0.2
        prop.check_state(live)
        prop.populate(None, n_samples=400, max_samples=10**8)
        return prop

    cube, plain = run(map_to_unit_hypercube=True), run()
    assert cube.last_populate_path == "trains" and cube.last_populate_lean == lean
    assert cube.prior_mode == "device" and cube._ll_spec is None  # the device likelihood needs physical rows
    xs = stack(cube.samples, cube.names)
    assert xs.shape == (400, 6) and cube.model.in_bounds(cube.samples).all()
    assert np.array_equal(cube.samples["logP"], cube.model.log_prior(cube.samples))
    np.testing.assert_allclose(cube.samples["logL"], cube.model.log_likelihood(cube.samples), rtol=1e-12)
    u = stack(cube.x, cube.names)  # the population itself lives in the cube
    assert ((u >= 0.0) & (u < 1.0)).all()
    assert cube.n_proposed == plain.n_proposed and cube.n_accepted == plain.n_accepted
    np.testing.assert_allclose(xs, stack(plain.samples, plain.names), rtol=0, atol=1e-9)


def test_save_training_data_and_the_default_hook(tmp_path):
    """``save_training_data=True`` writes the live points of every training as JSON under
    ``<output>/training/block_<k>/training_data.json`` and trains there (``base.py:887-900``;
    ``utils/io.py:116-130``); ``use_default_reparameterisations=True`` calls the (empty) hook (``base.py:437-439,
    602-603``)."""
    calls = []

    class Hooked(FlowProposal):
        def add_default_reparameterisations(self):
            calls.append(1)

    model = GaussianModel(4)
    prop = Hooked(model, rng=np.random.default_rng(2), poolsize=100, drawsize=2000, device_seed=3,
                  flow_config=dict(ftype="realnvp", n_neurons=8, n_blocks=2),
                  training_config=dict(max_epochs=3, patience=3, batch_size=100), output=str(tmp_path),
                  save_training_data=True, use_default_reparameterisations=True)
    assert prop.use_default_reparameterisations and not FlowProposal.use_default_reparameterisations
    prop.initialise()
    assert calls == [1]
    live = model.new_point(300, rng=np.random.default_rng(1))
    prop.train(live)
    block = tmp_path / "training" / "block_0"
    saved = json.loads((block / "training_data.json").read_text())
    assert set(saved) == set(live.dtype.names)
    for n in model.names:
        assert np.array_equal(np.asarray(saved[n]), live[n])
    assert os.path.dirname(prop.flow.weights_file) == str(block) and os.path.exists(prop.flow.weights_file)
    prop.train(live)
    assert (tmp_path / "training" / "block_1" / "training_data.json").exists()
    with pytest.raises(TypeError, match="Unknown FlowProposal arguments"):
        FlowProposal(model, poolsize=10, not_an_argument=1)
