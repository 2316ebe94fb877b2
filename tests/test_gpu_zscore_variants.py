"""The z-score reparameterisations with a rescaling in front of the affine map or behind it, on the device path (runs on
a real MI355X): ``z-score-logit``, ``log-z-score`` / ``log-standardise``, ``z-score-inv-gaussian-cdf``,
``z-score-gaussian-cdf`` and their ``zscore-`` spellings (``src/nessai/reparameterisations/__init__.py:99-162``;
``ScaleAndShift``: ``rescale.py:119-312``; rescaling functions: ``utils/rescaling.py:290-415``) - the draw kernel's
epilogue applies the affine map, then the sigmoid / exp / Gaussian CDF (or the inverse Gaussian CDF first), with the
row's log-Jacobian term (kinds ``NSA_RP_SS_*``, ``NSA_RP_INV_GAUSSIAN_CDF_SS``).

Goldens: ``tests/golden/zscore_variant_goldens.npz`` - the reference's own ``FlowProposal``
(``make_zscore_variant_goldens.py``); the float64 glue on the host is bit-exact (``tests/test_reparam.py``), the fused
kernel is held to float32 here.
"""
import os

import numpy as np
import pytest
import torch

from nessai_amd.livepoint import numpy_array_to_live_points
from nessai_amd.model import UniformBoxModel
from nessai_amd.proposal import FlowProposal

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zscore_variant_goldens.npz"))
NAMES = ["a", "b", "c", "d", "e"]
BOUNDS = {"a": [0.0, 1.0], "b": [0.1, 9.0], "c": [0.0, 1.0], "d": [-5.0, 5.0], "e": [0.5, 2.5]}
CONFIG = {"a": "z-score-logit", "b": "log-z-score", "c": "zscore-inv-gaussian-cdf", "d": "z-score-gaussian-cdf",
          "e": "z-score"}


class BoxModel(UniformBoxModel):
    def __init__(self):
        self.names = list(NAMES)
        self.bounds = {n: list(BOUNDS[n]) for n in NAMES}

    def log_likelihood(self, x):
        return -0.5 * np.sum(self.unstructured_view(x) ** 2, axis=-1)


def stack(x, names):
    return np.stack([np.asarray(x[n], dtype=np.float64) for n in names], axis=1)


@pytest.mark.parametrize("prior", ["device", "host"])
def test_zscore_variant_proposal_matches_the_reference(prior):
    prop = FlowProposal(BoxModel(), rng=np.random.default_rng(4242), poolsize=50, drawsize=400, device_seed=1,
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
    assert prop.prime_parameters == [str(p) for p in g["prime_parameters"]] and prop.dims == 5
    x_prime, log_j = prop.rescale(live)
    assert np.array_equal(stack(x_prime, prop.prime_parameters), g["x_prime"])
    assert np.array_equal(log_j, g["log_j"])
    # the fused draw kernel on the reference's latents (rows whose parameters leave the box are dropped by both)
    x, log_q, z_kept = prop.backward_pass(g["bp_z"], return_z=True)
    assert np.array_equal(z_kept, g["bp_z_kept"])
    names = [str(n) for n in g["bp_names"]]
    np.testing.assert_allclose(stack(x, names), g["bp_x"], rtol=2e-5, atol=3e-5, equal_nan=True)
    # a row whose Gaussian-CDF column left (0, 1) has no inverse: the reference keeps it with NaN for the parameter and
    # for log q; here log q = +inf (NaN marks a dropped row on the device) - never accepted either way
    lost = np.isnan(g["bp_log_q"])
    assert lost.any() and np.array_equal(np.isnan(stack(x, names)[:, 3]), lost) and np.all(np.isposinf(log_q[lost]))
    np.testing.assert_allclose(log_q[~lost], g["bp_log_q"][~lost], rtol=1e-5, atol=1e-4)
    with np.errstate(divide="ignore", invalid="ignore"):
        log_w = prop.compute_weights(x, log_q)
    np.testing.assert_allclose(log_w[~lost], g["bp_log_w"][~lost], rtol=1e-5, atol=1e-4)
    assert not np.any(log_w[lost] > -np.inf)


def test_populate_with_zscore_variants(tmp_path):
    model = BoxModel()

    def run(lean):
        torch.manual_seed(3)
        prop = FlowProposal(model, rng=np.random.default_rng(5), poolsize=300, drawsize=30_000, device_seed=11,
                            flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2),
                            training_config=dict(max_epochs=20, patience=20, batch_size=500),
                            reparameterisations=CONFIG, lean=lean, output=str(tmp_path / ("lean" if lean else "full")))
        prop.initialise()
        live = model.new_point(1000, rng=np.random.default_rng(1))
        live["logL"] = model.log_likelihood(live)
        prop.train(live)
        prop.populate(live[0], n_samples=300, max_samples=10**8)
        return prop

    a, b = run(True), run(False)
    assert a.n_proposed == b.n_proposed and np.array_equal(stack(a.samples, NAMES), stack(b.samples, NAMES))
    assert len(a.samples) == 300 and model.in_bounds(a.samples).all()
    assert np.array_equal(a.samples["logP"], model.log_prior(a.samples))
