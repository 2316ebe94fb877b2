"""Angle reparameterisations on the device path (runs on a real MI355X): ``angle-2pi`` / ``angle-pi`` / ``angle`` /
``periodic`` (``src/nessai/reparameterisations/angle.py:17-183``, registry ``reparameterisations/__init__.py:158-177``) -
an angle becomes a point of the plane, the flow has one input more per angle, the draw kernel's epilogue turns the two
columns back into the angle (``atan2``, modulo 2 pi for a range that starts at 0) and the auxiliary radius, whose
chi(2) density joins the prior and whose -log joins the Jacobian.

Goldens: ``tests/golden/angle_goldens.npz`` - the reference's own ``FlowProposal`` with these reparameterisations
(``tests/golden/make_angle_goldens.py``); the float64 glue on the host is bit-exact (``tests/test_reparam.py``), the
fused kernel is held to float32 here.
"""
import os

import numpy as np
import pytest
import torch

from nessai_amd.livepoint import numpy_array_to_live_points
from nessai_amd.model import UniformBoxModel
from nessai_amd.proposal import FlowProposal

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "angle_goldens.npz"))
NAMES = ["a", "phi", "psi", "c", "tau"]
BOUNDS = {"a": [-3.0, 7.0], "phi": [0.0, 2.0 * np.pi], "psi": [-np.pi / 2.0, np.pi / 2.0], "c": [0.5, 2.5],
          "tau": [0.0, 4.0]}
CONFIG = {"a": "rescaletobounds", "phi": "angle-2pi", "psi": "angle-pi", "c": "z-score", "tau": "periodic"}


class BoxModel(UniformBoxModel):
    def __init__(self):
        self.names = list(NAMES)
        self.bounds = {n: list(BOUNDS[n]) for n in NAMES}

    def log_likelihood(self, x):
        return -0.5 * np.sum(self.unstructured_view(x) ** 2, axis=-1)


def stack(x, names):
    return np.stack([np.asarray(x[n], dtype=np.float64) for n in names], axis=1)


def make(prior="auto", **extra):
    prop = FlowProposal(BoxModel(), rng=np.random.default_rng(4242), poolsize=50, drawsize=400, device_seed=1,
                        flow_config=dict(ftype="realnvp", n_blocks=2, n_neurons=8, n_layers=1,
                                         batch_norm_between_layers=False, linear_transform="permutation"),
                        reparameterisations=CONFIG, prior=prior, **extra)
    prop.initialise()
    state = {k[len("state/"):]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith("state/")}
    prop.flow.model.load_state_dict(state)
    prop.flow.model.eval()
    live = numpy_array_to_live_points(GOLD["live"], NAMES)
    prop.check_state(live)
    return prop, live


@pytest.mark.parametrize("prior", ["device", "host"])
def test_angle_proposal_matches_the_reference(prior):
    prop, live = make(prior)
    assert prop.prime_parameters == [str(p) for p in GOLD["prime_parameters"]]
    assert prop.dims == prop.prime_dims == 8 and prop.flow.model.features == 8
    assert prop.names == ["a", "phi", "phi_radial", "psi", "psi_radial", "c", "tau", "tau_radial"]
    x_prime, log_j = prop.rescale(live)  # (the proposal's generator starts where the generator script put the reference's)
    assert np.array_equal(stack(x_prime, prop.prime_parameters), GOLD["x_prime"])
    assert np.array_equal(log_j, GOLD["log_j"])
    # the fused draw kernel on the reference's latents: the kept rows, x with its auxiliary radii, log q, the weights
    x, log_q, z_kept = prop.backward_pass(GOLD["bp_z"], return_z=True)
    assert np.array_equal(z_kept, GOLD["bp_z_kept"])
    names = [str(n) for n in GOLD["bp_names"]]
    np.testing.assert_allclose(stack(x, names), GOLD["bp_x"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(log_q, GOLD["bp_log_q"], rtol=1e-5, atol=2e-5)
    with np.errstate(divide="ignore"):
        log_w = prop.compute_weights(x, log_q)
    np.testing.assert_allclose(log_w, GOLD["bp_log_w"], rtol=1e-5, atol=2e-5)
    samples = prop.convert_to_samples(x)
    assert list(samples.dtype.names) == [str(n) for n in GOLD["bp_sample_names"]]  # the radii stay behind


@pytest.mark.parametrize("prior", ["device", "host"])
def test_populate_with_angles(prior):
    """Every loop of ``populate`` with D' = D + (number of angles) device columns: the lean passes + replay equal the
    full passes bit for bit, the pool holds model parameters only, inside their bounds, with the model's log-prior;
    the auxiliary radii of the population follow chi(2) reweighted by the flow (finite, positive); and the weights of
    the passes used the radial prior (device table or host prior: the same pool)."""
    def run(lean, prior_mode):
        torch.manual_seed(3)
        prop = FlowProposal(BoxModel(), rng=np.random.default_rng(5), poolsize=300, drawsize=30_000, device_seed=11,
                            flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2), reparameterisations=CONFIG,
                            lean=lean, prior=prior_mode)
        prop.initialise()
        live = numpy_array_to_live_points(GOLD["live"], NAMES)
        prop.train_data = live
        prop.check_state(live)
        prop.populate(None, n_samples=300, max_samples=10**8)
        return prop

    a, b = run(True, prior), run(False, prior)
    if prior == "device":
        assert a.last_populate_path == "trains" and a.last_populate_lean and not b.last_populate_lean
    assert a.n_proposed == b.n_proposed and a.n_accepted == b.n_accepted
    assert np.array_equal(stack(a.samples, NAMES), stack(b.samples, NAMES))
    assert list(a.samples.dtype.names) == NAMES + ["logP", "logL", "it"]
    assert a.model.in_bounds(a.samples).all() and len(a.samples) == 300
    assert np.array_equal(a.samples["logP"], a.model.log_prior(a.samples))
    radii = stack(a.x, ["phi_radial", "psi_radial", "tau_radial"])
    assert np.isfinite(radii).all() and (radii > 0).all()
    if prior == "host":  # the host prior (model + radial, base.py:1046-1049) and the device table give the same pool
        c = run(True, "device")
        assert c.n_proposed == a.n_proposed
        np.testing.assert_allclose(stack(c.samples, NAMES), stack(a.samples, NAMES), rtol=0, atol=1e-9)


def test_angle_flow_trains_and_populates_end_to_end(tmp_path):
    model = BoxModel()
    prop = FlowProposal(model, rng=np.random.default_rng(7), poolsize=200, drawsize=20_000, device_seed=3,
                        flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2),
                        training_config=dict(max_epochs=30, patience=30, batch_size=200), reparameterisations=CONFIG,
                        output=str(tmp_path))
    prop.initialise()
    live = model.new_point(1000, rng=np.random.default_rng(1))
    live["phi"] = (1.0 + 0.3 * np.random.default_rng(2).normal(size=1000)) % (2 * np.pi)  # a bump away from the wrap
    live["logL"] = model.log_likelihood(live)
    prop.train(live)
    prop.populate(live[0], n_samples=200, max_samples=10**8)
    s = prop.samples
    assert len(s) == 200 and model.in_bounds(s).all()
    # the flow learnt the bump (the pool itself follows the PRIOR inside the contour - the weights are prior / q):
    # draws of the flow, mapped back, lie near phi = 1
    x, _ = prop.backward_pass(np.random.default_rng(4).normal(size=(4000, prop.dims)))
    assert np.mean(np.abs(x["phi"] - 1.0) < 1.0) > 0.8
    drawn = prop.draw(live[0])
    assert drawn.dtype.names == s.dtype.names


def test_gw_parameter_set_with_its_angle_reparameterisations():
    """Config 5's model (15 bilby-style parameters, separable priors on the device) with the angle maps the reference's
    GW defaults give such a parameter set (``angle-pi`` for psi, ``angle-2pi`` for the 2 pi-periodic angles,
    ``angle-sine`` - rescale-to-bounds - for the inclinations): a 20-input flow, the separable prior table extended by
    the radii's chi(2) rows, lean passes."""
    from nessai_amd.model import GWSurrogateModel

    model = GWSurrogateModel()
    config = {"psi": "angle-pi", "phase": "angle-2pi", "ra": "angle-2pi", "phi_12": "angle-2pi", "phi_jl": "angle-2pi",
              "tilt_1": "angle-sine", "tilt_2": "angle-sine", "theta_jn": "angle-sine", "dec": "angle-sine"}
    torch.manual_seed(1)
    prop = FlowProposal(model, rng=np.random.default_rng(3), poolsize=300, drawsize=40_000, device_seed=5,
                        flow_config=dict(ftype="nsf", n_neurons=32, n_blocks=2), reparameterisations=config)
    prop.initialise()
    assert prop.dims == 20 and prop.flow.model.features == 20 and prop.prior_mode == "device"
    assert int(prop._prior_kind_dev.numel()) == 20 and int((prop._prior_kind_dev == 5).sum()) == 5
    live = model.new_point(600, rng=np.random.default_rng(1))
    prop.check_state(live)
    prop.populate(None, n_samples=300, max_samples=10**8)
    assert prop.last_populate_path == "trains" and prop.last_populate_lean
    s = prop.samples
    assert len(s) == 300 and list(s.dtype.names) == model.names + ["logP", "logL", "it"]
    assert model.in_bounds(s).all() and np.isfinite(s["logP"]).all()
    np.testing.assert_allclose(s["logP"], model.log_prior(s), rtol=1e-13)
    # the weights of one pass through backward_pass + the host formula equal model prior + radial prior - log q
    z = np.random.default_rng(2).normal(size=(2000, 20))
    x, log_q = prop.backward_pass(z)
    from scipy import stats

    want = model.log_prior(x) + sum(stats.chi(2).logpdf(x[n + "_radial"]) for n in ("psi", "phase", "ra", "phi_12", "phi_jl"))
    np.testing.assert_allclose(prop.log_prior(x), want, rtol=1e-13)


# -- a pair of angles (angle.py:186-538; goldens: tests/golden/make_angle_pair_goldens.py) ---------------------------------

PAIR_GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "angle_pair_goldens.npz"))
PAIR_NAMES = ["a", "dec", "c", "ra", "az", "zen"]
PAIR_BOUNDS = {"a": [-3.0, 7.0], "dec": [-np.pi / 2.0, np.pi / 2.0], "c": [0.5, 2.5], "ra": [0.0, 2.0 * np.pi],
               "az": [-np.pi, np.pi], "zen": [0.0, np.pi]}
PAIR_CONFIG = {"a": "rescaletobounds", "dec": {"reparameterisation": "angle-pair", "parameters": ["dec", "ra"]},
               "c": "z-score", "az": {"reparameterisation": "angle-pair", "parameters": ["az", "zen"]}}
PAIR_DEVICE_NAMES = ["a", "ra", "dec", "ra_dec_radial", "c", "az", "zen", "az_zen_radial"]


class PairModel(UniformBoxModel):
    def __init__(self):
        self.names = list(PAIR_NAMES)
        self.bounds = {n: list(PAIR_BOUNDS[n]) for n in PAIR_NAMES}

    def log_likelihood(self, x):
        return -0.5 * np.sum(self.unstructured_view(x) ** 2, axis=-1)


@pytest.mark.parametrize("prior", ["device", "host"])
def test_angle_pair_proposal_matches_the_reference(prior):
    """Two pairs (ra-dec with the modulo, az-zen without; the first given vertical angle first and split by another
    parameter in the model): three x' columns per pair, the epilogue's atan2s / radius / -2 log r - log cos|sin terms,
    the chi(3) rows of the prior table."""
    prop = FlowProposal(PairModel(), rng=np.random.default_rng(4242), poolsize=50, drawsize=400, device_seed=1,
                        flow_config=dict(ftype="realnvp", n_blocks=2, n_neurons=8, n_layers=1,
                                         batch_norm_between_layers=False, linear_transform="permutation"),
                        reparameterisations=PAIR_CONFIG, prior=prior)
    prop.initialise()
    g = PAIR_GOLD
    state = {k[len("state/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
    prop.flow.model.load_state_dict(state)
    prop.flow.model.eval()
    live = numpy_array_to_live_points(g["live"], PAIR_NAMES)
    prop.check_state(live)
    assert prop.prime_parameters == [str(p) for p in g["prime_parameters"]]
    assert prop.dims == prop.prime_dims == 8 and prop.flow.model.features == 8
    assert prop.names == PAIR_DEVICE_NAMES
    if prior == "device":
        assert prop._prior_kind_dev.tolist() == [0, 0, 0, 5, 0, 0, 0, 5]
    x_prime, log_j = prop.rescale(live)
    assert np.array_equal(stack(x_prime, prop.prime_parameters), g["x_prime"])
    assert np.array_equal(log_j, g["log_j"])
    x, log_q, z_kept = prop.backward_pass(g["bp_z"], return_z=True)
    assert np.array_equal(z_kept, g["bp_z_kept"])
    names = [str(n) for n in g["bp_names"]]
    np.testing.assert_allclose(stack(x, names), g["bp_x"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(log_q, g["bp_log_q"], rtol=1e-5, atol=5e-5)
    with np.errstate(divide="ignore"):
        log_w = prop.compute_weights(x, log_q)
    np.testing.assert_allclose(log_w, g["bp_log_w"], rtol=1e-5, atol=5e-5)
    samples = prop.convert_to_samples(x)
    assert list(samples.dtype.names) == [str(n) for n in g["bp_sample_names"]]


@pytest.mark.parametrize("prior", ["device", "host"])
def test_populate_with_angle_pairs(prior):
    def run(lean, prior_mode):
        torch.manual_seed(3)
        prop = FlowProposal(PairModel(), rng=np.random.default_rng(5), poolsize=300, drawsize=30_000, device_seed=11,
                            flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2), reparameterisations=PAIR_CONFIG,
                            lean=lean, prior=prior_mode)
        prop.initialise()
        live = numpy_array_to_live_points(PAIR_GOLD["live"], PAIR_NAMES)
        prop.train_data = live
        prop.check_state(live)
        prop.populate(None, n_samples=300, max_samples=10**8)
        return prop

    a, b = run(True, prior), run(False, prior)
    if prior == "device":
        assert a.last_populate_path == "trains" and a.last_populate_lean and not b.last_populate_lean
    assert a.n_proposed == b.n_proposed and a.n_accepted == b.n_accepted
    assert np.array_equal(stack(a.samples, PAIR_NAMES), stack(b.samples, PAIR_NAMES))
    assert list(a.samples.dtype.names) == PAIR_NAMES + ["logP", "logL", "it"]
    assert a.model.in_bounds(a.samples).all() and len(a.samples) == 300
    assert np.array_equal(a.samples["logP"], a.model.log_prior(a.samples))
    radii = stack(a.x, ["ra_dec_radial", "az_zen_radial"])
    assert np.isfinite(radii).all() and (radii > 0).all()
    # the pool follows the prior: uniform in the angles (NOT isotropic on the sphere - the box model's prior),
    # the Jacobian's log cos / log sin terms are what make that so
    assert abs(np.mean(a.samples["dec"])) < 0.3 and abs(np.mean(a.samples["zen"]) - np.pi / 2) < 0.3
    if prior == "host":
        c = run(True, "device")
        assert c.n_proposed == a.n_proposed
        np.testing.assert_allclose(stack(c.samples, PAIR_NAMES), stack(a.samples, PAIR_NAMES), rtol=0, atol=1e-9)


def test_gw_sky_position_as_an_angle_pair():
    """Config 5's parameter set with the sky position as the reference's GW defaults treat it: (ra, dec) as one
    ``angle-pair`` - the flow sees a point of space instead of two periodic / bounded columns."""
    from scipy import stats

    from nessai_amd.model import GWSurrogateModel

    model = GWSurrogateModel()
    config = {"psi": "angle-pi", "phase": "angle-2pi", "angle-pair": {"parameters": ["ra", "dec"]},
              "tilt_1": "angle-sine", "tilt_2": "angle-sine", "theta_jn": "angle-sine"}
    torch.manual_seed(1)
    prop = FlowProposal(model, rng=np.random.default_rng(3), poolsize=300, drawsize=40_000, device_seed=5,
                        flow_config=dict(ftype="nsf", n_neurons=32, n_blocks=2), reparameterisations=config)
    prop.initialise()
    assert prop.dims == 18 and "ra_dec_x" in prop.prime_parameters and prop.prior_mode == "device"
    assert int((prop._prior_kind_dev == 5).sum()) == 3
    live = model.new_point(600, rng=np.random.default_rng(1))
    prop.check_state(live)
    prop.populate(None, n_samples=300, max_samples=10**8)
    assert prop.last_populate_path == "trains" and prop.last_populate_lean
    s = prop.samples
    assert len(s) == 300 and list(s.dtype.names) == model.names + ["logP", "logL", "it"]
    assert model.in_bounds(s).all() and np.isfinite(s["logP"]).all()
    np.testing.assert_allclose(s["logP"], model.log_prior(s), rtol=1e-13)
    z = np.random.default_rng(2).normal(size=(2000, 18))
    x, log_q = prop.backward_pass(z)
    want = (model.log_prior(x) + stats.chi(2).logpdf(x["psi_radial"]) + stats.chi(2).logpdf(x["phase_radial"])
            + stats.chi(3).logpdf(x["ra_dec_radial"]))
    np.testing.assert_allclose(prop.log_prior(x), want, rtol=1e-13)


# -- to-cartesian (angle.py:189-232; goldens: tests/golden/make_to_cartesian_goldens.py) -----------------------------------

CART_GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "to_cartesian_goldens.npz"))
CART_NAMES = ["s", "c", "d", "h"]
CART_BOUNDS = {"s": [-3.0, 7.0], "c": [0.5, 2.5], "d": [2.0, 5.0], "h": [0.0, 1.5]}
CART_CONFIG = {"s": "to-cartesian", "c": "z-score", "d": {"reparameterisation": "to-cartesian", "mode": "duplicate"},
               "h": {"reparameterisation": "to-cartesian", "mode": "half"}}


class CartModel(UniformBoxModel):
    def __init__(self):
        self.names = list(CART_NAMES)
        self.bounds = {n: list(CART_BOUNDS[n]) for n in CART_NAMES}

    def log_likelihood(self, x):
        return -0.5 * np.sum(self.unstructured_view(x) ** 2, axis=-1)


@pytest.mark.parametrize("prior", ["device", "host"])
def test_to_cartesian_proposal_matches_the_reference(prior):
    """The three modes in one proposal (split, duplicate - ``rescale`` returns twice the rows -, half): x' against the
    reference bit for bit, the fused kernel's fold |atan2| / pi -> bounds, radii, log q and weights to float32."""
    prop = FlowProposal(CartModel(), rng=np.random.default_rng(4242), poolsize=50, drawsize=400, device_seed=1,
                        flow_config=dict(ftype="realnvp", n_blocks=2, n_neurons=8, n_layers=1,
                                         batch_norm_between_layers=False, linear_transform="permutation"),
                        reparameterisations=CART_CONFIG, prior=prior)
    prop.initialise()
    g = CART_GOLD
    state = {k[len("state/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("state/")}
    prop.flow.model.load_state_dict(state)
    prop.flow.model.eval()
    live = numpy_array_to_live_points(g["live"], CART_NAMES)
    prop.check_state(live)
    assert prop.prime_parameters == [str(p) for p in g["prime_parameters"]] and prop.dims == 7
    assert prop.names == ["s", "s_radial", "c", "d", "d_radial", "h", "h_radial"]
    prop.rng.bit_generator.state = np.random.default_rng(4242).bit_generator.state
    x_prime, log_j = prop.rescale(live)
    assert len(x_prime) == 2 * len(live)
    assert np.array_equal(stack(x_prime, prop.prime_parameters), g["x_prime"])
    assert np.array_equal(log_j, g["log_j"])
    x, log_q, z_kept = prop.backward_pass(g["bp_z"], return_z=True)
    assert np.array_equal(z_kept, g["bp_z_kept"])
    names = [str(n) for n in g["bp_names"]]
    np.testing.assert_allclose(stack(x, names), g["bp_x"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(log_q, g["bp_log_q"], rtol=1e-5, atol=5e-5)
    with np.errstate(divide="ignore"):
        log_w = prop.compute_weights(x, log_q)
    np.testing.assert_allclose(log_w, g["bp_log_w"], rtol=1e-5, atol=5e-5)


def test_populate_with_to_cartesian(tmp_path):
    model = CartModel()

    def run(lean):
        torch.manual_seed(3)
        prop = FlowProposal(model, rng=np.random.default_rng(5), poolsize=300, drawsize=30_000, device_seed=11,
                            flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2),
                            training_config=dict(max_epochs=20, patience=20, batch_size=500),
                            reparameterisations=CART_CONFIG, lean=lean, output=str(tmp_path / ("lean" if lean else "full")))
        prop.initialise()
        live = model.new_point(1000, rng=np.random.default_rng(1))
        live["logL"] = model.log_likelihood(live)
        prop.train(live)  # (2000 training rows: the duplicate mode)
        prop.populate(live[0], n_samples=300, max_samples=10**8)
        return prop

    a, b = run(True), run(False)
    assert a.n_proposed == b.n_proposed and np.array_equal(stack(a.samples, CART_NAMES), stack(b.samples, CART_NAMES))
    assert len(a.samples) == 300 and model.in_bounds(a.samples).all()
    assert list(a.samples.dtype.names) == CART_NAMES + ["logP", "logL", "it"]
    assert np.array_equal(a.samples["logP"], model.log_prior(a.samples))
