"""Parity of the fused populate path against the CPU oracle (runs on a real MI355X).

The oracle (``oracle/proposal.py``) restates ``FlowProposal.populate`` /
``backward_pass``; the device path must reproduce it on identical weights,
latents and uniforms.  The glue arithmetic is float64 on both sides, so apart
from the float32 flow outputs (summation order) the results are identical;
accept decisions can only differ for draws within ~1e-6 of the threshold.
"""
import os

import numpy as np
import pytest
import torch
from scipy import stats

from helpers import DEV, load_c3_fixture
from nessai_amd.model import GaussianModel, RosenbrockModel
from nessai_amd.proposal import FlowProposal
from oracle import philox
from oracle import proposal as OP

pytestmark = pytest.mark.gpu


def build(dims, ftype, flow_kwargs, model_cls=RosenbrockModel, seed=3, poolsize=500, drawsize=4000,
          state=None, shift=None, scale=None, **kwargs):
    """(device proposal, oracle proposal) sharing weights and z-score constants."""
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    model = model_cls(dims)
    omodel = {RosenbrockModel: OP.RosenbrockModel, GaussianModel: OP.GaussianModel}[model_cls](dims)
    cfg = dict(ftype=ftype, **flow_kwargs)
    prop = FlowProposal(model, rng=np.random.default_rng(seed), flow_config=cfg, poolsize=poolsize,
                        drawsize=drawsize, device_seed=1451, **kwargs)
    prop.initialise()
    ocfg = dict(cfg, n_inputs=dims)
    ocfg.setdefault("n_neurons", 2 * dims)
    ocfg.setdefault("n_blocks", 4)
    ocfg.setdefault("n_layers", 2)
    ofm = OP.OracleFlowModel(ocfg, rng=np.random.default_rng(seed))
    if state is None:
        state = ofm.model.state_dict()
    ofm.model.load_state_dict(state)
    ofm.model.eval()
    prop.flow.model.load_state_dict(state)
    prop.flow.model.eval()
    oprop = OP.OracleFlowProposal(omodel, ofm, rng=rng, poolsize=poolsize, drawsize=drawsize,
                                  accumulate_weights=kwargs.get("accumulate_weights", False))
    if shift is None:
        shift = rng.normal(size=dims) * 0.3
        scale = rng.uniform(0.5, 1.5, size=dims)
    prop.set_rescaling_constants(shift, scale)
    for i, n in enumerate(omodel.names):
        oprop.shift[n] = float(shift[i])
        oprop.scale[n] = float(scale[i])
    return prop, oprop


def test_backward_pass_matches_oracle():
    """x, log_q and the kept rows of backward_pass (flowproposal.py:345-389)."""
    prop, oprop = build(6, "nsf", dict(n_neurons=16, n_blocks=3))
    z = np.random.default_rng(0).normal(size=(5000, 6)).astype(np.float32).astype(np.float64) * 1.3
    x, log_q, z_kept = prop.backward_pass(z, return_z=True)
    x_o, log_q_o, z_o = oprop.backward_pass(z)
    assert len(x) == len(x_o) and np.array_equal(z_kept, z_o)
    for n in prop.names:
        np.testing.assert_allclose(x[n], x_o[n], rtol=0, atol=2e-4)
    np.testing.assert_allclose(log_q, log_q_o, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("cfg", [
    ("nsf", dict(n_neurons=16, n_blocks=3), 6, RosenbrockModel),
    ("realnvp", dict(n_neurons=16, n_blocks=3), 8, GaussianModel),
])
def test_populate_injected_matches_oracle(cfg):
    """Same latents and uniforms in -> same accepted set out (flowproposal.py:431-512)."""
    ftype, fk, dims, mcls = cfg
    prop, oprop = build(dims, ftype, fk, model_cls=mcls, poolsize=300, drawsize=3000)
    g = np.random.default_rng(11)
    batches = [(g.normal(size=(3000, dims)).astype(np.float32).astype(np.float64), g.random(3000)) for _ in range(40)]
    radius = OP.compute_radius(dims, 0.95)

    # oracle run, remembering how many uniforms each pass consumed
    used = []

    class Rng:
        def __init__(self):
            self.k = 0

        def random(self, n):
            z, u = batches[self.k]
            kept = used[self.k]
            self.k += 1
            return u[kept]

        def permutation(self, n):
            return np.arange(n)

    # the reference draws uniforms only for rows that survive the cuts; map each
    # surviving row back to its draw index so both sides see the same u per draw
    def oracle_batches():
        for z, _ in batches:
            yield z

    orig_backward = oprop.backward_pass

    def tracking_backward(z_trunc):
        x, lq, zk = orig_backward(z_trunc)
        zb = batches[len(used)][0]
        lookup = {row.tobytes(): i for i, row in enumerate(zb)}
        used.append(np.array([lookup[row.tobytes()] for row in zk], dtype=np.int64))
        return x, lq, zk

    oprop.backward_pass = tracking_backward
    oprop.injected_latents = oracle_batches()
    oprop.rng = Rng()
    oprop.populate(n_samples=300, r=radius)

    prop.populate(None, n_samples=300, r=radius, injected=iter(batches))
    assert prop.n_proposed == oprop.n_proposed
    # accept decisions agree except for draws within float32 round-off of the threshold
    assert abs(prop.n_accepted - oprop.n_accepted) <= max(2, int(0.01 * oprop.n_accepted))
    a = np.stack([prop.x[n] for n in prop.names], axis=1)
    b = np.stack([oprop.x[n] for n in oprop.names], axis=1)
    if prop.n_accepted == oprop.n_accepted:
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-4)
    np.testing.assert_allclose(prop.samples["logP"], oprop.model.log_prior(oprop.samples)[0], rtol=0, atol=1e-12)
    assert np.isfinite(prop.samples["logL"]).all()
    assert prop.populated and sorted(prop.indices) == list(range(prop.samples.size))


REPARAM_CONFIG = {"logit": ["x_0"], "log-rescale": ["x_3"], "default": ["x_1"], "offset": ["x_5"],
                  "null": ["x_2"]}  # x_4: z-score


def build_with_reparameterisations(seed=3, **kwargs):
    """6-D Rosenbrock proposal pair with a mixed reparameterisation table."""
    from oracle import reparam as OR

    prop, oprop = build(6, "nsf", dict(n_neurons=16, n_blocks=3), reparameterisations=REPARAM_CONFIG, **kwargs)
    b = {n: [-5.0, 5.0] for n in prop.names}
    oprop.reparam = OR.OracleCombined([
        OR.OracleRescaleToBounds(["x_0"], b, rescale_bounds=[0.0, 1.0], update_bounds=False,
                                 post_rescaling="logit"),
        OR.OracleRescaleToBounds(["x_3"], b, rescale_bounds=[0.0, 1.0], update_bounds=False,
                                 post_rescaling="log"),
        OR.OracleRescaleToBounds(["x_1"], b),
        OR.OracleRescaleToBounds(["x_5"], b, offset=True),
        OR.OracleNull(["x_2"]),
        OR.OracleScaleAndShift(["x_4"], estimate_scale=True, estimate_shift=True),
    ])
    live = prop.model.new_point(400, rng=np.random.default_rng(seed))
    for n in prop.names:  # concentrate the live points so that the flow's unit-scale output is inside the prior
        live[n] = 0.4 * live[n]
    live["x_2"] = 0.1 * live["x_2"]
    prop.check_state(live)
    oprop.update_rescaling(live)
    return prop, oprop, live


def test_backward_pass_with_reparameterisation_table_matches_oracle():
    """rescale-to-bounds / logit / null / z-score columns in the draw kernel's epilogue
    (rescale.py:673-700, utils/rescaling.py:318-337) against the CPU restatement."""
    prop, oprop, live = build_with_reparameterisations()
    assert prop._rp_kind_dev is not None
    z = np.random.default_rng(0).normal(size=(6000, 6)).astype(np.float32).astype(np.float64) * 1.2
    x, log_q, z_kept = prop.backward_pass(z, return_z=True)
    x_o, log_q_o, z_o = oprop.backward_pass(z)
    assert len(x_o) > 500
    assert len(x) == len(x_o) and np.array_equal(z_kept, z_o)
    for n in prop.names:
        np.testing.assert_allclose(x[n], x_o[n], rtol=0, atol=1e-3 if n in ("x_0", "x_3") else 2e-4)
    np.testing.assert_allclose(log_q, log_q_o, rtol=1e-5, atol=2e-4)
    # host rescale is the reference arithmetic: forward_pass agrees with the oracle too
    zf, lp = prop.forward_pass(live[:200])
    zf_o, lp_o = oprop.forward_pass(live[:200])
    np.testing.assert_allclose(zf, zf_o, rtol=0, atol=2e-4)
    np.testing.assert_allclose(lp, lp_o, rtol=1e-5, atol=2e-4)


def test_backward_pass_with_boundary_inversion_matches_oracle():
    """Boundary inversion (registry entries ``inversion`` / ``inversion-duplicate``, rescale.py:541-612): the
    mirrored forward map runs on the host as in the reference (edges detected on the live points, half of
    the rows flipped or every row doubled), its inverse - fold |x'|, measure from the detected edge - in the draw
    kernel's epilogue (NSA_RP_INVERT / NSA_RP_MINUS_ONE_ONE), against the CPU restatement (itself bit-exact
    against the reference's classes, tests/test_reparam.py)."""
    from oracle import reparam as OR
    from nessai_amd.reparam import RP_INVERT, RP_MINUS_ONE_ONE

    config = {"x_0": "inversion", "x_3": "inversion-duplicate", "x_4": "inversion"}
    prop, oprop = build(6, "nsf", dict(n_neurons=16, n_blocks=3), reparameterisations=config)
    b = {n: [-5.0, 5.0] for n in prop.names}
    oprop.reparam = OR.OracleCombined([
        OR.OracleBoundaryInversion(["x_0"], b, "split", rng=np.random.default_rng(3)),
        OR.OracleBoundaryInversion(["x_3"], b, "duplicate"),
        OR.OracleBoundaryInversion(["x_4"], b, "split", rng=np.random.default_rng(3)),
        OR.OracleScaleAndShift(["x_1", "x_2", "x_5"], estimate_scale=True, estimate_shift=True),
    ])
    g = np.random.default_rng(4)
    live = prop.model.new_point(600, rng=g)
    live["x_0"] = -5.0 + np.abs(g.normal(0.0, 0.4, 600))   # piled up at the lower bound
    live["x_3"] = 5.0 - np.abs(g.normal(0.0, 0.5, 600))    # at the upper bound
    live["x_4"] = g.normal(0.5, 0.2, 600)                   # in the middle: no edge
    for n in ("x_1", "x_2", "x_5"):
        live[n] = 0.3 * live[n]
    prop.check_state(live)
    oprop.update_rescaling(live)
    edges = {p: e for grp in prop._reparam.groups if grp.kind == "to_bounds" for p, e in grp.edges.items()}
    assert edges == {"x_0": "lower", "x_3": "upper", "x_4": False}
    assert prop._rp_kind_dev.tolist() == [RP_INVERT, 0, 0, RP_INVERT, RP_MINUS_ONE_ONE, 0]
    z = g.normal(size=(6000, 6)).astype(np.float32).astype(np.float64) * 0.9
    x, log_q, z_kept = prop.backward_pass(z, return_z=True)
    x_o, log_q_o, z_o = oprop.backward_pass(z)
    assert len(x_o) > 1000
    assert len(x) == len(x_o) and np.array_equal(z_kept, z_o)
    for n in prop.names:
        np.testing.assert_allclose(x[n], x_o[n], rtol=0, atol=2e-4)
    np.testing.assert_allclose(log_q, log_q_o, rtol=1e-5, atol=2e-4)
    assert (x["x_0"] >= -5.0).all() and (x["x_3"] <= 5.0).all()  # folded back inside the bounds
    # the radius of the latent contour uses every mirror image: 4 latents per point here (two edges found)
    zr, _ = prop.forward_pass(live[:50], compute_radius=True)
    zr_o, _ = oprop.forward_pass(live[:50], compute_radius=True)
    assert zr.shape == (200, 6)
    np.testing.assert_allclose(zr, zr_o, rtol=0, atol=3e-4)
    # training data: the duplicating parameter doubles the rows the flow is fitted to
    x_prime, _ = prop.rescale(live)
    assert x_prime.size == 1200 and (x_prime["x_3_prime"][600:] == -x_prime["x_3_prime"][:600]).all()
    assert (x_prime["x_0_prime"] < 0).sum() == 600  # split: half of the (doubled) rows flipped


def test_saturated_sigmoid_rows_are_kept_with_infinite_log_q():
    """sigmoid(x') == 1.0 in float64: log|J| = -inf, log_q = +inf; the reference keeps the row
    (in bounds, never accepted) - flowproposal.py:366-389.  Every parameter uses the logit map so
    that the saturated rows cannot leave the prior bounds through another column."""
    from oracle import reparam as OR

    prop, oprop = build(6, "nsf", dict(n_neurons=16, n_blocks=3), reparameterisations="logit")
    names = list(prop.names)
    oprop.reparam = OR.OracleCombined([
        OR.OracleRescaleToBounds(names, {n: [-5.0, 5.0] for n in names}, rescale_bounds=[0.0, 1.0],
                                 update_bounds=False, post_rescaling="logit"),
    ])
    z = np.random.default_rng(1).normal(size=(200, 6)).astype(np.float32).astype(np.float64) * 0.3
    z[:7, 2] = 40.0  # linear tails: the value passes through the flow unchanged
    x, log_q, z_kept = prop.backward_pass(z, return_z=True)
    x_o, log_q_o, z_o = oprop.backward_pass(z)
    assert len(x_o) == 200 and np.array_equal(z_kept, z_o)
    sat_o = np.isposinf(log_q_o)
    assert sat_o.sum() == 7
    assert np.array_equal(np.isposinf(log_q), sat_o)
    for n in names:
        np.testing.assert_allclose(x[n], x_o[n], rtol=0, atol=2e-3)
    np.testing.assert_allclose(log_q[~sat_o], log_q_o[~sat_o], rtol=1e-5, atol=2e-4)
    log_w = prop.compute_weights(x, log_q)  # such rows are proposed but can never be accepted
    assert np.all(np.isneginf(log_w[sat_o]))


def test_populate_with_reparameterisation_table_matches_oracle():
    prop, oprop, _ = build_with_reparameterisations(poolsize=200, drawsize=3000)
    g = np.random.default_rng(12)
    batches = [(g.normal(size=(3000, 6)).astype(np.float32).astype(np.float64), g.random(3000)) for _ in range(60)]
    radius = OP.compute_radius(6, 0.95)
    used = []

    class Rng:
        def __init__(self):
            self.k = 0

        def random(self, n):
            z, u = batches[self.k]
            kept = used[self.k]
            self.k += 1
            return u[kept]

        def permutation(self, n):
            return np.arange(n)

    orig_backward = oprop.backward_pass

    def tracking_backward(z_trunc):
        x, lq, zk = orig_backward(z_trunc)
        zb = batches[len(used)][0]
        lookup = {row.tobytes(): i for i, row in enumerate(zb)}
        used.append(np.array([lookup[row.tobytes()] for row in zk], dtype=np.int64))
        return x, lq, zk

    oprop.backward_pass = tracking_backward
    oprop.injected_latents = (z for z, _ in batches)
    oprop.rng = Rng()
    oprop.populate(n_samples=200, r=radius)
    prop.populate(None, n_samples=200, r=radius, injected=iter(batches))
    assert prop.n_proposed == oprop.n_proposed
    assert abs(prop.n_accepted - oprop.n_accepted) <= max(2, int(0.01 * oprop.n_accepted))
    if prop.n_accepted == oprop.n_accepted:
        a = np.stack([prop.x[n] for n in prop.names], axis=1)
        b = np.stack([oprop.x[n] for n in oprop.names], axis=1)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-3)


def test_device_latents_match_numpy_philox():
    """The in-kernel Philox4x32-10 + Box-Muller stream equals its numpy restatement."""
    prop, _ = build(7, "nsf", dict(n_neurons=16, n_blocks=2))
    n = 3000
    z_out = torch.empty((n, 7), dtype=torch.float32, device=DEV)
    x64 = torch.empty((n, 7), dtype=torch.float64, device=DEV)
    log_q = torch.empty(n, dtype=torch.float64, device=DEV)
    prop._launch_draw(n, None, 12345, 1e9, x64, log_q, z_out)
    z_ref = philox.latents(1451, 12345, n, 7)
    np.testing.assert_allclose(z_out.cpu().numpy(), z_ref, rtol=0, atol=3e-6)
    z = z_out.cpu().numpy().ravel()
    assert stats.kstest(z, "norm").pvalue > 1e-3


def test_populate_device_rng_statistics():
    """With on-device draws: acceptance rate and accepted distribution agree with
    the oracle fed the SAME latents/uniforms through the numpy Philox restatement."""
    state, shift, scale, _ = load_c3_fixture()
    prop, oprop = build(32, "nsf", dict(n_neurons=64, n_blocks=6), poolsize=120, drawsize=20000,
                        state=state, shift=shift, scale=scale)
    prop.populate(None, n_samples=120, max_samples=400_000)  # ~6 passes: the oracle replays them on the CPU
    assert prop.populated and prop.samples.size > 0
    assert 0 < prop.population_acceptance < 1
    # oracle on the identical stream
    n_pass = prop.n_proposed // 20000
    zs = (philox.latents(1451, k * 20000, 20000, 32).astype(np.float64) for k in range(n_pass))
    oprop.injected_latents = zs
    acc = 0
    radius = OP.compute_radius(32, 0.95)
    for k in range(n_pass):
        z = oprop.sample_latent_distribution(20000)
        keep_idx = np.where(np.sqrt(np.sum(z**2, axis=-1)) <= radius)[0]
        x, lq, zk = oprop.backward_pass(z[keep_idx])
        lookup = {row.tobytes(): i for i, row in enumerate(z)}
        idx = np.array([lookup[r.tobytes()] for r in zk], dtype=np.int64)
        lw = oprop.compute_weights(x, lq)
        lw -= lw.max()
        u = philox.uniforms(1451, k * 20000, 20000)[idx]
        acc += int((lw > np.log(u)).sum())
    assert abs(acc - prop.n_accepted) <= max(3, int(0.02 * acc))


@pytest.mark.timeout(900)
def test_accepted_set_gate_c3_independent_streams():
    """SURVEY.md section 8d parity gate on the accepted SET: config-3 flow (trained weights), 2000
    accepted points from the device (its own Philox stream) against 2000 from the oracle on an
    independent stream (torch generator, numpy uniforms): acceptance rates within 3 sigma
    (binomial) and the per-dimension two-sample Kolmogorov-Smirnov test - family-wise 1 %, i.e.
    p > 0.01 / 32 in every dimension.  Draw batches of 2e4 rows keep the CPU side to ~2e6 rows."""
    from scipy.stats import ks_2samp

    state, shift, scale, _ = load_c3_fixture()
    prop, oprop = build(32, "nsf", dict(n_neurons=64, n_blocks=6), poolsize=2000, drawsize=20000,
                        state=state, shift=shift, scale=scale)
    prop.populate(None, n_samples=2000, max_samples=10**9)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    try:
        torch.manual_seed(20260930)
        oprop.rng = np.random.default_rng(20260930)
        oprop.populate(n_samples=2000, max_samples=10**9)
    finally:
        torch.set_num_threads(threads)
    assert prop.samples.size == oprop.x.size == 2000
    p_dev, p_orc = prop.n_accepted / prop.n_proposed, oprop.n_accepted / oprop.n_proposed
    sigma = np.sqrt(p_orc * (1 - p_orc) * (1 / prop.n_proposed + 1 / oprop.n_proposed))
    assert abs(p_dev - p_orc) <= 3 * sigma, (p_dev, p_orc, sigma)
    p_values = [ks_2samp(prop.samples[n], oprop.x[n]).pvalue for n in prop.names]
    assert min(p_values) > 0.01 / 32, sorted(p_values)[:4]


def test_populate_all_truncated_hits_max_samples():
    """Every draw outside the radius: loop exits on max_samples with an empty pool
    (flowproposal.py:434-438)."""
    prop, _ = build(4, "nsf", dict(n_neurons=8, n_blocks=2), poolsize=50, drawsize=1000)
    prop.populate(None, n_samples=50, r=1e-6, max_samples=3000)
    assert prop.samples.size == 0 and prop.n_accepted == 0
    assert prop.n_proposed == 4000


def test_populate_host_prior_matches_device_prior():
    """The general (host-evaluated) prior path gives the same pool as the constant
    device prior for a uniform model."""
    p1, _ = build(5, "nsf", dict(n_neurons=16, n_blocks=2), poolsize=200, drawsize=5000, prior="device")
    p2, _ = build(5, "nsf", dict(n_neurons=16, n_blocks=2), poolsize=200, drawsize=5000, prior="host")
    p1.populate(None, n_samples=200)
    p2.populate(None, n_samples=200)
    assert p1.n_accepted == p2.n_accepted and p1.n_proposed == p2.n_proposed
    for n in p1.names:
        np.testing.assert_array_equal(p1.x[n], p2.x[n])


def test_device_separable_prior_matches_host_prior():
    """Config-5 style priors (sine / cosine / power law / uniform) evaluated in the fused kernel
    give the same pool as the user's numpy log_prior on the host."""
    from nessai_amd.model import GWSurrogateModel

    pools = []
    for mode in ("device", "host"):
        model = GWSurrogateModel()
        rng = np.random.default_rng(4)
        torch.manual_seed(4)
        prop = FlowProposal(model, rng=rng, flow_config=dict(ftype="nsf", n_neurons=30, n_blocks=4),
                            poolsize=300, drawsize=20000, device_seed=9, prior=mode, precision="f32")
        prop.initialise()
        torch.manual_seed(4)
        prop.flow.model.load_state_dict(
            {k: v for k, v in _fresh_state(prop.flow.model).items()})
        live = model.new_point(500, rng=np.random.default_rng(5))
        prop.check_state(live)
        prop.populate(None, n_samples=300, max_samples=2_000_000)
        pools.append(prop)
    dev, host = pools
    assert dev.prior_mode == "device" and host.prior_mode == "host"
    assert dev.n_proposed == host.n_proposed and abs(dev.n_accepted - host.n_accepted) <= 1
    if dev.n_accepted == host.n_accepted:
        for n in dev.names:
            np.testing.assert_array_equal(dev.x[n], host.x[n])
    np.testing.assert_allclose(dev.samples["logP"], dev.model.log_prior(dev.samples), rtol=1e-12)
    assert dev.model.in_bounds(dev.samples).all()


def _fresh_state(flow):
    torch.manual_seed(123)
    from nessai_amd.flows import NeuralSplineFlow

    ref = NeuralSplineFlow(15, 30, 4, 2)
    return ref.state_dict()


def test_populate_accumulate_weights():
    """accumulate_weights=True (flowproposal.py:471-490, 504-510): pool of n_samples
    drawn from all accumulated candidates."""
    prop, oprop = build(4, "nsf", dict(n_neurons=8, n_blocks=2), poolsize=100, drawsize=2000,
                        accumulate_weights=True)
    prop.populate(None, n_samples=100, max_samples=200_000)
    assert prop.populated
    assert 0 < prop.samples.size <= 100
    assert prop.model.in_bounds(prop.samples).all()


def test_draw_pops_until_empty():
    """draw() repopulates when the pool is empty and pops one row per call
    (proposal/flowproposal/base.py:1152-1183)."""
    prop, _ = build(4, "nsf", dict(n_neurons=8, n_blocks=2), poolsize=20, drawsize=2000)
    prop.update_poolsize = False
    first = prop.draw(None)
    assert first.dtype.names[-3:] == ("logP", "logL", "it")
    n_left = len(prop.indices)
    for _ in range(n_left):
        prop.draw(None)
    assert not prop.populated


def test_adaptive_radius_uses_forward_pass():
    """radius_mode='adaptive': r = |z(worst point)| (truncation.py:343-351)."""
    prop, oprop = build(4, "nsf", dict(n_neurons=8, n_blocks=2), poolsize=20, drawsize=2000,
                        radius_mode="adaptive")
    from nessai_amd.livepoint import numpy_array_to_live_points

    worst = numpy_array_to_live_points(np.array([[0.4, -0.3, 0.2, 0.1]]), prop.names)
    r = prop.compute_population_radius(worst)
    oworst = OP.empty_structured_array(1, oprop.names)
    for i, n in enumerate(oprop.names):
        oworst[n] = [0.4, -0.3, 0.2, 0.1][i]
    z_o, _ = oprop.forward_pass(oworst)
    assert abs(r - float(np.sqrt(np.sum(z_o**2)))) < 1e-4


def test_full_size_c3_populate():
    """BASELINE config 3 at full size: drawsize 1e6, n_samples 2000, realistic weights."""
    state, shift, scale, _ = load_c3_fixture()
    prop, _ = build(32, "nsf", dict(n_neurons=64, n_blocks=6), poolsize=2000, drawsize=1_000_000,
                    state=state, shift=shift, scale=scale)
    prop.populate(None, n_samples=2000, max_samples=10**9)
    assert prop.samples.size == 2000
    assert prop.model.in_bounds(prop.samples).all()
    assert np.isfinite(prop.samples["logP"]).all() and np.isfinite(prop.samples["logL"]).all()
    assert prop.n_accepted >= 2000 and prop.n_proposed % 1_000_000 == 0


def test_latent_temperature_matches_oracle():
    """Tempered latents (base.py:393-414): injected z through backward_pass, and the on-device draw
    is sqrt(T) times the Philox normals with the density evaluated at the untempered values."""
    T = 0.7
    prop, oprop = build(6, "nsf", dict(n_neurons=16, n_blocks=3), latent_temperature=T)
    oprop.latent_temperature = T
    z = np.random.default_rng(0).normal(size=(4000, 6)).astype(np.float32).astype(np.float64)
    x, log_q, z_kept = prop.backward_pass(z, return_z=True)
    x_o, log_q_o, z_o = oprop.backward_pass(z)
    assert len(x) == len(x_o) and np.array_equal(z_kept, z_o)
    np.testing.assert_allclose(log_q, log_q_o, rtol=1e-5, atol=1e-4)
    # device draw: z_out = f32(sqrt(T) * z0), log_q consistent with the oracle fed those latents
    n, dev = 2048, prop.device
    x64 = torch.empty((n, 6), dtype=torch.float64, device=dev)
    lq = torch.empty(n, dtype=torch.float64, device=dev)
    z_out = torch.empty((n, 6), dtype=torch.float32, device=dev)
    prop._launch_draw(n, None, 100, np.inf, x64, lq, z_out)
    z0 = philox.latents(prop.device_seed, 100, n, 6)
    expected = (z0.astype(np.float64) * np.sqrt(T)).astype(np.float32)
    z_dev = z_out.cpu().numpy()
    np.testing.assert_allclose(z_dev, expected, rtol=0, atol=3e-6)  # Box-Muller differs by ulps
    # feed the oracle the device's own tempered latents; the reference keeps sqrt(T) z0 in float64
    x_d, lq_d, zk = oprop.backward_pass(z_dev.astype(np.float64))
    lq_h = lq.cpu().numpy()
    keep = ~np.isnan(lq_h)
    assert keep.sum() == len(lq_d)
    np.testing.assert_allclose(lq_h[keep], lq_d, rtol=1e-5, atol=1e-4)


def test_latent_temperature_validation():
    with pytest.raises(ValueError):
        FlowProposal(RosenbrockModel(2), poolsize=10, latent_temperature=0.0)
    with pytest.raises(TypeError):
        FlowProposal(RosenbrockModel(2), poolsize=10, latent_temperature=True)


@pytest.mark.parametrize("which", ["rosenbrock32", "gaussian16", "gw15", "rosenbrock5"])
def test_device_likelihood_equals_host_likelihood(which):
    """nsa_log_likelihood on the accepted rows == Model.log_likelihood on the host, bit for bit
    (float64, numpy's reduction order) - SURVEY.md §8f rank 4, flowproposal.py:521."""
    from nessai_amd.model import GWSurrogateModel

    model = {"rosenbrock32": lambda: RosenbrockModel(32), "rosenbrock5": lambda: RosenbrockModel(5),
             "gaussian16": lambda: GaussianModel(16), "gw15": GWSurrogateModel}[which]()
    props = {}
    for mode in ("device", "host"):
        prop = FlowProposal(model, rng=np.random.default_rng(5), poolsize=400, drawsize=20000, device_seed=9,
                            flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2), likelihood=mode)
        torch.manual_seed(0)  # same flow weights in both modes
        prop.initialise()
        live = model.new_point(500, rng=np.random.default_rng(1))
        prop.check_state(live)
        prop.populate(None, n_samples=400, max_samples=10**8)
        props[mode] = prop
    dev, host = props["device"], props["host"]
    assert dev._ll_spec is not None and host._ll_spec is None
    for n in dev.names:
        assert np.array_equal(dev.samples[n], host.samples[n])
    assert np.array_equal(dev.samples["logL"], host.samples["logL"])
    assert np.array_equal(dev.samples["logL"], model.log_likelihood(dev.samples))


@pytest.mark.parametrize("case", ["c3_fixture", "small_batches", "max_samples", "separable_prior",
                                  "small_batches_pass_by_pass", "separable_prior_pass_by_pass"])
def test_pass_trains_equal_one_pass_at_a_time(case):
    """nsa_populate_passes (loop control on the device, several passes per host synchronisation) and, for
    passes of <= 16384 rows, nsa_populate_passes_small (a whole train as one draw launch + two tail launches;
    the ..._pass_by_pass cases switch it off) reproduce the one-pass-per-synchronisation loop exactly: rows,
    order, counters, Philox use."""
    from nessai_amd.model import GWSurrogateModel

    small = not case.endswith("_pass_by_pass")
    case = case.replace("_pass_by_pass", "")
    results = {}
    for trains in (True, False):
        torch.manual_seed(0)
        if case == "c3_fixture":
            state, shift, scale, _ = load_c3_fixture()
            prop = FlowProposal(RosenbrockModel(32), rng=np.random.default_rng(3), poolsize=150, drawsize=200_000,
                                flow_config=dict(ftype="nsf", n_neurons=64, n_blocks=6, n_layers=2, num_bins=8),
                                device_seed=5, pass_trains=trains)
            prop.initialise()
            prop.flow.model.load_state_dict(state)
            prop.flow.model.eval()
            prop.set_rescaling_constants(shift, scale)
            calls = [dict(n_samples=150, max_samples=10**9)] * 2
        elif case == "separable_prior":
            model = GWSurrogateModel()
            prop = FlowProposal(model, rng=np.random.default_rng(3), poolsize=300, drawsize=5000, device_seed=5,
                                flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2), pass_trains=trains,
                                small_trains=small)
            prop.initialise()
            prop.check_state(model.new_point(500, rng=np.random.default_rng(1)))
            calls = [dict(n_samples=300, max_samples=10**8)] * 2
        else:
            prop = FlowProposal(GaussianModel(4), rng=np.random.default_rng(3), poolsize=500, drawsize=700,
                                flow_config=dict(ftype="realnvp", n_neurons=8, n_blocks=2), device_seed=5,
                                pass_trains=trains, small_trains=small)
            prop.initialise()
            prop.check_state(GaussianModel(4).new_point(300, rng=np.random.default_rng(1)))
            calls = ([dict(n_samples=500, max_samples=10**8)] * 3 if case == "small_batches"
                     else [dict(n_samples=5000, max_samples=3000)])
        out = []
        for kw in calls:
            prop.populate(None, **kw)
            out.append((np.stack([prop.x[n] for n in prop.names], axis=1), prop.n_accepted, prop.n_proposed,
                        prop._draw_counter, prop.samples["logL"].copy()))
        results[trains] = out
    for a, b in zip(results[True], results[False]):
        assert a[1:4] == b[1:4]
        assert np.array_equal(a[0], b[0])
        assert np.array_equal(a[4], b[4])
    if case == "max_samples":
        assert results[True][0][2] > 3000 and results[True][0][1] < 5000  # stopped by max_samples


def test_launch_geometry_is_tuned_once_and_never_changes_the_result(monkeypatch):
    """Big split-bf16 batches have three launch geometries (csrc/nsa_api.hip): a static rule picks one
    (nsa_big_batch_geometry), NESSAI_AMD_TUNE=1 times them once per process instead
    (nsa_tuned_occupancy); the choice is reported through nsa_last_geometry / nsa_last_row_tiles, and
    the rows drawn do not depend on it (row-indexed Philox streams, per-row arithmetic) - including
    the geometry in which every wave carries two 32-row tiles."""
    import ctypes

    from nessai_amd import lib

    def geometry():
        v = [ctypes.c_int(0) for _ in range(3)]
        lib.load().nsa_last_geometry(*[ctypes.addressof(x) for x in v])
        return tuple(x.value for x in v[:2]) + (int(lib.load().nsa_last_row_tiles()),)

    state, shift, scale, _ = load_c3_fixture()
    n = 300_000  # >= 8 tiles per CU: the range in which the three geometries are candidates
    prop, _ = build(32, "nsf", dict(n_neurons=64, n_blocks=6), poolsize=2000, drawsize=n,
                    state=state, shift=shift, scale=scale)
    prop.populate(None, n_samples=50, max_samples=10**9)  # rescaling constants, radius, device tables
    radius = prop.fixed_radius if prop.fixed_radius is not None else 1.0e9
    buf = prop._get_buffers(n, 2000)
    outs = {}
    for occ in ("3", "2", "2x2", "static", None, None):
        monkeypatch.delenv("NESSAI_AMD_TUNE", raising=False)
        if occ is None:
            monkeypatch.delenv("NESSAI_AMD_OCC", raising=False)
            monkeypatch.setenv("NESSAI_AMD_TUNE", "1")
        elif occ == "static":
            monkeypatch.delenv("NESSAI_AMD_OCC", raising=False)
        else:
            monkeypatch.setenv("NESSAI_AMD_OCC", occ)
        buf["x64"].zero_()
        buf["log_q"].zero_()
        prop._launch_draw(n, None, 0, radius, buf["x64"], buf["log_q"], None)
        torch.cuda.synchronize()
        outs.setdefault(occ, []).append((geometry(), buf["x64"][:n].clone(), buf["log_q"][:n].clone()))
    g3, x3, q3 = outs["3"][0]
    g2, x2, q2 = outs["2"][0]
    g22, x22, q22 = outs["2x2"][0]
    assert g3 == (3, 12, 1) and g2 == (2, 4, 1) and g22 == (2, 8, 2)
    gs, xs, qs = outs["static"][0]
    # 9375 tiles on 256 CUs: 5 rounds of 256 x 8 tiles at 1.09 (11.2 k) are estimated below 3 rounds of
    # 256 x 16 at 1.00 (12.3 k) and 4 rounds of 256 x 12 at 1.06 (13.0 k); at 1e6 rows it is two tiles per wave
    assert gs == (2, 4, 1)
    (ga, xa, qa), (gb, xb, qb) = outs[None]
    assert ga in ((3, 12, 1), (2, 4, 1), (2, 8, 2)) and gb == ga  # tuned once, kept
    for x, q in ((x2, q2), (x22, q22), (xs, qs), (xa, qa), (xb, qb)):
        assert torch.equal(x, x3)
        assert torch.equal(torch.nan_to_num(q, nan=12345.0), torch.nan_to_num(q3, nan=12345.0))


@pytest.mark.parametrize("drawsize", [2000, 40000])
def test_a_population_discarded_for_the_f16_range_has_no_side_effects(drawsize, monkeypatch):
    """A population whose draw kernel reports an operand beyond the f16 range is drawn again in float32.
    The discarded one must leave nothing behind but its Philox offsets: no likelihood call on its rows, no
    permutation taken from the host generator, no counters (both device-side loops: the small-passes one and
    the big-batch one).  The flag is forced here; the kernel's own report is covered in test_gpu_flows."""
    def make():
        prop, _ = build(6, "nsf", dict(n_neurons=16, n_blocks=2), model_cls=GaussianModel, poolsize=300,
                        drawsize=drawsize)
        prop._ll_spec = None  # host likelihood: every call is visible
        calls = []
        orig = prop.model.batch_evaluate_log_likelihood
        monkeypatch.setattr(prop.model, "batch_evaluate_log_likelihood", lambda x: (calls.append(len(x)), orig(x))[1])
        return prop, calls

    twin, twin_calls = make()
    twin.populate(None, n_samples=300, max_samples=10**9)

    prop, calls = make()
    real = prop._f16_range_left
    forced = [True]

    def flag(flag_host=None):
        if forced[0]:
            forced[0] = False
            return True
        return real(flag_host)

    monkeypatch.setattr(prop, "_f16_range_left", flag)
    monkeypatch.setattr(prop.flow.model, "note_f16_overflow", lambda: None)
    prop.populate(None, n_samples=300, max_samples=10**9)
    assert prop.populated and prop.populated_count == 1 and prop.samples.size == 300
    assert len(calls) == 1 and len(twin_calls) == 1            # one likelihood call, on the kept population
    assert prop.indices == twin.indices                         # one permutation from the host generator
    assert prop._draw_counter > twin._draw_counter              # the discarded draws are not drawn again
