"""Lean populate passes (``nsa_populate_passes_lean`` + ``nsa_populate_replay``): a pass that keeps no candidate
rows and the replay of its accepted rows from their Philox counters give the pool of the full passes BIT FOR BIT
(runs on a real MI355X).

Reference semantics: only accepted rows are ever used, ``samples[...] = x[accept][:m]``
(``src/nessai/proposal/flowproposal/flowproposal.py:431-498``).
"""
import numpy as np
import pytest
import torch

from helpers import load_c3_fixture
from nessai_amd import lib
from nessai_amd.model import GaussianModel, RosenbrockModel
from nessai_amd.proposal import FlowProposal

pytestmark = pytest.mark.gpu


def _pool(prop):
    return (np.stack([prop.x[n] for n in prop.names], axis=1), prop.n_accepted, prop.n_proposed, prop._draw_counter,
            prop.samples["logL"].copy(), prop.samples["logP"].copy())


def _make(case, lean, precision="auto", **extra):
    from nessai_amd.model import GWSurrogateModel

    torch.manual_seed(0)
    if case == "c3":
        state, shift, scale, _ = load_c3_fixture()
        prop = FlowProposal(RosenbrockModel(32), rng=np.random.default_rng(3), poolsize=150, drawsize=300_000,
                            flow_config=dict(ftype="nsf", n_neurons=64, n_blocks=6, n_layers=2, num_bins=8),
                            device_seed=5, lean=lean, precision=precision, **extra)
        prop.initialise()
        prop.flow.model.load_state_dict(state)
        prop.flow.model.eval()
        prop.set_rescaling_constants(shift, scale)
        calls = [dict(n_samples=150, max_samples=10**9)] * 2
    elif case == "separable_prior":  # bilby-style priors on the device, reparameterisation table in the epilogue
        model = GWSurrogateModel()
        prop = FlowProposal(model, rng=np.random.default_rng(3), poolsize=300, drawsize=40_000, device_seed=5,
                            flow_config=dict(ftype="nsf", n_neurons=16, n_blocks=2), lean=lean, precision=precision,
                            **extra)
        prop.initialise()
        prop.check_state(model.new_point(500, rng=np.random.default_rng(1)))
        calls = [dict(n_samples=300, max_samples=10**8)] * 2
    elif case == "realnvp_lu":  # the reference's default flow: weights-resident geometry
        prop = FlowProposal(GaussianModel(16), rng=np.random.default_rng(3), poolsize=1000, drawsize=100_000,
                            flow_config=dict(ftype="realnvp", n_neurons=32, n_blocks=4), device_seed=5, lean=lean,
                            precision=precision, **extra)
        prop.initialise()
        prop.check_state(GaussianModel(16).new_point(2000, rng=np.random.default_rng(1)))
        calls = [dict(n_samples=1000, max_samples=10**8)] * 2
    elif case == "pool_exceeds_draw":  # more rows replayed than one pass draws (poolsize > drawsize > 16384): the
        # replay's per-row outputs must not be sized by the pass (round-5 advisor finding, proposal.py _get_buffers)
        prop = FlowProposal(GaussianModel(4), rng=np.random.default_rng(3), poolsize=30_000, drawsize=20_000,
                            flow_config=dict(ftype="realnvp", n_neurons=8, n_blocks=2), device_seed=5, lean=lean,
                            precision=precision, **extra)
        prop.initialise()
        prop.check_state(GaussianModel(4).new_point(300, rng=np.random.default_rng(1)))
        calls = [dict(n_samples=30_000, max_samples=10**9)] * 2
    else:  # max_samples exit inside a train
        prop = FlowProposal(GaussianModel(4), rng=np.random.default_rng(3), poolsize=500, drawsize=20_000,
                            flow_config=dict(ftype="realnvp", n_neurons=8, n_blocks=2), device_seed=5, lean=lean,
                            precision=precision, **extra)
        prop.initialise()
        prop.check_state(GaussianModel(4).new_point(300, rng=np.random.default_rng(1)))
        calls = [dict(n_samples=10**6, max_samples=90_000)]
    return prop, calls


@pytest.mark.parametrize("precision", ["auto", "f32"])
@pytest.mark.parametrize("case", ["c3", "separable_prior", "realnvp_lu", "max_samples", "pool_exceeds_draw"])
def test_lean_passes_and_replay_equal_the_full_passes(case, precision):
    results = {}
    for lean in (True, False):
        prop, calls = _make(case, lean, precision)
        out = []
        for kw in calls:
            prop.populate(None, **kw)
            assert prop.last_populate_path == "trains" and prop.last_populate_lean == lean
            out.append(_pool(prop))
        results[lean] = out
    for a, b in zip(results[True], results[False]):
        assert a[1:4] == b[1:4]
        assert a[0].shape == b[0].shape and a[0].shape[0] > 0
        for k in (0, 4, 5):
            assert np.array_equal(a[k], b[k])
    if case == "max_samples":
        assert results[True][0][2] > 90_000 and results[True][0][1] < 10**6
    if case == "pool_exceeds_draw":  # the replay really covered more rows than a pass has
        assert results[True][0][0].shape[0] == 30_000 > 20_000


def test_lean_with_the_min_log_q_rule_and_not_with_the_likelihood_rule():
    """``min_log_q`` is a predicate on log_q: it stays lean.  The likelihood threshold needs every candidate's
    coordinates: the proposal falls back to the full passes by itself (and the C ABI refuses)."""
    from test_gpu_populate import build

    def run(lean, **rules):
        prop, _ = build(6, "nsf", dict(n_neurons=16, n_blocks=3), model_cls=GaussianModel, poolsize=200,
                        drawsize=20000, lean=lean, **rules)
        live = prop.model.new_point(300, rng=np.random.default_rng(4))
        for n in prop.names:
            live[n] *= 0.1
        live["logL"] = prop.model.log_likelihood(live)
        prop.training_data = live
        worst = live[np.argmin(live["logL"])]
        prop.populate(worst, n_samples=200, max_samples=10**8)
        assert prop.last_populate_path == "trains"
        return prop

    a, b = run(True, truncate_log_q=True), run(False, truncate_log_q=True)
    assert a.last_populate_lean and not b.last_populate_lean
    assert np.isfinite(a.truncation.device_limits()[1])
    assert _pool(a)[1:4] == _pool(b)[1:4]
    for k in (0, 4, 5):
        assert np.array_equal(_pool(a)[k], _pool(b)[k])
    c = run(True, truncate_log_q=True, enforce_likelihood_threshold=True)
    assert not c.last_populate_lean


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_replay_reproduces_arbitrary_rows_of_a_full_pass(precision):
    """The C entry point on its own: any subset of rows (any order, repeats, rows of different tiles and lanes)
    comes back as the full pass wrote it - x and log_q, dropped rows (NaN log_q) included."""
    state, shift, scale, _ = load_c3_fixture()
    torch.manual_seed(0)
    prop = FlowProposal(RosenbrockModel(32), rng=np.random.default_rng(3), poolsize=100, drawsize=70_001,
                        flow_config=dict(ftype="nsf", n_neurons=64, n_blocks=6, n_layers=2, num_bins=8),
                        device_seed=77, precision=precision)
    prop.initialise()
    prop.flow.model.load_state_dict(state)
    prop.flow.model.eval()
    prop.set_rescaling_constants(shift, scale)
    prop.populate(None, n_samples=10, max_samples=10**9)  # device tables, radius
    n, offset = 70_001, 123_456_789_012
    buf = prop._get_buffers(n, 100)
    radius = 5.7  # about half of the rows fall outside: dropped rows must replay as dropped
    prop._launch_draw(n, None, offset, radius, buf["x64"], buf["log_q"], None)
    g = np.random.default_rng(0)
    pick = np.concatenate([g.integers(0, n, size=1000), [0, n - 1, 31, 32, 33], g.integers(0, n, size=3)])
    idx = torch.from_numpy(pick + offset).to(prop.device)
    m = len(pick)
    x = torch.full((m, 32), np.nan, dtype=torch.float64, device=prop.device)
    lq = torch.zeros(m, dtype=torch.float64, device=prop.device)
    prop._replay(idx, m, radius, x, lq, None, prop._prepare_flow())
    torch.cuda.synchronize()
    want_x = buf["x64"][torch.from_numpy(pick).to(prop.device)]
    want_q = buf["log_q"][torch.from_numpy(pick).to(prop.device)]
    assert torch.isnan(want_q).any() and not torch.isnan(want_q).all()
    assert torch.equal(x, want_x)
    assert torch.equal(torch.nan_to_num(lq, nan=1e300), torch.nan_to_num(want_q, nan=1e300))


def test_lean_entry_points_refuse_what_needs_the_candidates():
    L = lib.load()
    rules = lib.PostRules(-np.inf, 0, None, None, 0.0, 0.0)  # a likelihood rule
    hdr = (np.zeros(40, dtype=np.uint32)).ctypes.data
    rc = L.nsa_populate_passes_lean(hdr, 1, 100_000, 1, 1, 0, 1.0, 1.0, 1, 1, 1, 1, 0.0, 1, None, None, None, 0.0,
                                    None, None, 1, 1, 10, 1, 1, 1, 1 << 30, 1, 0, 0, None, rules)
    assert rc != 0 and b"likelihood" in L.nsa_last_error()
    rc = L.nsa_populate_replay(hdr, 1, 10, None, 1, 1.0, 1.0, 1, 1, 1, 1, 0.0, 1, 1, None, None, None, None, None, 1,
                               0, 0, None)
    assert rc != 0


@pytest.mark.parametrize("case", ["c3", "separable_prior"])
def test_draw_weights_equals_draw_then_weights(case):
    """The head of a lean pass (``nsa_populate_draw_weights``: weights and their maximum out of the draw kernel) against
    the full pass followed by ``nsa_populate_truncate`` and ``nsa_populate_weights``: every weight, the -inf of the
    dropped rows and the maximum, bit for bit - with and without the ``min_log_q`` rule."""
    prop, _ = _make(case, True)
    prop.populate(None, n_samples=10, max_samples=10**9)  # device tables, radius
    n = 50_017
    L = lib.load()
    buf = prop._get_buffers(n, 100)
    radius = 5.7 if case == "c3" else prop.r
    separable = prop._prior_kind_dev is not None
    for min_log_q in (-np.inf, None):
        prop._launch_draw(n, None, 999, radius, buf["x64"], buf["log_q"], None,
                          log_prior=buf["log_p"] if separable else None)
        if min_log_q is None:  # a threshold that drops about a third of the live rows
            lq = buf["log_q"][:n]
            min_log_q = float(torch.quantile(lq[torch.isfinite(lq)], 0.33))
        prop._truncate_after_flow(buf, n, min_log_q, None)
        prop._weights_and_max(buf, n)
        torch.cuda.synchronize()
        want_w, want_key = buf["log_w"][:n].clone(), int(buf["max_key"].item())
        log_w = torch.zeros(n, dtype=torch.float64, device=prop.device)
        key = torch.full((1,), int(L.nsa_max_key_init()), dtype=torch.int64, device=prop.device)
        prop._launch_draw_weights(n, 999, radius, log_w, key, log_prior=buf["log_p"], min_log_q=min_log_q)
        torch.cuda.synchronize()
        assert torch.isinf(want_w).any() and torch.isfinite(want_w).any()
        assert torch.equal(log_w, want_w)
        assert int(key.item()) == want_key
        assert float(L.nsa_max_key_decode(want_key)) == float(want_w.max())
