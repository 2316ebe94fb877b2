"""The small-batch geometry (csrc/nsa_flow_kernel.h, ``NsaLaunch::team`` / ``ring`` / ``loader``; csrc/nsa_api.hip,
``nsa_launch_flow``): a batch with fewer 32-row tiles than the chip has CUs runs one workgroup per tile - ONE computing
wave, a loader wave that only requests the stage copies, a ring of up to eight stage buffers with counted waits
(``NsaRingPipe``).  Pinned options give a tile to a TEAM of 2 / 4 / 8 computing waves: each runs the conditioner's
trunk, the chunks of its last layer (parameters + element-wise transform) are dealt round the team, the Jacobian terms
meet in LDS and are added in the order a single wave adds them.  Whatever the team size, the ring's depth and whoever
requests the copies, the rows must be those of the staged one-wave-per-tile geometry BIT FOR BIT, for every transform
family, direction and mode, at ragged sizes.  The library takes the geometry by itself for conditioners of 64 features
and more whose program does not stay resident in LDS (config 4: ``ImportanceFlowProposal.draw`` / ``log_prob_all`` on a
few thousand rows, ``src/nessai/proposal/importance.py:443-526``; config 3's flow on the live points), the environment
pins it for the others (runs on a real MI355X)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import DEV, make_pair, rel_err
from nessai_amd import lib
from test_gpu_flows import CONFIGS, LOG_PROB_RTOL, LOGDET_ATOL, X_ATOL

pytestmark = pytest.mark.gpu

NAMES = ["nsf_c4_d64", "nsf_d96_h192", "rnvp_d100_lu", "nsf_c3", "nsf_c5_d15", "nsf_d2", "nsf_d5_k10_bn",
         "nsf_k4_lu_tanh", "nsf_k13_d32", "rnvp_c2", "rnvp_c2_mlp", "rnvp_additive_actnorm", "rnvp_svd"]


def geometry():
    L = lib.load()
    v = [ctypes.c_int(0) for _ in range(3)]
    L.nsa_last_geometry(*[ctypes.addressof(x) for x in v])
    return v[0].value, v[1].value, v[2].value, int(L.nsa_last_team())


def same(a, b):
    return torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0))


def run_all(flow, x):
    with torch.no_grad():
        lp = flow.log_prob(x)
        geo = geometry()
        z, ld = flow.forward(x)
        xi, ldi = flow.inverse(x)
    torch.cuda.synchronize()
    return geo, [t.clone() for t in (lp, z, ld, xi, ldi)]


@pytest.mark.parametrize("team", [1, 2, 4, 8])
@pytest.mark.parametrize("precision", ["f32", "f16x3", "bf16x3"])
@pytest.mark.parametrize("name", NAMES)
def test_team_rows_equal_single_wave_rows(name, precision, team, monkeypatch):
    if team not in (1, 4) and precision == "bf16x3":
        pytest.skip("split-bf16 shares the code path: two team sizes pin its instances")
    P, Oc, args, kw = CONFIGS[name]
    _, flow = make_pair(P, Oc, args, kw)
    flow.precision = precision
    monkeypatch.delenv("NESSAI_AMD_OCC", raising=False)
    for n in (1, 33, 1000):
        x = (torch.randn(n, args[0], generator=torch.Generator().manual_seed(n)) * 1.2).to(DEV)
        monkeypatch.setenv("NESSAI_AMD_TEAM", "0")
        monkeypatch.setenv("NESSAI_AMD_RESIDENT", "0")
        geo_s, single = run_all(flow, x)
        monkeypatch.setenv("NESSAI_AMD_TEAM", str(team))
        geo_t, shared = run_all(flow, x)
        if name == "rnvp_d100_lu" and precision != "f32" and geo_t[3] == 0:
            continue  # (its split-16-bit stages of 66 KiB leave no room for two buffers beside the team's record)
        if name in ("rnvp_d100_lu", "nsf_d96_h192") and team == 8 and geo_t[3] == 0:
            continue  # (256-wide instances are built for one wave per SIMD: workgroups of at most four waves)
        assert geo_s[3] == 0 and geo_t[3] == team and geo_t[1] == team, (geo_s, geo_t)
        for a, b in zip(single, shared):
            assert same(a, b)


@pytest.mark.parametrize("ring", [2, 3, 5, 8])
@pytest.mark.parametrize("name", ["nsf_c4_d64", "nsf_c3", "rnvp_c2", "nsf_d2"])
def test_ring_depth_never_changes_the_rows(name, ring, monkeypatch):
    """The stage ring of the team geometry (NsaRingPipe): 2 ... 8 buffers, copies requested ring - 1 stages ahead,
    counted waits - the rows are those of the double-buffered single-wave geometry whatever the depth (a flow with
    fewer stages than the ring has buffers included: nsf_d2)."""
    P, Oc, args, kw = CONFIGS[name]
    _, flow = make_pair(P, Oc, args, kw)
    flow.precision = "f16x3"
    monkeypatch.delenv("NESSAI_AMD_OCC", raising=False)
    monkeypatch.setenv("NESSAI_AMD_RESIDENT", "0")
    x = (torch.randn(3000, args[0], generator=torch.Generator().manual_seed(2)) * 1.2).to(DEV)
    monkeypatch.setenv("NESSAI_AMD_TEAM", "0")
    _, single = run_all(flow, x)
    for team, loader in ((1, "3"), (1, "1"), (1, "0"), (4, "2"), (4, "0")):
        monkeypatch.setenv("NESSAI_AMD_TEAM", str(team))
        monkeypatch.setenv("NESSAI_AMD_RING", str(ring))
        monkeypatch.setenv("NESSAI_AMD_LOADER", loader)  # loader waves; 0: every wave requests its share of the pieces
        geo, shared = run_all(flow, x)
        assert geo[3] == team and 2 <= int(lib.load().nsa_last_ring()) <= ring, geo
        for a, b in zip(single, shared):
            assert same(a, b)


@pytest.mark.parametrize("tiles", [2, 3, 5])
@pytest.mark.parametrize("name", ["nsf_c4_d64", "nsf_c3", "rnvp_c2", "nsf_d5_k10_bn"])
def test_tiles_sharing_a_workgroup_never_change_the_rows(name, tiles, monkeypatch):
    """Several tiles per workgroup of the loader geometry (``NESSAI_AMD_TILES``; the library's own choice for batches of
    one to five tiles per CU): every tile keeps its own computing wave(s), the loaders and the stage copies are shared -
    same rows bit for bit, ragged last workgroup included."""
    P, Oc, args, kw = CONFIGS[name]
    _, flow = make_pair(P, Oc, args, kw)
    flow.precision = "f16x3"
    monkeypatch.delenv("NESSAI_AMD_OCC", raising=False)
    monkeypatch.setenv("NESSAI_AMD_RESIDENT", "0")
    x = (torch.randn(2100, args[0], generator=torch.Generator().manual_seed(8)) * 1.2).to(DEV)  # 66 tiles
    monkeypatch.setenv("NESSAI_AMD_TEAM", "0")
    _, single = run_all(flow, x)
    for team in (1, 2):
        if team * tiles > 7:
            continue
        monkeypatch.setenv("NESSAI_AMD_TEAM", str(team))
        monkeypatch.setenv("NESSAI_AMD_TILES", str(tiles))
        geo, shared = run_all(flow, x)
        assert geo[3] == team and geo[2] == (66 + tiles - 1) // tiles, geo
        for a, b in zip(single, shared):
            assert same(a, b)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_team_is_the_default_for_small_batches_of_wide_flows_and_matches_the_oracle(precision, monkeypatch):
    for k in ("NESSAI_AMD_TEAM", "NESSAI_AMD_OCC", "NESSAI_AMD_RESIDENT", "NESSAI_AMD_TILES", "NESSAI_AMD_LOADER", "NESSAI_AMD_RING"):
        monkeypatch.delenv(k, raising=False)
    P, Oc, args, kw = CONFIGS["nsf_c4_d64"]
    ref, flow = make_pair(P, Oc, args, kw)
    flow.precision = precision
    x = torch.randn(5050, args[0], generator=torch.Generator().manual_seed(3)) * 1.3  # 158 tiles: config 4's draw
    with torch.no_grad():
        lp_ref = ref.log_prob(x)
        xi_ref, ldi_ref = ref.inverse(x)
    geo, (lp, z, ld, xi, ldi) = run_all(flow, x.to(DEV))
    assert geo[3] == 1 and geo[2] == 158 and int(lib.load().nsa_last_ring()) >= 4, geo  # one wave (+ loader) per tile
    assert rel_err(lp.cpu(), lp_ref, floor=1e-3) <= LOG_PROB_RTOL
    # (x and log|det J| of the inverse at the bars of tests/test_gpu_flows.py::test_forward_inverse_match_oracle)
    assert rel_err(xi.cpu(), xi_ref) <= X_ATOL and rel_err(ldi.cpu(), ldi_ref) <= LOGDET_ATOL
    # ... a mid-size batch shares the loaders among several tiles per workgroup: ONE round of workgroups over the chip
    mid = torch.randn(20000, args[0], generator=torch.Generator().manual_seed(5))  # 625 tiles -> 3 per workgroup
    with torch.no_grad():
        lp_mid_ref = ref.log_prob(mid)
    lp_mid = flow.log_prob(mid.to(DEV))
    geo = geometry()
    assert geo[3] == 1 and geo[2] == 209, geo
    assert rel_err(lp_mid.cpu(), lp_mid_ref, floor=1e-3) <= LOG_PROB_RTOL
    # ... and a batch that fills the chip several times over keeps the staged geometry
    big = torch.randn(100000, args[0], device=DEV)
    flow.log_prob(big)
    assert geometry()[3] == 0
    # flows whose program stays resident in LDS keep the resident geometry ...
    P, Oc, args, kw = CONFIGS["rnvp_c2"]
    _, small = make_pair(P, Oc, args, kw)
    small.log_prob(torch.randn(2000, args[0], device=DEV))
    assert geometry()[3] == 0 and int(lib.load().nsa_last_resident()) == 1
    # ... config 3's (135 KB of stages) does not: its live points take the loader geometry too
    P, Oc, args, kw = CONFIGS["nsf_c3"]
    ref3, c3 = make_pair(P, Oc, args, kw)
    c3.precision = precision
    x3 = torch.randn(2000, args[0], generator=torch.Generator().manual_seed(4))
    lp3 = c3.log_prob(x3.to(DEV))
    assert geometry()[3] == 1
    with torch.no_grad():
        assert rel_err(lp3.cpu(), ref3.log_prob(x3), floor=1e-3) <= LOG_PROB_RTOL


def test_team_populate_passes_give_the_same_pool(monkeypatch):
    """The populate mode through the team geometry: latent draw (Philox by global row), inverse flow, rescale, bounds,
    log_q - candidates and densities of a small pass equal those of the single-wave geometry bit for bit."""
    from nessai_amd.flows.base import HipFlow  # noqa: F401  (import check: the op library is loaded)
    from nessai_amd import ops  # noqa: F401

    P, Oc, args, kw = CONFIGS["nsf_c4_d64"]
    _, flow = make_pair(P, Oc, args, kw)
    flow.precision = "f16x3"
    out = {}
    for team in ("0", "1", "4"):
        monkeypatch.setenv("NESSAI_AMD_TEAM", team)
        torch.manual_seed(5)
        x, lp = flow.sample_and_log_prob(3000)
        out[team] = (geometry(), x.clone(), lp.clone())
    assert out["0"][0][3] == 0 and out["1"][0][3] == 1 and out["4"][0][3] == 4
    for team in ("1", "4"):
        assert same(out["0"][1], out[team][1]) and same(out["0"][2], out[team][2])


def test_team_reports_operands_beyond_the_f16_range_like_the_single_wave_geometry(monkeypatch):
    """An operand beyond the f16 range reaches the status word whichever member of the team meets it: a flow pinned
    to split-f16 raises in either geometry, and in float32 the two geometries give the same rows."""
    from nessai_amd.flows.base import F16RangeError

    P, Oc, args, kw = CONFIGS["nsf_c4_d64"]
    _, flow = make_pair(P, Oc, args, kw)
    x = torch.randn(500, args[0], device=DEV)
    x[17, 3] = 3.0e6
    res = {}
    for team in ("0", "4"):
        monkeypatch.setenv("NESSAI_AMD_TEAM", team)
        flow.precision = "f16x3"
        with pytest.raises(F16RangeError), torch.no_grad():
            flow.log_prob(x)
        flow.precision = "f32"
        with torch.no_grad():
            res[team] = flow.log_prob(x).clone()
    assert same(res["0"], res["4"])
    assert bool(torch.isfinite(res["4"][:17]).all())
