"""The launch geometry in which every wave carries TWO 32-row tiles through the flow
(csrc/nsa_flow_kernel.h, template parameter RT; csrc/nsa_api.hip, NSA_GEO_2X2) must give the rows of
the one-tile geometries bit for bit: same per-row arithmetic, same order, only the tiles share the
weight operands a wave reads from LDS.  Checked per transform family / spline bin count / conditioner
width / direction, at ragged sizes (a wave whose second tile is partly or wholly past the end, a single
row) - the library itself uses the geometry for big split-bf16 batches only, so the small cases pin it
through NESSAI_AMD_OCC (runs on a real MI355X)."""
import ctypes

import pytest
import torch

from helpers import DEV, make_pair
from nessai_amd import lib
from test_gpu_flows import CONFIGS

pytestmark = pytest.mark.gpu

# <= 64-wide conditioners only (the 128-wide ones keep one tile per wave)
NAMES = ["nsf_c3", "nsf_c5_d15", "nsf_d2", "nsf_d5_k10_bn", "nsf_k4_lu_tanh", "nsf_k16", "rnvp_c2",
         "rnvp_c2_mlp", "rnvp_additive_actnorm", "rnvp_silu_perm", "rnvp_pre_bn"]


def geometry():
    v = [ctypes.c_int(0) for _ in range(3)]
    lib.load().nsa_last_geometry(*[ctypes.addressof(x) for x in v])
    return v[0].value, v[1].value, int(lib.load().nsa_last_row_tiles())


@pytest.mark.parametrize("name", ["nsf_c3", "nsf_c5_d15", "rnvp_c2", "nsf_k4_lu_tanh"])
@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
def test_two_tiles_per_wave_against_the_oracle(name, precision, monkeypatch):
    """The geometry the benchmark runs, compared with the ORACLE directly (not only with the one-tile
    geometry): log_prob within the 1e-5 bar where the mode is inside its envelope."""
    from helpers import rel_err
    from test_gpu_flows import BF16X3_WITHIN_BAR, LOG_PROB_RTOL

    P, Oc, args, kw = CONFIGS[name]
    ref, flow = make_pair(P, Oc, args, kw)
    flow.precision = precision
    monkeypatch.setenv("NESSAI_AMD_OCC", "2x2")
    x = torch.randn(4000, args[0], generator=torch.Generator().manual_seed(9)) * 1.5
    with torch.no_grad():
        lp_ref = ref.log_prob(x)
        xi_ref, ldi_ref = ref.inverse(x)
        lp = flow.log_prob(x.to(DEV)).cpu()
        geo = geometry()
        xi, ldi = flow.inverse(x.to(DEV))
    assert geo[2] == 2
    if precision == "f16x3" or name in BF16X3_WITHIN_BAR:
        assert rel_err(lp, lp_ref, floor=1e-3) <= LOG_PROB_RTOL
        assert rel_err(xi.cpu(), xi_ref) <= 2e-4 and rel_err(ldi.cpu(), ldi_ref) <= 5e-4
    else:
        assert rel_err(lp, lp_ref) <= 2e-4


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("n", [1, 31, 33, 64, 65, 97, 1000, 4097])
def test_two_tiles_per_wave_equal_one_tile_per_wave(name, n, precision, monkeypatch):
    if precision == "f16x3" and n not in (1, 65, 4097):
        pytest.skip("split-f16 shares the code path: three sizes pin its instances")
    P, Oc, args, kw = CONFIGS[name]
    _, flow = make_pair(P, Oc, args, kw)
    flow.precision = precision
    x = (torch.randn(n, args[0], generator=torch.Generator().manual_seed(n)) * 1.2).to(DEV)
    out = {}
    for occ in ("2", "2x2"):
        monkeypatch.setenv("NESSAI_AMD_OCC", occ)
        with torch.no_grad():
            lp = flow.log_prob(x)
            z, ld = flow.forward(x)
            xi, ldi = flow.inverse(x)
        torch.cuda.synchronize()
        out[occ] = (geometry(), [t.clone() for t in (lp, z, ld, xi, ldi)])
    assert out["2"][0][2] == 1 and out["2x2"][0][2] == 2  # the second run really carried two tiles per wave
    for a, b in zip(out["2"][1], out["2x2"][1]):
        assert torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0))


def test_wide_conditioners_keep_one_tile_per_wave(monkeypatch):
    P, Oc, args, kw = CONFIGS["nsf_c4_d64"]
    _, flow = make_pair(P, Oc, args, kw)
    flow.precision = "bf16x3"
    monkeypatch.setenv("NESSAI_AMD_OCC", "2x2")
    with torch.no_grad():
        flow.log_prob(torch.randn(500, args[0]).to(DEV))
    torch.cuda.synchronize()
    assert geometry()[2] == 1


def test_exact_f32_keeps_one_tile_per_wave(monkeypatch):
    P, Oc, args, kw = CONFIGS["nsf_c3"]
    _, flow = make_pair(P, Oc, args, kw)
    flow.precision = "f32"
    monkeypatch.setenv("NESSAI_AMD_OCC", "2x2")
    with torch.no_grad():
        flow.log_prob(torch.randn(500, args[0]).to(DEV))
    torch.cuda.synchronize()
    assert geometry()[2] == 1
