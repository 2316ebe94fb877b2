#!/usr/bin/env python
"""The fused draw kernel of BASELINE config 2 (the reference's default flow: RealNVP, 4 couplings, 2 x 32
residual conditioner, LU-linear + batch norm; and its MLP-conditioner variant) in every launch geometry: the
staged ones (NESSAI_AMD_RESIDENT=0: stage copies + one barrier per stage) and the weights-resident candidates
(waves per SIMD x row tiles per wave).  One JSON line per (flow, geometry): average kernel time over 1e6 rows
(HIP events on the launch stream), rows/s, fraction of the HBM roof SURVEY.md section 8d assigns to the config,
and a checksum of the outputs (the geometries must agree bit for bit).  Run from the repo root on a GPU box."""
import ctypes
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

from bench_configs import kernel_roofline, live_points, time_draw  # noqa: E402
from nessai_amd import lib  # noqa: E402
from nessai_amd.model import GaussianModel  # noqa: E402
from nessai_amd.proposal import FlowProposal  # noqa: E402


def geometry():
    v = [ctypes.c_int(0) for _ in range(3)]
    L = lib.load()
    L.nsa_last_geometry(*[ctypes.addressof(x) for x in v])
    return dict(waves_per_simd=v[0].value, waves_per_group=v[1].value, groups=v[2].value,
                row_tiles=int(L.nsa_last_row_tiles()), resident=int(L.nsa_last_resident()))


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", nargs="*", type=int, default=[1_000_000])
    ap.add_argument("--geometry", action="append", help="NESSAI_AMD_RESIDENT value(s) to time; default: all + library choice")
    ap.add_argument("--net", action="append", help="resnet / mlp; default both")
    args = ap.parse_args()
    sizes = args.sizes
    geos = args.geometry or ["0", "4x1", "3x1", "2x2", "2x1", None]
    geos = [None if g in (None, "auto") else g for g in geos]
    model = GaussianModel(16)
    for net in args.net or ("resnet", "mlp"):
        rng = np.random.default_rng(1451)
        torch.manual_seed(1451)
        live = live_points(model, 2000, 80_000, rng)
        with tempfile.TemporaryDirectory() as tmp:
            prop = FlowProposal(model, rng=rng, poolsize=2000, drawsize=sizes[0], output=tmp, device_seed=1451,
                                flow_config=dict(ftype="realnvp", n_neurons=32, n_blocks=4, n_layers=2, net=net))
            prop.initialise()
            prop.train(live)
            prop.populate(None, n_samples=2000, max_samples=10**12)
            precision = prop.flow.model.effective_precision
            for n in sizes:
                for geo in geos:
                    if geo is None:
                        os.environ.pop("NESSAI_AMD_RESIDENT", None)
                    else:
                        os.environ["NESSAI_AMD_RESIDENT"] = geo
                    time_draw(prop, n, launches=20)  # (the clocks settle: the first launches of a geometry run slower)
                    ms = time_draw(prop, n, launches=40)
                    spread = dict(time_draw.last)
                    g = geometry()
                    buf = prop._get_buffers(n, prop.poolsize)
                    prop._launch_draw(n, None, 0, prop.r, buf["x64"], buf["log_q"], None, prepared=prop._prepare_flow())
                    torch.cuda.synchronize()
                    lq = buf["log_q"][:n]
                    ok = torch.isfinite(lq)
                    check = float(lq[ok].sum().item()) + float(buf["x64"][:n][ok].sum().item())
                    roof = kernel_roofline(prop.flow.model, precision, n, ms, 8 * model.dims + 8, "hbm")
                    print(json.dumps(dict(config=f"C2 16-D RealNVP net={net}", rows=n, geometry_env=geo or "library choice",
                                          geometry=g, kernel_ms=ms, kernel_ms_median=spread["median"], kernel_ms_min=spread["min"], rows_per_s=n / ms * 1e3, frac_hbm=roof["frac_hbm"],
                                          frac_mfma=roof["frac_mfma"], precision=precision, kept_rows=int(ok.sum().item()),
                                          checksum=check)), flush=True)


if __name__ == "__main__":
    main()
