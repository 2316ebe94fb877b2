#!/usr/bin/env python
"""Per-section cycle budget of a SMALL batch of config 4's flow (64-D spline flow, 128-wide conditioner, 5050 rows =
158 tiles) from a -DNSA_TIMELINE build, per launch geometry (a wave per tile / teams of four / ring depths):

    cd nessai_amd/csrc && make -j8 BUILD=build_tl4 TARGET=../../variants/libnessai_amd_tl4.so \\
        HTS=4 TRANSFORMS=1_8 EXTRA=-DNSA_TIMELINE
    NESSAI_AMD_LIBRARY=variants/libnessai_amd_tl4.so python scripts/timeline_c4.py        # GPU box
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SLOTS = ["fill+prologue", "layer0 gemm", "layer0 tail", "hidden gemm", "hidden tail", "chunk gemm", "transform",
         "copy wait", "barrier", "epilogue", "other ops", "loop top", "copy issue", "bias read"]


def main(rows=5050):
    from nessai_amd.flows import NeuralSplineFlow

    torch.manual_seed(3)
    flow = NeuralSplineFlow(64, 128, 6, 2)
    flow.to("cuda:0")
    flow.eval()
    flow.precision = "f16x3"
    x = torch.randn(rows, 64, device="cuda:0")
    for team, ring in (("0", None), ("1", "2"), ("1", "8"), ("4", "2"), ("4", "8"), ("8", "8"), ("2", "8")):
        os.environ["NESSAI_AMD_TEAM"] = team
        if ring is None:
            os.environ.pop("NESSAI_AMD_RING", None)
        else:
            os.environ["NESSAI_AMD_RING"] = ring
        os.environ.pop("NSA_TL_PTR", None)
        for _ in range(3):
            flow.log_prob(x)
        torch.cuda.synchronize()
        ev = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            flow.log_prob(x)
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
        dbg = torch.zeros(16, dtype=torch.int64, device="cuda:0")
        os.environ["NSA_TL_PTR"] = str(dbg.data_ptr())
        flow.log_prob(x)
        torch.cuda.synchronize()
        os.environ.pop("NSA_TL_PTR", None)
        v = dbg.cpu().numpy().astype("float64")
        tiles, waves = max(v[14], 1.0), max(v[15], 1.0)
        per = v[:14] / waves  # cycles per WAVE over the launch (a team's waves each carry the whole flow)
        import ctypes
        from nessai_amd import lib
        L = lib.load()
        g = [ctypes.c_int(0) for _ in range(3)]
        L.nsa_last_geometry(*[ctypes.addressof(q) for q in g])
        print(json.dumps(dict(library=lib._LIB_PATH, geometry=[q.value for q in g], last_team=int(L.nsa_last_team()),
                              last_ring=int(L.nsa_last_ring()), precision=flow.effective_precision)), flush=True)
        print(json.dumps(dict(team=team, ring=ring, ms_median=round(ms, 4), waves=int(waves), tile_visits=int(tiles),
                              cycles_per_wave=round(float(per.sum()), 0),
                              sections={k: round(float(c), 0) for k, c in zip(SLOTS, per)})), flush=True)


if __name__ == "__main__":
    main()
