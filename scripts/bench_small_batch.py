#!/usr/bin/env python
"""Time of ONE small launch of a flow (default: config 4's - 64-D spline flow, 128-wide conditioner - over 5050 rows =
158 tiles) in the geometry the environment selects (NESSAI_AMD_TEAM / NESSAI_AMD_RING / NESSAI_AMD_OCC ..., read at
every launch); with NSA_TL=1 and a -DNSA_TIMELINE build (scripts/timeline_c4.py) also the per-section cycles.
One JSON line.  usage: bench_small_batch.py [rows] [D H layers blocks]"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SLOTS = ["fill+prologue", "layer0 gemm", "layer0 tail", "hidden gemm", "hidden tail", "chunk gemm", "transform",
         "copy wait", "barrier", "epilogue", "other ops", "loop top", "copy issue", "bias read"]


def main():
    from nessai_amd import lib
    from nessai_amd.flows import NeuralSplineFlow

    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5050
    shape = [int(a) for a in sys.argv[2:6]] if len(sys.argv) > 5 else [64, 128, 6, 2]
    torch.manual_seed(3)
    flow = NeuralSplineFlow(*shape)
    flow.to("cuda:0")
    flow.eval()
    flow.precision = os.environ.get("PRECISION", "f16x3")
    torch.set_grad_enabled(False)  # (with gradients on, the forward direction is the TRAINING kernels' business)
    x = torch.randn(rows, shape[0], device="cuda:0")
    inverse = os.environ.get("INVERSE", "0") == "1"
    call = (lambda: flow.inverse(x)) if inverse else (lambda: flow.log_prob(x))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    L = lib.load()
    g = [ctypes.c_int(0) for _ in range(3)]
    L.nsa_last_geometry(*[ctypes.addressof(q) for q in g])
    ev = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        call()
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    out = dict(rows=rows, shape=shape, inverse=inverse, team_env=os.environ.get("NESSAI_AMD_TEAM"), loader_env=os.environ.get("NESSAI_AMD_LOADER"), tiles_env=os.environ.get("NESSAI_AMD_TILES"), rounds_env=os.environ.get("NESSAI_AMD_ROUNDS"), precision=flow.precision, ring_env=os.environ.get("NESSAI_AMD_RING"),
               geometry=[q.value for q in g], team=int(L.nsa_last_team()), ring=int(L.nsa_last_ring()),
               ms_median=round(t[len(t) // 2], 4), ms_min=round(t[0], 4), library=os.path.basename(lib._LIB_PATH))
    if os.environ.get("NSA_TL") == "1":
        dbg = torch.zeros(16, dtype=torch.int64, device="cuda:0")
        os.environ["NSA_TL_PTR"] = str(dbg.data_ptr())
        call()
        torch.cuda.synchronize()
        os.environ.pop("NSA_TL_PTR", None)
        v = dbg.cpu().numpy().astype("float64")
        waves = max(v[15], 1.0)
        out["waves"] = int(v[15])
        out["cycles_per_wave"] = {k: int(c / waves) for k, c in zip(SLOTS, v[:14]) if c > 0}
        out["cycles_total_per_wave"] = int(v[:14].sum() / waves)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
