#!/usr/bin/env python
"""What keeping a config-3-shaped coupling's weights in LDS would buy, measured on flows small enough to try:
spline flows with config 3's conditioner width and feature count but ONE coupling and one residual block
(135 KB of split-f16 stages: it fits beside four x tiles), and narrower / shallower relatives, each timed per
1e6 rows in the staged geometries and in the weights-resident one the library would pick.  One JSON line per
flow.  A whole config-3 coupling (168 KB of stages) does not fit in the 160 KB of a CU at all; see DESIGN.md
section 4.1 for what follows from these numbers.  Run from the repo root on a GPU box."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nessai_amd import lib  # noqa: E402
from nessai_amd.flows import NeuralSplineFlow  # noqa: E402


def geometry():
    v = [ctypes.c_int(0) for _ in range(3)]
    L = lib.load()
    L.nsa_last_geometry(*[ctypes.addressof(x) for x in v])
    return dict(waves_per_simd=v[0].value, waves_per_group=v[1].value, groups=v[2].value,
                row_tiles=int(L.nsa_last_row_tiles()), resident=int(L.nsa_last_resident()))


def time_inverse(flow, z, launches=8):
    with torch.no_grad():
        for _ in range(2):
            flow.inverse(z)
        torch.cuda.synchronize()
        ev = []
        for _ in range(launches):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            flow.inverse(z)
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / launches, geometry()


def main():
    n = 1_000_000
    for name, args in (("C3 conditioner, 1 coupling, 1 block (135 KB of stages)", (32, 64, 1, 1)),
                       ("C3 conditioner, 1 coupling, 2 blocks (168 KB: does not fit)", (32, 64, 1, 2)),
                       ("32-wide conditioner, 2 couplings, 2 blocks", (32, 32, 2, 2)),
                       ("16 features, 64-wide, 1 coupling, 2 blocks", (16, 64, 1, 2))):
        torch.manual_seed(3)
        flow = NeuralSplineFlow(*args)
        flow.to("cuda:0")
        flow.eval()
        flow.precision = "f16x3"
        z = torch.randn(n, args[0], device="cuda:0")
        out = dict(flow=name, args=args, rows=n)
        for label, env in (("staged 2x2", {"NESSAI_AMD_RESIDENT": "0", "NESSAI_AMD_OCC": "2x2"}),
                           ("staged 2 x 4 waves", {"NESSAI_AMD_RESIDENT": "0", "NESSAI_AMD_OCC": "2"}),
                           ("resident (library choice)", {})):
            for k in ("NESSAI_AMD_RESIDENT", "NESSAI_AMD_OCC"):
                os.environ.pop(k, None)
            os.environ.update(env)
            ms, geo = time_inverse(flow, z)
            out[label] = dict(ms=round(ms, 4), geometry=geo)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
