#!/usr/bin/env python
"""Flow-kernel time of config 3's flow (32-D NSF, 6 couplings, 2 x 64 residual conditioner) by conditioner activation:
inverse pass of 1e6 rows, precision from the command line (default f16x3).  One JSON line per activation."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nessai_amd.flows import NeuralSplineFlow  # noqa: E402


PRECISION = sys.argv[1] if len(sys.argv) > 1 else "f16x3"


def main():
    torch.manual_seed(3)
    z = torch.randn(1_000_000, 32, device="cuda")
    for name, act in (("relu", torch.relu), ("tanh", torch.tanh), ("silu", torch.nn.functional.silu)):
        flow = NeuralSplineFlow(32, 64, 6, 2, activation=act)
        flow.to("cuda")
        flow.eval()
        flow.precision = PRECISION
        with torch.inference_mode():
            for _ in range(20):
                flow.inverse(z)
            torch.cuda.synchronize()
            ev = []
            for _ in range(20):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                x, ld = flow.inverse(z)
                b.record()
                ev.append((a, b))
            torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        print(json.dumps(dict(activation=name, precision=flow.effective_precision, ms_median=ms[len(ms) // 2], ms_min=ms[0],
                              checksum=float(ld.double().sum()))), flush=True)


if __name__ == "__main__":
    main()
