#!/usr/bin/env python
"""Latency of the fused inference kernel on training-sized batches of config 3 (forward + log_prob on 200 / 800 /
1000 rows, per precision), beside the 6 x trf_forward launches the training step uses: what a training forward
on the inference kernel could cost.  HIP events, device tensors."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_c3_fixture  # noqa: E402
from nessai_amd.flows import NeuralSplineFlow  # noqa: E402

state, shift, scale, live = load_c3_fixture()
flow = NeuralSplineFlow(32, 64, 6, 2)
flow.load_state_dict(state)
flow.to("cuda:0")
flow.eval()
x_all = torch.from_numpy(((live - shift) / scale).astype(np.float32)).to("cuda:0")
out = {}
with torch.no_grad():
    for prec in ("f16x3", "f32"):
        flow.precision = prec
        for n in (200, 800, 1000, 2000):
            x = x_all[:n].contiguous()
            for _ in range(3):
                flow.forward_and_log_prob(x)
            torch.cuda.synchronize()
            ev = []
            for _ in range(20):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                flow.forward_and_log_prob(x)
                b.record()
                ev.append((a, b))
            torch.cuda.synchronize()
            out[f"{prec}_{n}_rows_us"] = round(1e3 * float(np.median([a.elapsed_time(b) for a, b in ev])), 1)
print(json.dumps(out))
