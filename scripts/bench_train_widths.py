#!/usr/bin/env python
"""Optimiser step time of neural spline flows with the reference's default conditioner width (n_neurons = 2 n_inputs)
at 36 ... 64 inputs - 72 ... 128 wide: the 16-row kernels of csrc/nsa_train_fused3.h against the first
generation (NESSAI_AMD_TRAIN_V3=0, read per call), 1000-row minibatches, 6 coupling layers, 2 blocks:

    python scripts/bench_train_widths.py
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nessai_amd.flows import NeuralSplineFlow  # noqa: E402
from nessai_amd.train import Trainer  # noqa: E402

for dims in (36, 40, 48, 50, 56, 64):
    for mode in ("0", "1"):
        os.environ["NESSAI_AMD_TRAIN_V3"] = mode
        torch.manual_seed(1)
        flow = NeuralSplineFlow(dims, 2 * dims, 6, 2)
        flow.to("cuda")
        tr = Trainer(flow)
        x = torch.randn(1000, dims, device="cuda")
        for _ in range(10):
            tr.step(x, clip_norm=5.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            tr.step(x, clip_norm=5.0)
        torch.cuda.synchronize()
        print(json.dumps({"metric": "optimiser step, ms", "dims": dims, "neurons": 2 * dims, "NESSAI_AMD_TRAIN_V3": mode,
                          "ms_per_step": (time.perf_counter() - t0) / 200 * 1e3, "loss": float(tr.loss_acc.item())}))
