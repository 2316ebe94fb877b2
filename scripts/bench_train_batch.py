#!/usr/bin/env python
"""Optimiser step time of config 4's flow (64 inputs, 128-wide conditioners) against the minibatch size, for a given cap
on the per-workgroup gradient copies (NESSAI_AMD_TRAIN_SLAB_MB, read once per process: run once per setting):

    NESSAI_AMD_TRAIN_SLAB_MB=256 python scripts/bench_train_batch.py
    NESSAI_AMD_TRAIN_SLAB_MB=4096 python scripts/bench_train_batch.py

Below the cap the backward kernels leave gradient copies (summed in a fixed order); above it they add to the flat
gradient with float atomics."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nessai_amd.flows import NeuralSplineFlow  # noqa: E402
from nessai_amd.train import Trainer  # noqa: E402

for n in (500, 1000, 1024, 1100, 2000, 4000, 8000):
    torch.manual_seed(1)
    flow = NeuralSplineFlow(64, 128, 6, 2)
    flow.to("cuda")
    tr = Trainer(flow)
    x = torch.randn(n, 64, device="cuda")
    for _ in range(5):
        tr.step(x, clip_norm=5.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        tr.step(x, clip_norm=5.0)
    torch.cuda.synchronize()
    print(json.dumps({"rows": n, "slab_cap_mb": os.environ.get("NESSAI_AMD_TRAIN_SLAB_MB", "4096 (default)"),
                      "ms_per_step": (time.perf_counter() - t0) / 50 * 1e3, "loss": float(tr.loss_acc.item())}))
