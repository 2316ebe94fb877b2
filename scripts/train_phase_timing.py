#!/usr/bin/env python
"""Phase timestamps of the fused training kernels (workgroup 0, last coupling layer).

    NESSAI_AMD_TRAIN_TIMING=1 python scripts/train_phase_timing.py

Prints the cycle counter (s_memtime, 100 MHz constant clock on gfx950 -> 10 ns units) at the
marks of nsa_train_fused.h: forward 0..22, backward 60 (end)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NESSAI_AMD_TRAIN_TIMING"] = "1"

from nessai_amd.flows import NeuralSplineFlow  # noqa: E402
from nessai_amd.train import Trainer  # noqa: E402

torch.manual_seed(1)
SHAPE = (64, 128, 6, 2) if "--c4" in sys.argv else (32, 64, 6, 2)  # --c4: config 4's flow (third-generation kernels; NESSAI_AMD_TRAIN_V3=0: the first)
flow = NeuralSplineFlow(*SHAPE)
flow.to("cuda")
tr = Trainer(flow)
x = torch.randn(1000, SHAPE[0], device="cuda")
for _ in range(3):
    tr.step(x, clip_norm=5.0)
torch.cuda.synchronize()
marks = tr._ws[64:128].cpu().numpy()
if "--fine" in sys.argv:  # stamps inside the layer h_1 of the first-generation forward kernel (trf_linear)
    fine = tr._ws[108:108 + 12].cpu().numpy()
    print("fine stamps of layer h_1 (entry, operands at hand, then per step: MFMAs issued / next operands landed, ..., end):")
    print("  ", [int(v) for v in fine], "\n   deltas", [int(b - a) for a, b in zip(fine[:-1], fine[1:])])
names = {0: "x staged", 1: "layer 0", 2: "u_0", 3: "h_1", 4: "u_1", 5: "h_2", 20: "params", 21: "barrier",
         22: "transform fwd", 30: "bwd: tiles in LDS", 31: "bwd: transform (gen 1: items done)", 32: "bwd: final layer",
         33: "bwd: last block", 34: "bwd: block before", 35: "bwd: tile loads issued (gen 3)", 36: "bwd: tiles stored (gen 3)",
         37: "bwd: operand streams requested (gen 3)", 40: "bwd: later loads issued", 41: "bwd: items done",
         42: "bwd: final layer dX (gen 3)", 43: "bwd: final layer dW (gen 3)", 50: "bwd: block tiles requested (gen 3)",
         51: "bwd: barrier (gen 3)", 60: "bwd: initial layer + end"}
prev = 0.0
for group in (range(0, 23), range(30, 61)):
    prev = 0.0
    for i in sorted(group, key=lambda k: marks[k]):
        if marks[i] > 0:
            print(f"mark {i:2d} {names.get(i, ''):22s} t = {marks[i]:9.0f}  (+{marks[i] - prev:7.0f})")
            prev = marks[i]
    print()
