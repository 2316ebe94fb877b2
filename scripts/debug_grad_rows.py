#!/usr/bin/env python
"""Which single rows give the fused training kernels a gradient that differs from a float64 twin (diagnostic)."""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nessai_amd.flows import NeuralSplineFlow  # noqa: E402
from nessai_amd.train import Trainer  # noqa: E402
from oracle import flows as O  # noqa: E402

D, H, NL, NB = 64, 128, 6, 2
torch.manual_seed(3)
ref = O.NeuralSplineFlow(D, H, NL, NB)
with torch.no_grad():
    for n_, p in ref.named_parameters():
        if "linear_layers.1" in n_:
            p.mul_(200.0)
flow = NeuralSplineFlow(D, H, NL, NB)
flow.load_state_dict(ref.state_dict())
flow.to("cuda")
x = torch.randn(64, D) * 1.5
ref64 = copy.deepcopy(ref).double()
ref64.train()
tr = Trainer(flow)
names = [n for n, _ in ref64.named_parameters()]
for i in range(64):
    ref64.zero_grad()
    (-ref64.log_prob(x[[i, i]].double()).mean()).backward()
    g64 = torch.cat([p.grad.reshape(-1) for p in ref64.parameters()])
    tr.step(x[[i, i]].to("cuda"), clip_norm=5.0, lr=0.0)
    torch.cuda.synchronize()
    err = (tr.grads.cpu().double() - g64).abs()
    if err.max() > 1e-4:
        j = int(err.argmax())
        off = 0
        for n_, p in ref64.named_parameters():
            if j < off + p.numel():
                print("row", i, "max err", float(err.max()), "at", n_, "index", j - off, "shape", tuple(p.shape),
                      "ref", float(g64[j]), "dev", float(tr.grads[j]))
                break
            off += p.numel()
        # the coupling's input/outputs for that row in the oracle: which transformed feature is it?
        if "final_layer.weight" in n_:
            o = (j - off) // p.shape[1]
            print("   final-layer output", o, "= feature", o // 23, "parameter", o % 23)
