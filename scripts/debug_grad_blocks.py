#!/usr/bin/env python
"""Per-parameter-tensor gradient error of the device trainer against torch autograd of a float64 twin of the oracle
(diagnostic): python scripts/debug_grad_blocks.py [dims neurons couplings blocks rows]"""
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

args = [int(v) for v in sys.argv[1:6]] + [64, 128, 6, 2, 333][len(sys.argv) - 1:]
D, H, NL, NB, N = args
scale_up = float(os.environ.get("SCALE_UP", "200"))
torch.manual_seed(3)
kw = dict(activation=torch.tanh) if os.environ.get("ACT") == "tanh" else {}
ref = O.NeuralSplineFlow(D, H, NL, NB, **kw)
with torch.no_grad():
    for n_, p in ref.named_parameters():
        if "linear_layers.1" in n_:
            p.mul_(scale_up)
flow = NeuralSplineFlow(D, H, NL, NB, **kw)
flow.load_state_dict(ref.state_dict())
flow.to("cuda")
x = torch.randn(N, D) * 1.5
ref64 = copy.deepcopy(ref).double()
ref64.train()
loss = -ref64.log_prob(x.double()).mean()
loss.backward()
tr = Trainer(flow)
tr.step(x.to("cuda"), clip_norm=5.0, lr=0.0)
torch.cuda.synchronize()
g = tr.grads.cpu().double()
print("loss", float(loss), float(tr.loss_acc.item()))
off = 0
worst = []
for name, p in ref64.named_parameters():
    k = p.numel()
    err = (g[off:off + k] - p.grad.reshape(-1)).abs().max().item()
    worst.append((err, name, p.grad.abs().max().item()))
    off += k
for err, name, mx in sorted(worst, reverse=True)[:12]:
    print(f"{err:.3e}  (max |g| {mx:.3e})  {name}")
