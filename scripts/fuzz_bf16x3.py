#!/usr/bin/env python
"""Randomised check of the split-bf16 envelope of the populate precision policy
(nessai_amd.flows.base.populate_precision): rational-quadratic flows with >= 15 features and >= 30
hidden units, random shapes, weights at initialisation scale and blown up by 2x / 3x (trained flows
carry larger weights), inputs at 0.5 / 1 / 3 sigma.  log_prob of the bf16x3 kernel against the CPU
oracle; prints the worst relative error per weight scale and every case beyond the 1e-5 bar.

    python scripts/fuzz_bf16x3.py [--iterations 120] [--seed 0]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import DEV, randomise, rel_err  # noqa: E402
from nessai_amd.flows import NeuralSplineFlow  # noqa: E402
from nessai_amd.flows.base import populate_precision  # noqa: E402
from oracle import flows as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=120)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    g = np.random.default_rng(args.seed)
    worst, beyond = {}, []
    for it in range(args.iterations):
        D = int(g.integers(15, 65))
        H = int(g.choice([30, 32, 48, 64, 96, 128]))
        blocks, layers = int(g.integers(2, 9)), int(g.integers(1, 4))
        kw = dict(num_bins=int(g.choice([4, 8, 10, 16])))
        if g.random() < 0.3:
            kw["linear_transform"] = str(g.choice(["lu", "permutation"]))
        wscale = float(g.choice([1.0, 2.0, 3.0]))
        torch.manual_seed(int(g.integers(0, 2**31)))
        ref = O.NeuralSplineFlow(D, H, blocks, layers, **kw)
        randomise(ref, it)
        with torch.no_grad():
            for name, p in ref.named_parameters():
                if name.endswith("weight") and p.ndim == 2:
                    p.mul_(wscale)
        ref.eval()
        flow = NeuralSplineFlow(D, H, blocks, layers, **kw)
        flow.load_state_dict(ref.state_dict())
        flow.to(DEV)
        flow.eval()
        assert populate_precision(flow) == "bf16x3"
        flow.precision = "bf16x3"
        scale = float(g.choice([0.5, 1.0, 3.0]))
        x = torch.randn(2000, D, generator=torch.Generator().manual_seed(it)) * scale
        with torch.no_grad():
            lp_ref = ref.log_prob(x)
            lp = flow.log_prob(x.to(DEV)).cpu()
        e = rel_err(lp, lp_ref, floor=1e-3)
        worst[wscale] = max(worst.get(wscale, 0.0), e)
        if e > 1e-5:
            beyond.append(dict(case=it, D=D, H=H, blocks=blocks, layers=layers, weight_scale=wscale, x_scale=scale,
                               rel=e, min_abs_log_prob=float(lp_ref.abs().min()), **kw))
            print("BEYOND", json.dumps(beyond[-1]), flush=True)
    print(json.dumps(dict(iterations=args.iterations, worst_by_weight_scale=worst, beyond_bar=len(beyond))))


if __name__ == "__main__":
    main()
