#!/usr/bin/env python
"""Randomised check of the 16-bit matrix-core modes (``bf16x3``, ``f16x3``) against the CPU oracle:
rational-quadratic flows with >= 15 features and >= 30 hidden units (the static envelope in which
``precision="auto"`` tries split-bf16 first), random shapes, weights at initialisation scale and blown up
by 2x / 3x (trained flows carry larger weights), inputs at 0.5 / 1 / 3 sigma; with ``--wide`` also
RealNVP / narrow flows outside that envelope.  Prints, per mode, the worst relative ``log_prob`` error
per weight scale and every case beyond the 1e-5 bar, and how often split-f16 left the f16 range.

    python scripts/fuzz_precision.py [--iterations 120] [--seed 0] [--modes bf16x3,f16x3,f32] [--wide]"""
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
from nessai_amd.flows import NeuralSplineFlow, RealNVP  # noqa: E402
from nessai_amd.flows.base import F16RangeError  # noqa: E402
from oracle import flows as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=120)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--modes", default="bf16x3,f16x3")
    ap.add_argument("--wide", action="store_true", help="also RealNVP and flows outside the split-bf16 envelope")
    args = ap.parse_args()
    modes = [m for m in args.modes.split(",") if m]
    g = np.random.default_rng(args.seed)
    worst = {m: {} for m in modes}
    beyond = {m: 0 for m in modes}
    range_left = 0
    for it in range(args.iterations):
        family = "nsf"
        if args.wide and g.random() < 0.5:
            family = str(g.choice(["realnvp", "narrow"]))
        if family == "nsf":
            D = int(g.integers(15, 65))
            H = int(g.choice([30, 32, 48, 64, 96, 128]))
        else:
            D = int(g.integers(2, 33))
            H = int(g.choice([8, 16, 32, 64]))
        blocks, layers = int(g.integers(2, 9)), int(g.integers(1, 4))
        kw = {}
        if family == "realnvp":
            P, Oc = RealNVP, O.RealNVP
            if g.random() < 0.5:
                kw["linear_transform"] = str(g.choice(["lu", "permutation"]))
                kw["batch_norm_between_layers"] = bool(g.random() < 0.5)
        else:
            P, Oc = NeuralSplineFlow, O.NeuralSplineFlow
            kw = dict(num_bins=int(g.choice([4, 8, 10, 16])))
            if g.random() < 0.3:
                kw["linear_transform"] = str(g.choice(["lu", "permutation"]))
        wscale = float(g.choice([1.0, 2.0, 3.0]))
        torch.manual_seed(int(g.integers(0, 2**31)))
        ref = Oc(D, H, blocks, layers, **kw)
        randomise(ref, it)
        with torch.no_grad():
            for name, p in ref.named_parameters():
                if name.endswith("weight") and p.ndim == 2:
                    p.mul_(wscale)
        ref.eval()
        flow = P(D, H, blocks, layers, **kw)
        flow.load_state_dict(ref.state_dict())
        flow.to(DEV)
        flow.eval()
        scale = float(g.choice([0.5, 1.0, 3.0]))
        x = torch.randn(2000, D, generator=torch.Generator().manual_seed(it)) * scale
        with torch.no_grad():
            lp_ref = ref.log_prob(x)
        for mode in modes:
            flow.precision = mode
            try:
                with torch.no_grad():
                    lp = flow.log_prob(x.to(DEV)).cpu()
            except F16RangeError:
                range_left += 1
                continue
            e = rel_err(lp, lp_ref, floor=1e-3)
            worst[mode][wscale] = max(worst[mode].get(wscale, 0.0), e)
            if e > 1e-5:
                beyond[mode] += 1
                print("BEYOND", json.dumps(dict(mode=mode, case=it, family=family, D=D, H=H, blocks=blocks, layers=layers,
                                                weight_scale=wscale, x_scale=scale, rel=e,
                                                min_abs_log_prob=float(lp_ref.abs().min()), **kw)), flush=True)
    print(json.dumps(dict(iterations=args.iterations, wide=args.wide, worst_by_mode_and_weight_scale=worst,
                          beyond_bar=beyond, f16_range_left=range_left)))


if __name__ == "__main__":
    main()
