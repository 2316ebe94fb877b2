#!/usr/bin/env python
"""Randomised parity sweep of the training kernels: loss and every gradient of one minibatch
(-mean log_prob) against torch autograd of the CPU oracle, over random supported flow
configurations and batch sizes (fused per-coupling kernels where they apply, layer-wise otherwise).
A case beyond the bars is re-judged against the oracle in float64 before it counts as a failure:
  * ill-conditioned flow - the float32 oracle is itself further than the bar from its float64 twin:
    the device has to be within 3x the float32 oracle's error of the float64 gradients;
  * relu kink - some pre-activation of some row is within float32 round-off of zero (|pre| < 4e-6 of the
    layer's median: a 100-term float32 sum carries up to ~6e-6; seed 11, case 6 sits at 1.15e-6 and moves one
    gradient by 1.5 % of the largest in the fused kernels of either generation while the layer-wise kernels
    land on the oracle's side): which side of the kink a row falls on, and with it that row's whole contribution
    to the upstream gradients, is decided by the rounding of the last product.  Reported as
    "kink_cases", not as failures (seed 9, case 38: one pre-activation of 3.7e-8 among 0.26-sized ones);
  * spline bin edge - an input within round-off of a knot (see the comment at the re-judgement): "bin_edge_cases".

    python scripts/fuzz_train.py [--iterations 60] [--seed 0]"""
import argparse
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import fuzz_parity as F  # noqa: E402  (random_case)
from helpers import DEV  # noqa: E402
from nessai_amd.flows import NeuralSplineFlow, RealNVP  # noqa: E402
from nessai_amd.train import Trainer  # noqa: E402
from oracle import flows as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, default=None, help="evaluate this case of the sequence only")
    args = ap.parse_args()
    g = np.random.default_rng(args.seed)
    bad, skipped, worst = [], 0, dict(loss=0.0, grad=0.0)
    rejudged = dict(ill_conditioned=0, kink_cases=0, bin_edge_cases=0)
    for it in range(args.iterations):
        kind, shape, kw, act = F.random_case(g)
        if shape[0] > 48:
            shape = (int(g.integers(2, 49)),) + shape[1:]
        P, Oc = (NeuralSplineFlow, O.NeuralSplineFlow) if kind == "nsf" else (RealNVP, O.RealNVP)
        desc = dict(case=it, kind=kind, shape=shape, act=act, **{k: (v if not callable(v) else act) for k, v in kw.items()})
        torch.manual_seed(int(g.integers(0, 2**31)))
        try:
            ref = Oc(*shape, **kw)
            flow = P(*shape, **kw)
        except (ValueError, NotImplementedError, RuntimeError):
            skipped += 1
            continue
        with torch.no_grad():
            gen = torch.Generator().manual_seed(it)
            for n_, p in ref.named_parameters():
                if "linear_layers.1" in n_:
                    p.mul_(200.0)  # residual branches start at ~1e-3: make them matter
                if "entries" in n_ or "unconstrained" in n_:
                    p.add_(0.1 * torch.randn(p.shape, generator=gen))
        flow.load_state_dict(ref.state_dict())
        flow.to(DEV)
        n = int(g.choice([33, 200, 333, 1000]))
        x = torch.randn(n, shape[0], generator=torch.Generator().manual_seed(1000 + it)) * float(g.choice([0.7, 1.5]))
        if args.only is not None and it != args.only:
            continue
        if os.environ.get("FUZZ_NUDGE"):  # the same case with every input moved by a relative FUZZ_NUDGE
            x = x * (1.0 + float(os.environ["FUZZ_NUDGE"]))
        ref.train()
        ref64 = copy.deepcopy(ref).double()
        pre = []  # smallest |pre-activation| of every linear layer, relative to the layer's median
        hooks = [m.register_forward_hook(lambda m_, i_, o_: pre.append(float((o_.detach().abs().min() / o_.detach().abs().median().clamp_min(1e-30)))))
                 for m in ref.modules() if isinstance(m, torch.nn.Linear)]
        loss_ref = -ref.log_prob(x).mean()
        for h in hooks:
            h.remove()
        if not torch.isfinite(loss_ref):
            skipped += 1
            continue
        loss_ref.backward()
        g_ref = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ref.parameters()])
        try:
            tr = Trainer(flow)
            tr.step(x.to(DEV), clip_norm=5.0, lr=0.0)
        except (ValueError, NotImplementedError, RuntimeError) as exc:
            skipped += 1
            if "--verbose" in sys.argv:
                print("skip", json.dumps(desc, default=str), str(exc)[:80])
            continue
        torch.cuda.synchronize()
        e_loss = abs(tr.loss_acc.item() - loss_ref.item()) / max(1.0, abs(loss_ref.item()))
        gd = tr.grads.cpu()
        scale = max(1e-3, float(g_ref.abs().max()))
        e_grad = float((gd - g_ref).abs().max()) / scale
        if e_loss <= 1e-5 and e_grad > 2e-4:
            (-ref64.log_prob(x.double()).mean()).backward()
            g64 = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ref64.parameters()])
            own = float((g_ref.double() - g64).abs().max()) / scale
            if float((gd.double() - g64).abs().max()) / scale <= 3.0 * own:
                rejudged["ill_conditioned"] += 1
                continue
            if act == "relu" and min(pre) < 4e-6:
                rejudged["kink_cases"] += 1
                continue
        if e_loss <= 1e-5 and e_grad > 2e-4:
            # spline bin edge: an input within round-off of a knot is searched into one bin by the device and into its
            # neighbour by torch - same value, same dL/dx (the spline is C1), but the parameter gradients of THAT row
            # go to different bins.  Such a case passes with every input moved by +-1e-5 relative (seed 24, case 34).
            moved = []
            for nudge in (1e-5, -1e-5):
                xm = x * (1.0 + nudge)
                ref.zero_grad()
                (-ref.log_prob(xm).mean()).backward()
                gm = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ref.parameters()])
                flow2 = P(*shape, **kw)
                flow2.load_state_dict(ref.state_dict())
                flow2.to(DEV)
                tr2 = Trainer(flow2)
                tr2.step(xm.to(DEV), clip_norm=5.0, lr=0.0)
                moved.append(float((tr2.grads.cpu() - gm).abs().max()) / max(1e-3, float(gm.abs().max())))
            if max(moved) <= 2e-4:
                rejudged["bin_edge_cases"] += 1
                continue
        worst["loss"] = max(worst["loss"], e_loss)
        worst["grad"] = max(worst["grad"], e_grad)
        if not (e_loss <= 1e-5 and e_grad <= 2e-4):
            at, pos, where = int((gd - g_ref).abs().argmax()), 0, "?"
            for n_, p_ in ref.named_parameters():  # which parameter holds the worst element
                if pos <= at < pos + p_.numel():
                    where = f"{n_}[{at - pos}] device {float(gd[at]):.6g} oracle {float(g_ref[at]):.6g}"
                pos += p_.numel()
            if args.only is not None:  # per-parameter errors of the case under the lens
                pos = 0
                for n_, p_ in ref.named_parameters():
                    e_ = float((gd[pos:pos + p_.numel()] - g_ref[pos:pos + p_.numel()]).abs().max()) / scale
                    if e_ > 1e-5:
                        print(f"  {n_:60s} {tuple(p_.shape)} {e_:.3e}")
                    pos += p_.numel()
            bad.append(dict(n=n, loss_rel=e_loss, grad_rel_to_max=e_grad, smallest_pre_activation=min(pre), worst_at=where,
                            **desc))
            print("FAIL", json.dumps(bad[-1], default=str), flush=True)
    print(json.dumps(dict(iterations=args.iterations, skipped=skipped, failures=len(bad), worst=worst, **rejudged)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
