#!/usr/bin/env python
"""A nested-sampling run on BASELINE config 3 driven through the proposal protocol (the sampler loop itself is
out of scope: this is the smallest caller, as examples/rosenbrock_proposal.py), to see what a REAL run asks of
the path: the flow is retrained every ``nlive`` iterations, so every population is drawn from freshly trained
weights.  Reports, per training, what ``precision="auto"`` measured and chose, and where the wall time went
(training / populate incl. the image rebuild / the Python loop).  One JSON line at the end.

    python scripts/run_ns_c3.py [--iterations 12000] [--dims 32] [--nlive 2000] [--drawsize 1000000]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nessai_amd.model import RosenbrockModel  # noqa: E402
from nessai_amd.proposal import FlowProposal  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=12000)
    ap.add_argument("--dims", type=int, default=32)
    ap.add_argument("--nlive", type=int, default=2000)
    ap.add_argument("--drawsize", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=1451)
    ap.add_argument("--precision", default="auto")
    ap.add_argument("--save-branch", default=None,
                    help="file (.npz): the weights of the first retrained flow that 'auto' kept in split-f16 through its "
                         "noise-floor branch (probe above the 5e-6 bar), or failing that of the retrain with the largest probe")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    torch.manual_seed(args.seed)
    model = RosenbrockModel(args.dims)
    with tempfile.TemporaryDirectory() as tmp:
        prop = FlowProposal(model, rng=rng, poolsize=args.nlive, drawsize=args.drawsize, output=tmp,
                            flow_config=dict(ftype="nsf", n_neurons=2 * args.dims, n_blocks=6, n_layers=2, num_bins=8),
                            device_seed=args.seed, precision=args.precision)
        prop.initialise()
        live = model.new_point(args.nlive, rng=rng)
        live["logL"] = model.log_likelihood(live)
        live = np.sort(live, order="logL")
        trainings, populations = [], []
        t_train = t_pop = 0.0
        seen_populated = prop.populated_count
        seen_stats = {}
        best_probe = [0.0]
        n_draws = 0
        t0 = time.perf_counter()
        for it in range(args.iterations):
            if it % args.nlive == 0:
                torch.cuda.synchronize()
                t = time.perf_counter()
                prop.train(live)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t
                t_train += dt
                trainings.append(dict(iteration=it, seconds=dt, worst_logL=float(live["logL"][0])))
            worst = live[0]
            for _ in range(100_000):
                t = time.perf_counter()
                new = prop.draw(worst)
                n_draws += 1
                if prop.populated_count != seen_populated:  # this draw repopulated the pool
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t
                    t_pop += dt
                    seen_populated = prop.populated_count
                    flow = prop.flow.model
                    stats = dict(flow.__dict__.get("image_stats", {}))
                    populations.append(dict(iteration=it, ms=1e3 * dt, passes=prop.n_proposed // args.drawsize,
                                            acceptance=prop.population_acceptance, precision=flow.effective_precision,
                                            probe=dict(flow.probe_errors), loop=prop.last_populate_path,
                                            phases_ms={k: 1e3 * v for k, v in (prop.timing or {}).items()},
                                            images={k: stats.get(k, 0) - seen_stats.get(k, 0) for k in stats}))
                    seen_stats = stats
                    err = flow.probe_errors.get("f16x3", 0.0)
                    if args.save_branch and err > best_probe[0] and populations[-1]["images"].get("probe"):
                        best_probe[0] = err
                        state = {k: v.detach().cpu().numpy() for k, v in flow.state_dict().items()}
                        np.savez_compressed(args.save_branch, **state, __probe_f16x3=err,
                                            __probe_f32_one_ulp=flow.probe_errors.get("f32_one_ulp", float("nan")),
                                            __iteration=it, __seed=args.seed)
                if new["logL"] > worst["logL"]:
                    break
            else:
                raise RuntimeError("no point above the worst likelihood in 100000 draws")
            idx = np.searchsorted(live["logL"], new["logL"])
            live[: idx - 1] = live[1:idx]
            live[idx - 1] = new
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    chosen = {}
    for p in populations:
        chosen[p["precision"]] = chosen.get(p["precision"], 0) + 1
    first_after_train = [p for k, p in enumerate(populations) if k == 0 or any(
        t_["iteration"] <= p["iteration"] and t_["iteration"] > populations[k - 1]["iteration"] for t_ in trainings)]
    def phase_means(pops):
        keys = sorted({k for p in pops for k in p["phases_ms"]})
        return {k: float(np.mean([p["phases_ms"].get(k, 0.0) for p in pops])) for k in keys} if pops else {}

    steady = [p for p in populations if p not in first_after_train]
    # how often "auto" kept split-f16 through its noise-floor branch (probe above the 5e-6 bar but within 4 x the
    # float32 kernel's own response to a one-ulp input change: HipFlow._pack_auto)
    noise_branch = [p for p in populations if p["precision"] == "f16x3" and p["probe"].get("f16x3", 0.0) > 5e-6]
    out = dict(
        config=f"{args.dims}-D Rosenbrock, NSF 8 bins x 6 layers, nlive {args.nlive}, drawsize {args.drawsize}",
        phases_ms_first_after_train=phase_means(first_after_train[1:]),  # (the very first call also compiles / allocates)
        phases_ms_steady=phase_means(steady),
        images_built_first_after_train=[p["images"] for p in first_after_train],
        populate_ms_first_after_train_excluding_first_call=(float(np.mean([p["ms"] for p in first_after_train[1:]]))
                                                            if len(first_after_train) > 1 else None),
        populations_through_noise_floor_branch=len(noise_branch),
        trainings_through_noise_floor_branch=len({p["iteration"] // args.nlive for p in noise_branch}),
        timing_note="phase marks synchronise the device (NESSAI_AMD_TIMING=1): the per-call times include that",
        iterations=args.iterations, draws=n_draws, insertion_acceptance=args.iterations / n_draws,
        trainings=len(trainings), populations=len(populations), wall_s=wall, train_s=t_train, populate_s=t_pop,
        loop_s=wall - t_train - t_pop,
        train_ms_mean=1e3 * t_train / max(1, len(trainings)),
        populate_ms_mean=float(np.mean([p["ms"] for p in populations])) if populations else None,
        populate_ms_first_after_train_mean=float(np.mean([p["ms"] for p in first_after_train])) if first_after_train else None,
        passes_per_population_mean=float(np.mean([p["passes"] for p in populations])) if populations else None,
        precision_chosen=chosen,
        worst_probe_error_f16x3=max((p["probe"].get("f16x3", 0.0) for p in populations), default=None),
        final_worst_logL=float(live["logL"][0]),
        populations_detail=[dict(it=p["iteration"], ms=round(p["ms"], 2), passes=p["passes"],
                                 acc=float(f"{p['acceptance']:.3g}"), prec=p["precision"],
                                 probe={k: float(f"{v:.2g}") for k, v in p["probe"].items()},
                                 phases={k: round(v, 2) for k, v in p["phases_ms"].items()}) for p in populations],
    )
    print(json.dumps(out))


if __name__ == "__main__":
    main()
