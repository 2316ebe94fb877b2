#!/usr/bin/env python
"""Drive the MI355X proposal the way the reference's NestedSampler does.

The sequential nested-sampling loop is out of the MI355X scope; this
script is the smallest *caller* of the proposal protocol (SURVEY.md §8b surface B3,
``src/nessai/samplers/nestedsampler.py:643-667, 937-968``): keep ``nlive`` live points, train
the flow on them, and replace the worst point with ``proposal.draw(worst)`` until the pool of
the current flow is exhausted, retraining every ``nlive`` iterations.

    python examples/rosenbrock_proposal.py --dims 8 --nlive 1000 --iterations 3000
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nessai_amd.model import RosenbrockModel
from nessai_amd.proposal import FlowProposal


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", type=int, default=8)
    ap.add_argument("--nlive", type=int, default=1000)
    ap.add_argument("--iterations", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=1451)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    model = RosenbrockModel(args.dims)
    proposal = FlowProposal(
        model, rng=rng, poolsize=args.nlive, drawsize=200_000,
        flow_config=dict(ftype="nsf", n_blocks=4, n_layers=2),
        training_config=dict(max_epochs=200, patience=20),
    )
    proposal.initialise()
    live = model.new_point(args.nlive, rng=rng)
    live["logL"] = model.log_likelihood(live)
    live = np.sort(live, order="logL")
    log_x, log_z = 0.0, -np.inf
    t0 = time.time()
    accepted = 0
    for it in range(args.iterations):
        if it % args.nlive == 0:  # (re)train on the current live points
            proposal.train(live)
        worst = live[0]
        # evidence bookkeeping (trapezoid-free nested sampling estimate)
        log_w = log_x + np.log1p(-np.exp(-1.0 / args.nlive)) + worst["logL"]
        log_z = np.logaddexp(log_z, log_w)
        log_x -= 1.0 / args.nlive
        while True:  # draw until a point above the worst likelihood is found
            new = proposal.draw(worst)
            if new["logL"] > worst["logL"]:
                break
        accepted += 1
        idx = np.searchsorted(live["logL"], new["logL"])
        live[: idx - 1] = live[1:idx]
        live[idx - 1] = new
    dt = time.time() - t0
    print(f"{args.iterations} iterations in {dt:.1f}s, log Z ~ {log_z:.2f}, "
          f"flow trained {proposal.training_count}x, populated {proposal.populated_count}x, "
          f"population time {proposal.population_time.total_seconds():.2f}s, "
          f"last population acceptance {proposal.population_acceptance:.2e}")


if __name__ == "__main__":
    main()
