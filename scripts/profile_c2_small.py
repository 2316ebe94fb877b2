#!/usr/bin/env python
"""Where a populate() of BASELINE config 2 at the reference's default ``drawsize = nlive = 2000`` spends its
time: wall time per call, GPU time of its kernels (HIP events around the whole call), and a cProfile of the
host side.  Run from the repo root on a GPU box."""
import cProfile
import json
import os
import pstats
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from nessai_amd.model import GaussianModel  # noqa: E402
from nessai_amd.proposal import FlowProposal  # noqa: E402


def main():
    rng = np.random.default_rng(1451)
    torch.manual_seed(1451)
    model = GaussianModel(16)
    with tempfile.TemporaryDirectory() as tmp:
        prop = FlowProposal(model, rng=rng, flow_config=dict(ftype="realnvp", n_neurons=32, n_blocks=4, n_layers=2),
                            output=tmp, poolsize=2000, drawsize=2000, device_seed=1451)
        prop.initialise()
        pts = model.new_point(80_000, rng=rng)
        pts["logL"] = model.log_likelihood(pts)
        prop.train(pts[np.argsort(pts["logL"])[-2000:]])
        for _ in range(3):
            prop.populate(None, n_samples=2000, max_samples=10**12)
        torch.cuda.synchronize()
        wall, gpu, passes = [], [], []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t = time.perf_counter()
            a.record()
            prop.populate(None, n_samples=2000, max_samples=10**12)
            b.record()
            torch.cuda.synchronize()
            wall.append(1e3 * (time.perf_counter() - t))
            gpu.append(a.elapsed_time(b))
            passes.append(prop.n_proposed // 2000)
        print(json.dumps(dict(config="C2 drawsize=nlive=2000", wall_ms=float(np.mean(wall)), first_to_last_event_ms=float(np.mean(gpu)),
                              passes_per_populate=float(np.mean(passes)), precision=prop.flow.model.effective_precision,
                              path=prop.last_populate_path)))
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(200):
            prop.populate(None, n_samples=2000, max_samples=10**12)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
