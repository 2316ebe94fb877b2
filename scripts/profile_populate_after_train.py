#!/usr/bin/env python
"""cProfile of the FIRST populate() after a real train() (config 3): what a nested-sampling run pays once per
training on top of the populate itself - the precision probe, the flow images, the rescaling update."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    prop = bench.build_proposal(0, 1, device)
    rng = np.random.default_rng(3)
    live = prop.model.new_point(bench.NLIVE, rng=rng)
    live["logL"] = prop.model.log_likelihood(live)
    for rep in range(3):
        prop.train(live)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        t = time.perf_counter()
        pr.enable()
        prop.populate(live[0], n_samples=bench.NLIVE, max_samples=10**12)
        torch.cuda.synchronize()
        pr.disable()
        first = 1e3 * (time.perf_counter() - t)
        t = time.perf_counter()
        prop.populate(live[0], n_samples=bench.NLIVE, max_samples=10**12)
        torch.cuda.synchronize()
        second = 1e3 * (time.perf_counter() - t)
        print(f"rep {rep}: first populate after train {first:.2f} ms, next {second:.2f} ms, precision "
              f"{prop.flow.model.effective_precision}")
        if rep < 2:
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12)
            print(s.getvalue()[:3000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(25)
    print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
