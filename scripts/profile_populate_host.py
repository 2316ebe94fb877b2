#!/usr/bin/env python
"""cProfile of the host side of ``FlowProposal.populate`` on the benchmark configuration
(run from the repo root on a GPU box): shows what the CPU does between kernel launches."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import bench
prop = bench.build_proposal(0, 1, torch.device("cuda", 0))
for _ in range(2):
    prop.populate(None, n_samples=2000, max_samples=10**12)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    prop.populate(None, n_samples=2000, max_samples=10**12)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
