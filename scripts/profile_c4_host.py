#!/usr/bin/env python
"""Host-side profile of ImportanceFlowProposal.draw / update_log_q (config 4): cProfile over 20 calls."""
import cProfile
import os
import pstats
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(n_flows=8, nlive=5000, dims=64):
    from nessai_amd.model import CorrelatedGaussianModel
    from nessai_amd.proposal_importance import ImportanceFlowProposal

    torch.manual_seed(1451)
    model = CorrelatedGaussianModel(dims, rho=0.5)
    with tempfile.TemporaryDirectory() as tmp:
        prop = ImportanceFlowProposal(model, output=tmp, rng=np.random.default_rng(1451), device_seed=1451,
                                      flow_config=dict(ftype="nsf", n_neurons=128, n_blocks=6, n_layers=2, num_bins=8))
        prop.initialise()
        while prop.flow.n_models < n_flows:
            prop.flow.add_new_flow(reset=True)
            prop._weights[prop.flow.n_models - 1] = np.nan
        prop.level_count = n_flows - 1
        prop._weights = {k - 1: 1.0 / (n_flows + 1) for k in range(n_flows + 1)}
        prop.flow.models.eval()
        for _ in range(3):
            pts, _ = prop.draw(nlive)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            prop.draw(nlive)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
        stored = np.concatenate([pts] * 4)
        prop.update_log_q(stored, np.zeros((stored.size, n_flows)))
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            prop.update_log_q(stored, np.zeros((stored.size, n_flows)))
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
