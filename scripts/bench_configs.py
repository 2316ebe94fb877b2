#!/usr/bin/env python
"""Secondary measurements on the other BASELINE.json configs (one JSON line each).

  C2  16-D unit Gaussian, RealNVP (4 coupling layers, 2x32 residual net, LU + batch norm), nlive 2000
  C5  15-D GW analytic surrogate (bilby-style priors), NSF (8 bins, 4 layers), nlive 4000
  C4  64-D NSF flows of the importance sampler: log_prob_all over n_flows for nlive 5000

Each config: live points = best 2000/4000 of prior draws; the flow is TRAINED on the GPU
(FlowProposal.train), then populate() is timed; the CPU oracle runs one bounded pass with the
same weights on one host thread.
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nessai_amd.flowmodel_importance import ImportanceFlowModel  # noqa: E402
from nessai_amd.model import GaussianModel, GWSurrogateModel  # noqa: E402
from nessai_amd.proposal import FlowProposal  # noqa: E402
from oracle import proposal as OP  # noqa: E402


def live_points(model, nlive, n_prior, rng):
    pts = model.new_point(n_prior, rng=rng)
    logl = model.log_likelihood(pts)
    pts["logL"] = logl
    return pts[np.argsort(logl)[-nlive:]]


def run(tag, model, flow_config, nlive, drawsize, steps=5, cpu_draw=50_000):
    rng = np.random.default_rng(1451)
    torch.manual_seed(1451)
    live = live_points(model, nlive, 40 * nlive, rng)
    with tempfile.TemporaryDirectory() as tmp:
        prop = FlowProposal(model, rng=rng, flow_config=flow_config, output=tmp, poolsize=nlive,
                            drawsize=drawsize, device_seed=1451)
        prop.initialise()
        torch.cuda.synchronize()
        t = time.perf_counter()
        prop.train(live)
        torch.cuda.synchronize()
        train_s = time.perf_counter() - t
        prop.populate(None, n_samples=nlive, max_samples=10**12)
        torch.cuda.synchronize()
        t = time.perf_counter()
        acc = prop_n = 0
        for _ in range(steps):
            prop.populate(None, n_samples=nlive, max_samples=10**12)
            acc += prop.n_accepted
            prop_n += prop.n_proposed
        torch.cuda.synchronize()
        gpu_s = time.perf_counter() - t
        state = {k: v.detach().cpu() for k, v in prop.flow.model.state_dict().items()}
        # CPU oracle, one thread, one bounded pass
        torch.set_num_threads(1)
        cfg = dict(flow_config, n_inputs=model.dims)
        cfg.setdefault("n_neurons", 2 * model.dims)
        ofm = OP.OracleFlowModel(cfg, rng=np.random.default_rng(1))
        ofm.model.load_state_dict(state)
        ofm.model.eval()
        oprop = OP.OracleFlowProposal(model, ofm, rng=np.random.default_rng(1), poolsize=nlive, drawsize=cpu_draw)
        for n in model.names:
            oprop.shift[n], oprop.scale[n] = prop.shift[n], prop.scale[n]
        t = time.perf_counter()
        oprop.populate(n_samples=nlive, max_samples=cpu_draw - 1)
        cpu_s = time.perf_counter() - t
    out = {
        "config": tag, "nlive": nlive, "drawsize": drawsize,
        "precision": prop.flow.model.effective_precision,
        "gpu_train_seconds": train_s,
        "gpu_accepted_per_s": acc / gpu_s, "gpu_rows_per_s": prop_n / gpu_s,
        "gpu_acceptance": acc / prop_n, "gpu_ms_per_populate": 1e3 * gpu_s / steps,
        "cpu_accepted_per_s": oprop.n_accepted / cpu_s, "cpu_rows_per_s": oprop.n_proposed / cpu_s,
        "cpu_acceptance": oprop.population_acceptance, "cpu_sample_draws": cpu_draw,
    }
    out["speedup_rows"] = out["gpu_rows_per_s"] / out["cpu_rows_per_s"]
    print(json.dumps(out), flush=True)


def run_c4(n_flows_list=(1, 8), n=5000, dims=64):
    rng = np.random.default_rng(1451)
    torch.manual_seed(1451)
    with tempfile.TemporaryDirectory() as tmp:
        ifm = ImportanceFlowModel(flow_config=dict(n_inputs=dims, n_neurons=128, n_blocks=6, n_layers=2, ftype="nsf"),
                                  output=tmp, rng=rng)
        x = rng.normal(size=(n, dims))
        for target in n_flows_list:
            while ifm.n_models < target:
                ifm.add_new_flow(reset=True)
            ifm.log_prob_all(x)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                lp = ifm.log_prob_all(x)
            torch.cuda.synchronize()
            gpu_s = (time.perf_counter() - t) / 5
            torch.set_num_threads(1)
            ofm = OP.OracleFlowModel(dict(n_inputs=dims, n_neurons=128, n_blocks=6, n_layers=2, ftype="nsf"))
            ofm.model.load_state_dict({k: v.cpu() for k, v in ifm.models[0].state_dict().items()})
            t = time.perf_counter()
            ref = ofm.log_prob(x)
            cpu_one = time.perf_counter() - t
            err = float(np.max(np.abs(lp[:, 0] - ref) / np.maximum(np.abs(ref), 1.0)))
            print(json.dumps({"config": "C4 log_prob_all 64-D NSF", "n_flows": target, "rows": n,
                              "gpu_ms": 1e3 * gpu_s, "cpu_ms_estimate": 1e3 * cpu_one * target,
                              "speedup": cpu_one * target / gpu_s, "log_prob_rel_err_flow0": err}), flush=True)


def run_c4_draw(n_flows_list=(1, 8, 32), nlive=5000, dims=64):
    """Config 4 as a workload: ImportanceFlowProposal.draw(nlive) on the 64-D correlated Gaussian
    (draw -> sigmoid -> log_prob under every stored flow -> weighted logsumexp), points per second."""
    from nessai_amd.model import CorrelatedGaussianModel
    from nessai_amd.proposal_importance import ImportanceFlowProposal

    torch.manual_seed(1451)
    model = CorrelatedGaussianModel(dims, rho=0.5)
    with tempfile.TemporaryDirectory() as tmp:
        prop = ImportanceFlowProposal(model, output=tmp, rng=np.random.default_rng(1451), device_seed=1451,
                                      flow_config=dict(ftype="nsf", n_neurons=128, n_blocks=6, n_layers=2, num_bins=8))
        prop.initialise()
        for target in n_flows_list:
            while prop.flow.n_models < target:
                prop.flow.add_new_flow(reset=True)
                prop._weights[prop.flow.n_models - 1] = np.nan
            prop.level_count = target - 1
            prop._weights = {k - 1: 1.0 / (target + 1) for k in range(target + 1)}
            prop.flow.models.eval()
            prop.draw(nlive)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                pts, log_q = prop.draw(nlive)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 5
            stored = np.concatenate([pts] * 4)
            t = time.perf_counter()
            for _ in range(3):
                prop.update_log_q(stored, np.zeros((stored.size, target)))
            torch.cuda.synchronize()
            du = (time.perf_counter() - t) / 3
            print(json.dumps({"config": "C4 ImportanceFlowProposal.draw, 64-D correlated Gaussian", "n_flows": target,
                              "nlive": nlive, "draw_ms": 1e3 * dt, "points_per_s": nlive / dt,
                              "update_log_q_ms_20000_rows": 1e3 * du}), flush=True)


if __name__ == "__main__":
    run("C2 16-D Gaussian RealNVP (LU+BN), drawsize=nlive", GaussianModel(16),
        dict(ftype="realnvp", n_neurons=32, n_blocks=4, n_layers=2), 2000, 2000, cpu_draw=2000)
    run("C2 16-D Gaussian RealNVP (LU+BN), drawsize=1e6", GaussianModel(16),
        dict(ftype="realnvp", n_neurons=32, n_blocks=4, n_layers=2), 2000, 1_000_000, cpu_draw=100_000)
    run("C5 15-D GW surrogate NSF", GWSurrogateModel(),
        dict(ftype="nsf", n_neurons=30, n_blocks=4, n_layers=2), 4000, 1_000_000, cpu_draw=100_000)
    run_c4()
    run_c4_draw()
