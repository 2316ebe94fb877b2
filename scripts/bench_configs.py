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


PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "f16x3": 2500.0}  # dense MFMA peaks (MI355X_MICROARCH.md)
PEAK_HBM_TBS = 8.0


def flops_per_row(flow):
    """Algorithmic conditioner FLOPs per row of one pass through ``flow`` (GEMMs only,
    one product per weight): every Linear layer of every conditioner plus the LU-linear products."""
    from nessai_amd.flows import modules as M

    total = 0
    for m in flow.modules():
        if isinstance(m, torch.nn.Linear):
            total += 2 * m.in_features * m.out_features
        elif isinstance(m, M.LULinear):
            total += 2 * flow.features * flow.features
    return total


def kernel_roofline(flow, precision, rows, kernel_ms, bytes_per_row, bound):
    """SURVEY.md section 8d for one launch: achieved TFLOP/s against the mode's dense MFMA peak AND achieved
    algorithmic GB/s against HBM; ``bound`` names the one the contract assigns to this config."""
    flops = flops_per_row(flow)
    tf = flops * rows / (kernel_ms * 1e-3) / 1e12
    tbs = bytes_per_row * rows / (kernel_ms * 1e-3) / 1e12
    frac_mfma, frac_hbm = tf / PEAK_TFLOPS[precision], tbs / PEAK_HBM_TBS
    return {
        "bound": bound, "kernel_ms_avg": kernel_ms, "rows_per_launch": rows, "precision": precision,
        "flops_per_row": flops, "bytes_per_row": bytes_per_row,
        "achieved_tflops": tf, "peak_tflops": PEAK_TFLOPS[precision], "frac_mfma": frac_mfma,
        "achieved_tb_per_s": tbs, "peak_tb_per_s": PEAK_HBM_TBS, "frac_hbm": frac_hbm,
        "achieved": tbs * 1e3 if bound == "hbm" else tf, "peak": PEAK_HBM_TBS * 1e3 if bound == "hbm" else PEAK_TFLOPS[precision],
        "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": frac_hbm if bound == "hbm" else frac_mfma,
        "traffic": None,
    }


def time_draw(prop, n, launches=8):
    """Average duration of the fused draw kernel on ``n`` rows (HIP events on the launch stream)."""
    buf = prop._get_buffers(n, prop.poolsize)
    prepared = prop._prepare_flow()
    log_p = buf["log_p"] if prop._prior_kind_dev is not None else None
    # (warm state: the first ~100 ms of kernels after an idle spell run ~8 % slower - the clocks ramp)
    for _ in range(max(2, min(40, int(4e7 // max(n, 1))))):
        prop._launch_draw(n, None, 0, prop.r, buf["x64"], buf["log_q"], None, log_prior=log_p, prepared=prepared)
    torch.cuda.synchronize()
    events = []
    for i in range(launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        prop._launch_draw(n, None, (i + 1) * n, prop.r, buf["x64"], buf["log_q"], None, log_prior=log_p, prepared=prepared)
        b.record()
        events.append((a, b))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in events]
    time_draw.last = dict(mean=float(np.mean(ms)), median=float(np.median(ms)), min=float(np.min(ms)), launches=launches)
    return float(np.mean(ms))


def live_points(model, nlive, n_prior, rng):
    pts = model.new_point(n_prior, rng=rng)
    logl = model.log_likelihood(pts)
    pts["logL"] = logl
    return pts[np.argsort(logl)[-nlive:]]


def run(tag, model, flow_config, nlive, drawsize, steps=5, cpu_draw=50_000, bound="mfma"):
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
        # the dominant kernel by itself: x as float64 + log_q (+ the per-row log prior of a separable prior) out,
        # nothing in (the latents are generated on-chip)
        precision = prop.flow.model.effective_precision
        kernel_ms = time_draw(prop, drawsize)
        bytes_per_row = 8 * model.dims + 8 + (8 if prop._prior_kind_dev is not None else 0)
        roofline = kernel_roofline(prop.flow.model, precision, drawsize, kernel_ms, bytes_per_row, bound)
        roofline["kernel"] = "nsa_flow_kernel (fused draw + inverse flow + rescale + bounds + log_q)"
        roofline["passes_per_populate"] = prop_n / drawsize / steps
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
        "precision": precision, "probe_rel_errs": prop.flow.model.probe_errors,
        "roofline": roofline,
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
            # the grouped launch by itself on device tensors (HIP events), against both roofs
            from nessai_amd.flows.base import log_prob_many

            xd = torch.from_numpy(x).to(torch.float32).to(ifm.models[0]._device())
            log_prob_many(ifm.models, xd)
            torch.cuda.synchronize()
            ev = []
            for _ in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                log_prob_many(ifm.models, xd)
                b.record()
                ev.append((a, b))
            torch.cuda.synchronize()
            k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
            roof = kernel_roofline(ifm.models[0], ifm.models[0].effective_precision, n * target, k_ms,
                                   4 * dims / target + 4, "mfma")
            roof["kernel"] = f"nsa_flow_kernel, forward + log_prob, grouped over {target} flows (nsa_flow_run_many)"
            torch.set_num_threads(1)
            ofm = OP.OracleFlowModel(dict(n_inputs=dims, n_neurons=128, n_blocks=6, n_layers=2, ftype="nsf"))
            ofm.model.load_state_dict({k: v.cpu() for k, v in ifm.models[0].state_dict().items()})
            t = time.perf_counter()
            ref = ofm.log_prob(x)
            cpu_one = time.perf_counter() - t
            err = float(np.max(np.abs(lp[:, 0] - ref) / np.maximum(np.abs(ref), 1.0)))
            print(json.dumps({"config": "C4 log_prob_all 64-D NSF", "n_flows": target, "rows": n,
                              "precision": ifm.models[0].effective_precision, "roofline": roof,
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
            prop.update_log_q(stored, np.zeros((stored.size, target)))  # (the first call of a level packs the image)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                prop.update_log_q(stored, np.zeros((stored.size, target)))
            torch.cuda.synchronize()
            du = (time.perf_counter() - t) / 3
            # the dominant kernel of a draw: the grouped forward + log_prob of the drawn rows under every stored
            # flow (nsa_flow_run_many), by itself on device tensors of the draw's size (1.01 n rows,
            # proposal/importance.py:469) - HIP events, against both roofs
            from nessai_amd.flows.base import log_prob_many

            models = list(prop.flow.models)
            n_rows = int(1.01 * nlive)
            xd = torch.randn(n_rows, dims, device=models[0]._device())
            log_prob_many(models, xd)
            torch.cuda.synchronize()
            ev = []
            for _ in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                log_prob_many(models, xd)
                b.record()
                ev.append((a, b))
            torch.cuda.synchronize()
            k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
            roof = kernel_roofline(models[0], models[0].effective_precision, n_rows * target, k_ms,
                                   4 * dims / target + 4, "mfma")
            roof["kernel"] = (f"nsa_flow_kernel, forward + log_prob of the {n_rows} drawn rows, grouped over {target} flows "
                              "(nsa_flow_run_many): a small batch - 158 tiles per flow - i.e. latency, not throughput")
            print(json.dumps({"config": "C4 ImportanceFlowProposal.draw, 64-D correlated Gaussian", "n_flows": target,
                              "nlive": nlive, "draw_ms": 1e3 * dt, "points_per_s": nlive / dt,
                              "update_log_q_ms_20000_rows": 1e3 * du, "roofline": roof}), flush=True)


if __name__ == "__main__":
    if "--c4" in sys.argv:  # config 4 only
        run_c4()
        run_c4_draw()
        sys.exit(0)
    # the fused affine flow of config 2 is HBM-bound by contract (132-136 B per row against
    # 41 kFLOP); the spline flows are MFMA-bound
    run("C2 16-D Gaussian RealNVP (LU+BN), drawsize=nlive", GaussianModel(16),
        dict(ftype="realnvp", n_neurons=32, n_blocks=4, n_layers=2), 2000, 2000, cpu_draw=2000, bound="hbm")
    run("C2 16-D Gaussian RealNVP (LU+BN), drawsize=1e6", GaussianModel(16),
        dict(ftype="realnvp", n_neurons=32, n_blocks=4, n_layers=2), 2000, 1_000_000, cpu_draw=100_000, bound="hbm")
    run("C2 16-D Gaussian RealNVP (MLP conditioner), drawsize=1e6", GaussianModel(16),
        dict(ftype="realnvp", n_neurons=32, n_blocks=4, n_layers=2, net="mlp"), 2000, 1_000_000, cpu_draw=100_000,
        bound="hbm")
    run("C5 15-D GW surrogate NSF", GWSurrogateModel(),
        dict(ftype="nsf", n_neurons=30, n_blocks=4, n_layers=2), 4000, 1_000_000, cpu_draw=100_000)
    run_c4()
    run_c4_draw()
