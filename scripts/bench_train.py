#!/usr/bin/env python
"""Secondary measurement: FlowModel.train step time, GPU kernels vs the CPU oracle.

Config 3 (32-D NSF, 6 layers, 2x64 residual conditioner, 8 bins), 2000 live points
(1800 train / 200 validation, minibatches of 1000 + 800), reference schedule.
Prints one JSON line.  `python scripts/bench_train.py [--epochs 40] [--cpu-epochs 3]`
"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--cpu-epochs", type=int, default=3)
    args = ap.parse_args()
    data = np.load(os.path.join(ROOT, "tests", "golden", "c3_trained.npz"))
    live, shift, scale = data["live_points"], data["shift"], data["scale"]
    x = (live - shift) / scale
    cfg = dict(n_inputs=32, n_neurons=64, n_blocks=6, n_layers=2, ftype="nsf")

    from nessai_amd.flowmodel import FlowModel

    torch.manual_seed(1451)
    with tempfile.TemporaryDirectory() as tmp:
        fm = FlowModel(flow_config=cfg, training_config=dict(max_epochs=args.epochs, patience=10**6),
                       output=tmp, rng=np.random.default_rng(1451))
        fm.initialise()
        fm.train(x, max_epochs=2)  # warm-up (workspace allocation, first launches)
        fm.reset_model()
        torch.cuda.synchronize()
        t = time.perf_counter()
        hist = fm.train(x, max_epochs=args.epochs)
        torch.cuda.synchronize()
        gpu_s = time.perf_counter() - t
    steps = 2 * len(hist["loss"])

    from oracle.proposal import OracleFlowModel

    torch.set_num_threads(1)
    torch.manual_seed(1451)
    ofm = OracleFlowModel(cfg, training_config=dict(max_epochs=args.cpu_epochs, patience=10**6),
                          rng=np.random.default_rng(1451))
    t = time.perf_counter()
    ohist = ofm.train(x, max_epochs=args.cpu_epochs)
    cpu_s = time.perf_counter() - t
    out = {
        "metric": "FlowModel.train optimiser step time, 32-D NSF, 2000 live points",
        "gpu_ms_per_step": 1e3 * gpu_s / steps,
        "gpu_epochs": len(hist["loss"]),
        "gpu_seconds": gpu_s,
        "gpu_final_loss": hist["loss"][-1],
        "gpu_best_val_loss": float(np.nanmin(hist["val_loss"])),
        "cpu_ms_per_step": 1e3 * cpu_s / (2 * len(ohist["loss"])),
        "cpu_threads": 1,
        "cpu_epochs": len(ohist["loss"]),
        "cpu_loss_after": ohist["loss"][-1],
        "gpu_loss_after_same_epochs": hist["loss"][len(ohist["loss"]) - 1],
    }
    out["speedup"] = out["cpu_ms_per_step"] / out["gpu_ms_per_step"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
