#!/usr/bin/env python
"""FlowModel.train step time on BASELINE config 2's flow (16-D RealNVP, 4 coupling layers, 32-wide conditioner, the
reference's defaults: LU-linear layers and batch norm between the couplings), 2000 live points.  One JSON line.

    python scripts/bench_train_c2.py [--epochs 40] [--plain]     (--plain: permutations, no batch norm)"""
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
    ap.add_argument("--plain", action="store_true")
    ap.add_argument("--config", choices=("c2", "c4", "c5"), default="c2",
                    help="c4: 64-D NSF, 128 neurons, 6 layers, 5000 points; c5: 15-D NSF, 30 neurons, 4 layers, 4000 points")
    args = ap.parse_args()
    rng = np.random.default_rng(7)
    if args.config == "c4":
        x = rng.normal(size=(5000, 64)).astype(np.float64)
        cfg = dict(n_inputs=64, n_neurons=128, n_blocks=6, n_layers=2, ftype="nsf")
    elif args.config == "c5":
        x = rng.normal(size=(4000, 15)).astype(np.float64)
        cfg = dict(n_inputs=15, n_neurons=30, n_blocks=4, n_layers=2, ftype="nsf")
    else:
        x = rng.normal(size=(2000, 16)).astype(np.float64)
        cfg = dict(n_inputs=16, n_neurons=32, n_blocks=4, n_layers=2, ftype="realnvp")
    if args.plain:
        cfg.update(linear_transform="permutation", batch_norm_between_layers=False)
    from nessai_amd.flowmodel import FlowModel

    torch.manual_seed(7)
    with tempfile.TemporaryDirectory() as tmp:
        fm = FlowModel(flow_config=cfg, training_config=dict(max_epochs=args.epochs, patience=10**6), output=tmp,
                       rng=np.random.default_rng(7))
        fm.initialise()
        fm.train(x, max_epochs=2)
        fm.reset_model()
        torch.cuda.synchronize()
        t = time.perf_counter()
        hist = fm.train(x, max_epochs=args.epochs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    n_train = int(round(0.9 * len(x)))
    steps = -(-n_train // 1000) * len(hist["loss"])
    print(json.dumps({"metric": f"FlowModel.train optimiser step time, config {args.config[1]} flow" + (" (permutations, no batch norm)" if args.plain else ""),
                      "rows": len(x), "steps_per_epoch": -(-n_train // 1000),
                      "gpu_ms_per_step": 1e3 * dt / steps, "epochs": len(hist["loss"]), "final_loss": hist["loss"][-1]}))


if __name__ == "__main__":
    main()
