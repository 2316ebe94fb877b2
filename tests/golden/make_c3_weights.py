"""Build the "realistic" weights of the headline config.

32-D Rosenbrock, NeuralSplineFlow(D=32, H=64, 6 coupling layers, 2 residual
blocks, 8 bins), trained with the ORACLE's CPU trainer (the reference schedule:
AdamW 1e-3, batch 1000, val 0.1, <=500 epochs, patience 20, clip 5.0) on 2000
live points = the top-logL 2000 of 2e5 prior draws, seed 1451.  The z-score
constants are the mean/std of those live points.

Output: tests/golden/c3_trained.npz  (state_dict arrays + live points stats)
Run from the repo root:  python tests/golden/make_c3_weights.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.proposal import OracleFlowModel, RosenbrockModel, empty_structured_array  # noqa: E402

SEED = 1451
D = 32


def main():
    torch.manual_seed(SEED)
    torch.set_num_threads(4)
    rng = np.random.default_rng(SEED)
    model = RosenbrockModel(D)
    n_prior = 200_000
    draws = rng.uniform(-5.0, 5.0, size=(n_prior, D))
    pts = empty_structured_array(n_prior, model.names)
    for i, n in enumerate(model.names):
        pts[n] = draws[:, i]
    logl = model.log_likelihood(pts)
    keep = np.argsort(logl)[-2000:]
    live = draws[keep]
    shift = live.mean(axis=0)
    scale = live.std(axis=0)
    x_prime = (live - shift) / scale

    fm = OracleFlowModel(
        dict(n_inputs=D, n_neurons=64, n_blocks=6, n_layers=2, ftype="nsf"), rng=rng
    )
    t = time.time()
    history = fm.train(x_prime)
    print(f"trained {len(history['loss'])} epochs in {time.time() - t:.1f}s; "
          f"loss {history['loss'][-1]:.4f} val {min(history['val_loss']):.4f}")
    out = {"state/" + k: v.numpy() for k, v in fm.model.state_dict().items()}
    out["live_points"] = live
    out["shift"] = shift
    out["scale"] = scale
    out["loss"] = np.asarray(history["loss"])
    out["val_loss"] = np.asarray(history["val_loss"])
    path = os.path.join(ROOT, "tests", "golden", "c3_trained.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
