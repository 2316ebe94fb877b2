#!/usr/bin/env python
"""Randomised sweep of the draw kernel's epilogue over EVERY reparameterisation kind of the registry, mixed at random in
one proposal: each parameter of a random box model gets a random registry name (z-score and its logit / log /
Gaussian-CDF variants, rescale-to-bounds and its logit / log / offset forms, null, the angle kinds, to-cartesian with
its three modes, dequantise and its logit form; pairs of parameters an angle-pair of either convention), the groups are
listed in a random order, optionally applied in reverse.  The fused kernel's ``backward_pass`` on injected latents is
compared with the float64 host table (which the golden tests pin bit for bit to the reference's ``FlowProposal``)
applied to the flow's own x' (``flow.inverse`` on the device, float32): the kept rows, every column of x - auxiliary
radii included - and log q with its row-dependent Jacobian terms.

    python scripts/fuzz_reparam_kinds.py [--iterations 60] [--seed 0]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nessai_amd.model import UniformBoxModel  # noqa: E402
from nessai_amd.proposal import FlowProposal  # noqa: E402

# name -> (bounds, how to draw live points inside them)
UNIT, POS, WIDE = [0.0, 1.0], [0.1, 9.0], [-4.0, 6.0]
KINDS = {
    "z-score": WIDE, "zscore-logit": UNIT, "log-z-score": POS, "z-score-inv-gaussian-cdf": UNIT, "zscore-gaussian-cdf": WIDE,
    "default": WIDE, "logit": WIDE, "log-rescale": WIDE, "offset": WIDE, "null": [-1.0, 1.0], "angle-sine": [0.0, np.pi],
    "angle": [0.0, 2 * np.pi], "angle-pi": [-np.pi / 2, np.pi / 2], "periodic": [1.0, 4.0], "angle-2pi": [-np.pi, np.pi],
    "to-cartesian": WIDE, "dequantise": [0.0, 6.0], "dequantise-logit": [2.0, 5.0],
}
MODES = ["split", "duplicate", "half"]


class Box(UniformBoxModel):
    def __init__(self, names, bounds, discrete):
        self.names = list(names)
        self.bounds = {n: list(bounds[n]) for n in names}
        self.discrete = tuple(discrete)

    def new_point(self, N=1, rng=None):
        x = super().new_point(N, rng=rng)
        for p in self.discrete:
            x[p] = np.floor(x[p])
        return x

    def log_likelihood(self, x):
        return np.zeros(x.size)


def run(iterations, seed, verbose=True):
    """The sweep; returns (summary, failures)."""
    g = np.random.default_rng(seed)
    bad, worst = [], dict(x=0.0, log_q=0.0)
    for it in range(iterations):
        n_single = int(g.integers(2, 8))
        names, bounds, config, discrete = [], {}, [], []
        for i in range(n_single):
            kind = str(g.choice(list(KINDS)))
            n = f"p{i}"
            names.append(n)
            bounds[n] = KINDS[kind]
            spec = kind
            if kind == "to-cartesian":
                spec = {"reparameterisation": kind, "mode": str(g.choice(MODES))}
            if kind.startswith("dequantise"):
                discrete.append(n)
            config.append((n, spec))
        for j in range(int(g.integers(0, 3))):  # angle pairs, either convention, either listing order
            h, v = f"h{j}", f"v{j}"
            radec = bool(g.random() < 0.5)
            bounds[h] = [0.0, 2 * np.pi] if g.random() < 0.5 else [-np.pi, np.pi]
            bounds[v] = [-np.pi / 2, np.pi / 2] if radec else [0.0, np.pi]
            pair = [h, v] if g.random() < 0.5 else [v, h]
            pos = int(g.integers(0, len(names) + 1))  # anywhere in the model's order, possibly apart
            names.insert(pos, pair[0])
            names.insert(int(g.integers(pos + 1, len(names) + 1)), pair[1])
            config.append((pair[0], {"reparameterisation": "angle-pair", "parameters": pair}))
        order = g.permutation(len(config))
        reparam = {config[i][0]: config[i][1] for i in order}
        reverse = bool(g.random() < 0.3)
        desc = dict(case=it, names=names, reparameterisations={k: (v if isinstance(v, str) else dict(v)) for k, v in reparam.items()},
                    reverse=reverse)
        model = Box(names, bounds, discrete)
        torch.manual_seed(int(g.integers(1, 10**6)))
        try:
            prop = FlowProposal(model, rng=np.random.default_rng(it), poolsize=50, drawsize=400, device_seed=1,
                                flow_config=dict(ftype=str(g.choice(["nsf", "realnvp"])), n_neurons=int(g.choice([8, 16, 32])),
                                                 n_blocks=int(g.integers(1, 4)), n_layers=1, batch_norm_between_layers=False,
                                                 linear_transform="permutation"),
                                reparameterisations=reparam, reverse_reparameterisations=reverse)
            prop.initialise()
        except Exception as e:  # noqa: BLE001  (a refusal is a failure here: every name is supported)
            bad.append(dict(kind="construction", error=repr(e), **desc))
            if verbose:
                print("FAIL", json.dumps(bad[-1], default=str), flush=True)
            continue
        live = model.new_point(400, rng=np.random.default_rng(1000 + it))
        for n in names:  # away from the edges: finite x' for the logit / log / inverse-CDF kinds
            lo, hi = bounds[n]
            if n not in discrete:
                live[n] = lo + (hi - lo) * (0.05 + 0.9 * np.random.default_rng(it + hash(n) % 1000).random(live.size))
        prop.check_state(live)
        table = prop._reparam
        dims = prop.dims
        z = g.normal(size=(3000, dims)).astype(np.float32).astype(np.float64) * 0.7
        x, log_q, z_kept = prop.backward_pass(z, return_z=True)
        # the same through the host table: x' from the flow alone (device, float32), then float64 on the host
        with torch.no_grad():
            zt = torch.from_numpy(z).to(torch.float32).to(prop.device)
            xp_t, log_det = prop.flow.model.inverse(zt)
            log_pz = -0.5 * (zt.double() ** 2).sum(dim=1) - 0.5 * dims * np.log(2 * np.pi)
            log_q_flow = (log_pz - log_det.double()).cpu().numpy()
            xp = xp_t.double().cpu().numpy()
        xps = np.zeros(len(xp), dtype=[(n, "f8") for n in table.prime_names] + [("logP", "f8"), ("logL", "f8"), ("it", "i4")])
        for i, n in enumerate(table.prime_names):
            xps[n] = xp[:, i]
        with np.errstate(all="ignore"):
            x_h, log_j = table.inverse_rescale(xps)
            ok_rows = np.isfinite(log_q_flow) & model.in_bounds(x_h)
        log_q_h = log_q_flow - log_j
        if len(x) != int(ok_rows.sum()):
            if abs(len(x) - int(ok_rows.sum())) <= 3:  # a row on a bound within float32 rounding
                continue
            bad.append(dict(kind="kept rows", got=len(x), host=int(ok_rows.sum()), **desc))
            if verbose:
                print("FAIL", json.dumps(bad[-1], default=str), flush=True)
            continue
        if not np.array_equal(z_kept, z[ok_rows]):
            continue  # (the same count, another row on a bound)
        e_x = 0.0
        for n in table.x_names:
            a, b = np.asarray(x[n], dtype=np.float64), np.asarray(x_h[n][ok_rows], dtype=np.float64)
            fin = np.isfinite(b)
            if n in discrete:
                e = float(np.mean(a[fin] != b[fin]))  # whole numbers: the fraction that differs (a step next to rounding)
                tol = 0.01
            else:
                e = float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin])))) if fin.any() else 0.0
                tol = 1e-5
            e_x = max(e_x, e / tol)
        b = log_q_h[ok_rows]
        fin = np.isfinite(b) & np.isfinite(log_q)
        e_q = float(np.max(np.abs(log_q[fin] - b[fin]) / np.maximum(1.0, np.abs(b[fin])))) if fin.any() else 0.0
        worst["x"], worst["log_q"] = max(worst["x"], e_x), max(worst["log_q"], e_q)
        if e_x > 1.0 or e_q > 1e-5:
            bad.append(dict(kind="values", x_over_tol=e_x, log_q_rel=e_q, **desc))
            if verbose:
                print("FAIL", json.dumps(bad[-1], default=str), flush=True)
    return dict(iterations=iterations, failures=len(bad), worst_x_over_tol=worst["x"], worst_log_q_rel=worst["log_q"]), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    summary, bad = run(args.iterations, args.seed)
    print(json.dumps(summary))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
