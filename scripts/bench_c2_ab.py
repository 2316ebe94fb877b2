#!/usr/bin/env python
"""Same-box A/B of the config-2 draw kernel between source trees (round 6, verdict item 2: 0.280 -> 0.368 ms).

  python scripts/bench_c2_ab.py --root . --root variants/r04 [--rounds 3]

Each tree is imported in its OWN subprocess (its package, its library); the trees are timed alternately, `rounds`
times, on the same box: the full draw launch (x as float64 + log_q out: what `scripts/bench_configs.py` times) at
1e6 rows of config 2 (RealNVP 16-D, LU + batch norm, 2 x 32 residual net), its MLP variant, config 5's NSF - and,
where the tree has them, the lean head (`nsa_populate_draw_weights`).  One JSON line per (tree, round).
"""
import argparse
import json
import os
import subprocess
import sys

CHILD = r'''
import json, os, sys, tempfile
import numpy as np, torch
root = os.path.abspath(sys.argv[1]); sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "scripts"))
from nessai_amd.model import GaussianModel, GWSurrogateModel
from nessai_amd.proposal import FlowProposal
from nessai_amd import lib
import bench_configs as BC

def build(model, flow_config, nlive, drawsize):
    rng = np.random.default_rng(1451); torch.manual_seed(1451)
    trained = os.environ.get("AB_TRAINED") == "1"   # as scripts/bench_configs.py: fit the best nlive of 40 nlive prior draws
    tmp = tempfile.mkdtemp()
    prop = FlowProposal(model, rng=rng, flow_config=flow_config, poolsize=nlive, drawsize=drawsize, device_seed=1451,
                        output=tmp)
    prop.initialise()
    if trained:
        prop.train(BC.live_points(model, nlive, 40 * nlive, rng))
    else:
        prop.check_state(model.new_point(nlive, rng=np.random.default_rng(1)))
    prop.populate(None, n_samples=50, max_samples=3 * drawsize)   # device tables, radius, precision probe
    return prop

def lean_ms(prop, n, launches=8):
    if not hasattr(prop, "_launch_draw_weights"):
        return None
    L = lib.load()
    log_w = torch.zeros(n, dtype=torch.float64, device=prop.device)
    key = torch.full((1,), int(L.nsa_max_key_init()), dtype=torch.int64, device=prop.device)
    buf = prop._get_buffers(n, prop.poolsize)
    log_p = buf["log_p"] if prop._prior_kind_dev is not None else None
    for _ in range(40):
        prop._launch_draw_weights(n, 0, prop.r, log_w, key, log_prior=log_p)
    torch.cuda.synchronize()
    ev = []
    for i in range(launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); prop._launch_draw_weights(n, (i + 1) * n, prop.r, log_w, key, log_prior=log_p); b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))

out = {"root": sys.argv[1], "trained": os.environ.get("AB_TRAINED") == "1"}
n = 1_000_000
for tag, model, cfg, nlive in (
    ("c2_lu_bn", GaussianModel(16), dict(ftype="realnvp", n_neurons=32, n_blocks=4, n_layers=2), 2000),
    ("c2_mlp", GaussianModel(16), dict(ftype="realnvp", n_neurons=32, n_blocks=4, n_layers=2, net="mlp"), 2000),
    ("c5_nsf", GWSurrogateModel(), dict(ftype="nsf", n_neurons=30, n_blocks=4, n_layers=2), 4000),
):
    prop = build(model, cfg, nlive, n)
    full = BC.time_draw(prop, n)
    kept = float(torch.isfinite(prop._get_buffers(n, prop.poolsize)["log_q"][:n]).double().mean())
    geo = None
    try:
        import ctypes
        g = (ctypes.c_int * 3)()
        L = lib.load(); L.nsa_last_geometry(ctypes.byref(g, 0), ctypes.byref(g, 4), ctypes.byref(g, 8)); geo = list(g) + [int(L.nsa_last_row_tiles()), int(L.nsa_last_resident())]
    except Exception as e:
        geo = repr(e)
    out[tag] = {"full_draw_ms": full, "lean_head_ms": lean_ms(prop, n), "precision": prop.flow.model.effective_precision,
                "geometry": geo, "lean_populate": getattr(prop, "last_populate_lean", None), "rows_kept": kept, "radius": float(prop.r)}
print(json.dumps(out), flush=True)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", action="append", required=True)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    for r in range(args.rounds):
        for root in args.root:
            env = dict(os.environ)
            env.pop("NESSAI_AMD_LIBRARY", None)
            tag = root
            if "@" in root:  # "TREE@LIBRARY": the tree's package with another build of its library (variants/)
                root, library = root.split("@", 1)
                env["NESSAI_AMD_LIBRARY"] = os.path.abspath(library)
            p = subprocess.run([sys.executable, "-c", CHILD, root], capture_output=True, text=True, env=env)
            root = tag
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else json.dumps({"root": root, "error": p.stderr[-800:]})
            print(json.dumps(dict(json.loads(line), round=r, root=tag)), flush=True)


if __name__ == "__main__":
    main()
