#!/usr/bin/env python
"""Host cost of the SHARDED populate loop (one Python iteration per pass: head launch, all-reduce(max),
tail, all-reduce(sum), commit) measured on one GPU: a one-rank RCCL group, the per-GPU batch of an 8-rank
strong-scaling run (125 000 rows).  Prints the wall time per pass of the sharded loop beside the
device-side loop of the unsharded path (nsa_populate_passes) on the same batch - the difference is what
the per-pass host work and the two collectives cost when they are not hidden behind the kernels.

    python scripts/bench_sharded_host.py [rows per pass]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(prop, steps=10):
    prop.populate(None, n_samples=bench.NLIVE, max_samples=10**12)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    passes = 0
    for _ in range(steps):
        prop.populate(None, n_samples=bench.NLIVE, max_samples=10**12)
        passes += prop.n_proposed // prop.drawsize
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / passes * 1e3, passes / steps


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    device = torch.device("cuda", 0)
    bench.DRAWSIZE = rows
    out = {"rows_per_pass": rows}
    for shard in (False, True):
        prop = bench.build_proposal(0, 2 if shard else 1, device)  # world > 1 only switches shard=True on
        ms, per_step = run(prop)
        out["sharded_loop" if shard else "device_loop"] = {"ms_per_pass": round(ms, 4), "passes_per_populate": per_step}
    print(json.dumps(out))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
