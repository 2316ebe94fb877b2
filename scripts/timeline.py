#!/usr/bin/env python
"""Per-section cycle budget of the draw kernel (config 3, 1e6 rows) from a -DNSA_TIMELINE build:

    cd nessai_amd/csrc && make -j8 BUILD=build_tl TARGET=../../variants/libnessai_amd_tl.so \\
        HTS=2 TRANSFORMS=1_8 EXTRA=-DNSA_TIMELINE
    NESSAI_AMD_LIBRARY=variants/libnessai_amd_tl.so python scripts/timeline.py        # GPU box

Every wave adds the shader-clock cycles it spent in each section of its tile loop to a device
array (csrc/nsa_flow_kernel.h, NSA_TL); this prints them per 32-row tile, averaged over waves."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SLOTS = ["fill+prologue", "layer0 gemm", "layer0 tail", "hidden gemm", "hidden tail", "chunk gemm", "transform",
         "copy wait", "barrier", "epilogue", "other ops", "loop top", "copy issue", "bias read"]


def main():
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    prop = bench.build_proposal(0, 1, device)
    prop.populate(None, n_samples=bench.NLIVE, max_samples=10**12)
    for occ, wpg in (("2x2", 0), (3, 12), (2, 4)):  # "2x2": a "tile" below is a wave's PAIR of tiles
        os.environ["NESSAI_AMD_OCC"] = str(occ)
        prop.flow.model.waves_per_group = wpg
        os.environ.pop("NSA_TL_PTR", None)
        bench.time_draw_kernel(prop, launches=2, precision="bf16x3")
        dbg = torch.zeros(16, dtype=torch.int64, device=device)
        os.environ["NSA_TL_PTR"] = str(dbg.data_ptr())
        launches = 4
        ms, ms_min = bench.time_draw_kernel(prop, launches=launches, precision="bf16x3")
        torch.cuda.synchronize()
        os.environ.pop("NSA_TL_PTR", None)
        v = dbg.cpu().numpy().astype("float64")
        tiles, waves = v[14], v[15]
        per_tile = v[:14] / tiles
        total = per_tile.sum()
        print(json.dumps(dict(occ=occ, waves_per_group=wpg, ms_avg=ms, tiles_per_wave=tiles / waves, cycles_per_tile=total,
                              sections={k: round(float(x), 1) for k, x in zip(SLOTS, per_tile)},
                              share={k: round(float(x / total), 4) for k, x in zip(SLOTS, per_tile)})), flush=True)


if __name__ == "__main__":
    main()
