#!/usr/bin/env python
"""Draw-kernel time of config 3 at the per-GPU batch sizes of a strong-scaling run (1e6 rows over
1, 2, 4, 8 GPUs) and a few sizes in between, under each launch geometry (NESSAI_AMD_OCC) and under the
library's own choice: the check of nsa_big_batch_geometry's round-count estimate (csrc/nsa_api.hip)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    prop = bench.build_proposal(0, 1, device)
    prop.populate(None, n_samples=bench.NLIVE, max_samples=10**12)
    keep = prop.precision
    prop.precision = "bf16x3"
    try:
        for n in (62_500, 100_000, 125_000, 196_608, 250_000, 300_000, 500_000, 1_000_000):
            row = {"rows": n, "tiles": (n + 31) // 32}
            for occ in ("2", "3", "2x2", None):
                if occ is None:
                    os.environ.pop("NESSAI_AMD_OCC", None)
                else:
                    os.environ["NESSAI_AMD_OCC"] = occ
                ms, ms_min = bench._time_draw_kernel(prop, n, 8)
                g = bench.LAST_GEOMETRY
                row[occ or "library"] = round(ms_min, 4)
                if occ is None:
                    row["library_geometry"] = [g["waves_per_simd"], g["waves_per_group"], g["row_tiles_per_wave"]]
            print(json.dumps(row), flush=True)
    finally:
        prop.precision = keep
        os.environ.pop("NESSAI_AMD_OCC", None)


if __name__ == "__main__":
    main()
