#!/usr/bin/env python
"""Draw-kernel time of the headline workload (config 3, 1e6 rows) under launch variants:
waves per SIMD (NESSAI_AMD_OCC) x waves per workgroup, both precisions.  One JSON line each."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

VARIANTS = [(2, 4), (3, 4), (3, 12), (2, 8), (3, 6)]
QUICK = "--quick" in sys.argv  # the three big-batch geometries, 16-bit modes only
# --precisions a,b,c overrides the list of modes
PRECISIONS = None
if "--precisions" in sys.argv:
    PRECISIONS = tuple(sys.argv[sys.argv.index("--precisions") + 1].split(","))


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    prop = bench.build_proposal(0, 1, device)
    prop.populate(None, n_samples=bench.NLIVE, max_samples=10**12)
    ref = None
    for precision in (PRECISIONS or (("bf16x3", "f16x3") if QUICK else ("bf16x3", "f16x3", "f32"))):
        for occ, wpg in ([("2x2", 0)] if "--2x2" in sys.argv else [(2, 4), (3, 12), ("2x2", 0)] if QUICK else VARIANTS + [("2x2", 0)]):
            os.environ["NESSAI_AMD_OCC"] = str(occ)
            prop.flow.model.waves_per_group = wpg
            ms, ms_min = bench.time_draw_kernel(prop, launches=8, precision=precision)
            buf = prop._get_buffers(bench.DRAWSIZE, bench.NLIVE)
            torch.cuda.synchronize()
            lq = buf["log_q"][: bench.DRAWSIZE].clone()
            kept = torch.isfinite(lq)
            checksum = float(lq[kept].sum().item())
            if precision == "bf16x3" and ref is None:
                ref = (int(kept.sum().item()), checksum)
            status = prop.flow.model._status.tolist()
            print(json.dumps(dict(skew=os.environ.get("NESSAI_AMD_SKEW", ""), status=status, library=os.path.basename(os.environ.get("NESSAI_AMD_LIBRARY", "libnessai_amd.so")), precision=precision, occ=occ, waves_per_group=wpg, ms_avg=ms, ms_min=ms_min,
                                  kept=int(kept.sum().item()), log_q_sum=checksum)), flush=True)
    os.environ.pop("NESSAI_AMD_OCC", None)


if __name__ == "__main__":
    main()
