#!/usr/bin/env python
"""Round 5: time of the LEAN draw kernel (config 3, 1e6 rows; the head of a lean pass, nsa_populate_draw_weights) for
the library named by NESSAI_AMD_LIBRARY, in the shipped geometry, with a checksum of the weights.  One JSON line.
Environment knobs are reported: NESSAI_AMD_PRIO, NESSAI_AMD_SKEW, NESSAI_AMD_OCC."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    torch.cuda.set_device(0)
    prop = bench.build_proposal(0, 1, torch.device("cuda", 0))
    if "--no-populate" in sys.argv:  # ablation builds: their rows are garbage, a populate would never fill its pool
        prop.last_populate_lean = True
    else:
        prop.populate(None, n_samples=bench.NLIVE, max_samples=30 * bench.DRAWSIZE)
    for _ in range(3):  # warm clocks
        bench.time_draw_kernel(prop, launches=10)
    runs = [bench.time_draw_kernel(prop, launches=10) for _ in range(3)]
    buf = prop._get_buffers(bench.DRAWSIZE, bench.NLIVE)
    torch.cuda.synchronize()
    w = buf["log_w"][: bench.DRAWSIZE]
    kept = torch.isfinite(w)
    print(json.dumps(dict(library=os.path.basename(os.environ.get("NESSAI_AMD_LIBRARY", "libnessai_amd.so")),
                          prio=os.environ.get("NESSAI_AMD_PRIO", ""), skew=os.environ.get("NESSAI_AMD_SKEW", ""),
                          occ=os.environ.get("NESSAI_AMD_OCC", ""), lean=bool(prop.last_populate_lean),
                          ms_avg=[round(r[0], 4) for r in runs], ms_min=round(min(r[1] for r in runs), 4),
                          geometry=bench.LAST_GEOMETRY, kept=int(kept.sum().item()),
                          log_w_sum=float(w[kept].sum().item()), status=prop.flow.model._status.tolist())), flush=True)


if __name__ == "__main__":
    main()
