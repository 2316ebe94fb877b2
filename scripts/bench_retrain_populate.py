#!/usr/bin/env python
"""What the FIRST populate() after a change of the weights costs (a nested-sampling run retrains the flow every
~nlive iterations, i.e. about once per populate): the image of the flow must be rebuilt for every precision the
"auto" policy probes.  Config 3, trained weights; the weights are nudged in place before every call.  Compares
the rebuild on the device (nsa_repack_image) with the host pack (NESSAI_AMD_DEVICE_REPACK=0) and with a
populate on unchanged weights.  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def run(prop, nudge, calls=6):
    out = []
    for _ in range(calls):
        if nudge:
            with torch.no_grad():
                for p in prop.flow.model.parameters():
                    p.mul_(1.0 + 1e-6)
        torch.cuda.synchronize()
        t = time.perf_counter()
        prop.populate(None, n_samples=bench.NLIVE, max_samples=10**12)
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t))
    return float(np.median(out[1:]))


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    result = {}
    for mode in ("device", "host"):
        os.environ["NESSAI_AMD_DEVICE_REPACK"] = "1" if mode == "device" else "0"
        prop = bench.build_proposal(0, 1, device)
        for _ in range(2):
            prop.populate(None, n_samples=bench.NLIVE, max_samples=10**12)
        result[f"populate_ms_unchanged_weights_{mode}"] = run(prop, nudge=False)
        result[f"populate_ms_after_weight_change_{mode}_repack"] = run(prop, nudge=True)
        result[f"precision_{mode}"] = prop.flow.model.effective_precision
    result["rebuild_cost_ms_device"] = (result["populate_ms_after_weight_change_device_repack"]
                                        - result["populate_ms_unchanged_weights_device"])
    result["rebuild_cost_ms_host"] = (result["populate_ms_after_weight_change_host_repack"]
                                      - result["populate_ms_unchanged_weights_host"])
    print(json.dumps(result))


if __name__ == "__main__":
    main()
