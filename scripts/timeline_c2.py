#!/usr/bin/env python
"""Per-section cycle budget of the draw kernel on BASELINE config 2 (the reference's default RealNVP) in the
weights-resident geometry, from a -DNSA_TIMELINE build:

    cd nessai_amd/csrc && make -j8 BUILD=build_tl TARGET=../../variants/libnessai_amd_tl.so \\
        HTS=1 TRANSFORMS=2_0 EXTRA=-DNSA_TIMELINE
    NESSAI_AMD_LIBRARY=variants/libnessai_amd_tl.so python scripts/timeline_c2.py [geometry ...]     # GPU box

Every wave adds the shader-clock cycles it spent in each section of its tile loop to a device array
(csrc/nsa_flow_kernel.h, NSA_TL); printed per 32-row tile, averaged over waves.  With unsynchronised waves on
a VALU-bound SIMD a section's share of the wave cycles is close to its share of the vector instructions."""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_configs import live_points, time_draw  # noqa: E402
from nessai_amd.model import GaussianModel  # noqa: E402
from nessai_amd.proposal import FlowProposal  # noqa: E402

SLOTS = ["prologue (norm, log p(z))", "layer0 gemm", "layer0 tail", "hidden gemm", "hidden tail", "chunk gemm", "transform",
         "-", "f16 range check", "epilogue", "other ops (LU / batch norm)", "loop top", "fill / latent draw", "bias read"]


def main():
    geos = sys.argv[1:] or ["4x1"]
    model = GaussianModel(16)
    rng = np.random.default_rng(1451)
    torch.manual_seed(1451)
    live = live_points(model, 2000, 80_000, rng)
    n = 1_000_000
    with tempfile.TemporaryDirectory() as tmp:
        prop = FlowProposal(model, rng=rng, poolsize=2000, drawsize=n, output=tmp, device_seed=1451,
                            flow_config=dict(ftype="realnvp", n_neurons=32, n_blocks=4, n_layers=2))
        prop.initialise()
        prop.train(live)
        prop.populate(None, n_samples=2000, max_samples=10**12)
        for geo in geos:
            os.environ["NESSAI_AMD_RESIDENT"] = geo
            os.environ.pop("NSA_TL_PTR", None)
            time_draw(prop, n, launches=2)
            dbg = torch.zeros(16, dtype=torch.int64, device="cuda:0")
            os.environ["NSA_TL_PTR"] = str(dbg.data_ptr())
            ms = time_draw(prop, n, launches=4)
            torch.cuda.synchronize()
            os.environ.pop("NSA_TL_PTR", None)
            v = dbg.cpu().numpy().astype("float64")
            tiles, waves = v[14], v[15]
            per_tile = v[:14] / tiles
            total = per_tile.sum()
            print(json.dumps(dict(geometry=geo, ms_avg=ms, tiles_per_wave=tiles / waves, cycles_per_tile=total,
                                  sections={k: round(float(x), 1) for k, x in zip(SLOTS, per_tile)},
                                  share={k: round(float(x / total), 4) for k, x in zip(SLOTS, per_tile)})), flush=True)


if __name__ == "__main__":
    main()
