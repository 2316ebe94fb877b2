#!/usr/bin/env python
"""Where the device loses the reference's invertibility bar on one case (tests/test_gpu_reference_bars.py):
forward and inverse log-Jacobians of the device and of the float32 oracle against the float64 oracle, row by row."""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(kind="nsf", dim=2, seed=2):
    from nessai_amd.flows import NeuralSplineFlow, RealNVP
    from oracle import flows as O

    P, Oc = {"realnvp": (RealNVP, O.RealNVP), "nsf": (NeuralSplineFlow, O.NeuralSplineFlow)}[kind]
    torch.manual_seed(seed)
    ref = Oc(dim, 8, 2, 2, batch_norm_between_layers=False).eval()
    flow = P(dim, 8, 2, 2, batch_norm_between_layers=False)
    flow.load_state_dict(ref.state_dict())
    flow.to("cuda:0")
    flow.eval()
    r64 = copy.deepcopy(ref).double()
    x = torch.randn(1000, dim)
    for precision in ("f32", "f16x3"):
        flow.precision = precision
        with torch.inference_mode():
            z, lj = flow.forward(x.cuda())
            xo, lji = flow.inverse(z)
            z, lj, xo, lji = z.cpu(), lj.cpu(), xo.cpu(), lji.cpu()
            z_o, lj_o = ref.forward(x)
            xo_o, lji_o = ref.inverse(z_o)
            z64, lj64 = r64.forward(x.double())
            _, lji64_at_dev_z = r64.inverse(z.double())
            _, lji64_at_o_z = r64.inverse(z_o.double())
        bad = int(torch.argmax((lj + lji).abs()))
        print(f"{precision}: worst row {bad}: x = {x[bad].tolist()}, |lj + lji| = {float((lj + lji).abs()[bad]):.3e}")
        print(f"   device : lj - lj64 = {float(lj[bad] - lj64[bad]):+.3e}   lji - lji64(z_dev) = {float(lji[bad] - lji64_at_dev_z[bad]):+.3e}"
              f"   lj64 + lji64(z_dev) = {float(lj64[bad] + lji64_at_dev_z[bad]):+.3e}   |z - z64| = {float((z[bad] - z64[bad]).abs().max()):.3e}")
        print(f"   oracle : lj - lj64 = {float(lj_o[bad] - lj64[bad]):+.3e}   lji - lji64(z_o)   = {float(lji_o[bad] - lji64_at_o_z[bad]):+.3e}"
              f"   lj64 + lji64(z_o)   = {float(lj64[bad] + lji64_at_o_z[bad]):+.3e}   |z - z64| = {float((z_o[bad] - z64[bad]).abs().max()):.3e}")
        print(f"   max over rows: device |lj - lj64| {float((lj - lj64).abs().max()):.3e}, |lji - lji64(z_dev)| {float((lji - lji64_at_dev_z).abs().max()):.3e};"
              f" oracle {float((lj_o - lj64).abs().max()):.3e}, {float((lji_o - lji64_at_o_z).abs().max()):.3e}")


if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["nsf"]), *[int(a) for a in sys.argv[2:4]])
