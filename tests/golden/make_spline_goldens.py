#!/usr/bin/env python
"""Golden vectors for the rational-quadratic spline from an INDEPENDENT port of the nflows code.

The reference's spline arithmetic lives in ``glasflow.nflows`` (absent here).  HuggingFace
``transformers`` (installed: 5.15.0) ships its own port of the same nflows functions for the VITS
model: ``transformers.models.vits.modeling_vits._unconstrained_rational_quadratic_spline`` (linear
tails, the derivative padding constant ``log(exp(1 - min_derivative) - 1)``, searchsorted with the
1e-6 bump).  This script evaluates it on fixed inputs, forward and inverse, for the bin counts the
kernels are instantiated for, and stores inputs and outputs in ``spline_hf_goldens.npz``.

    python tests/golden/make_spline_goldens.py
"""
import os

import numpy as np
import torch
from transformers.models.vits import modeling_vits as V


def main():
    out = {}
    for K in (4, 8, 10, 16):
        g = torch.Generator().manual_seed(100 + K)
        n = 1500
        x = torch.randn(n, generator=g) * 3.0
        x[:8] = torch.tensor([-5.0, 5.0, -4.999999, 4.999999, 0.0, -7.5, 7.5, 1e-7])
        uw = torch.randn(n, K, generator=g) * 1.5
        uh = torch.randn(n, K, generator=g) * 1.5
        ud = torch.randn(n, K - 1, generator=g) * 2.0
        out[f"k{K}/x"], out[f"k{K}/uw"], out[f"k{K}/uh"], out[f"k{K}/ud"] = (t.numpy() for t in (x, uw, uh, ud))
        for tag, reverse in (("fwd", False), ("inv", True)):
            y, lad = V._unconstrained_rational_quadratic_spline(
                x, uw, uh, ud, reverse=reverse, tail_bound=5.0, min_bin_width=1e-3, min_bin_height=1e-3,
                min_derivative=1e-3)
            out[f"k{K}/{tag}/y"], out[f"k{K}/{tag}/logabsdet"] = y.numpy(), lad.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spline_hf_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays")


if __name__ == "__main__":
    main()
