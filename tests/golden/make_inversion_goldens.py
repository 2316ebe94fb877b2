#!/usr/bin/env python
"""Golden vectors for the boundary-inversion reparameterisations, from the REFERENCE's own classes.

Run in the build container only (needs /root/reference):

    python tests/golden/make_inversion_goldens.py

A 5-parameter model: ``a`` piles up at its lower bound (registry entry ``inversion``: mirrored by flipping
half of the rows with the reparameterisation's generator), ``c`` at its upper bound (``inversion-duplicate``:
every row doubled), ``d`` sits in the middle of its range (``inversion`` with edge detection: no edge, the
plain [-1, 1] map), ``b`` is z-scored, ``e`` rescaled to bounds.  Stores the live points, the outputs of
``CombinedReparameterisation.reparameterise`` (also with ``compute_radius=True``, which duplicates every
mirrored parameter), the detected edges, and ``inverse_reparameterise`` of fresh x' values of both signs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import reference_harness  # noqa: E402

reference_harness.install()

from nessai.livepoint import empty_structured_array  # noqa: E402
from nessai.reparameterisations import CombinedReparameterisation, get_reparameterisation  # noqa: E402

NAMES = ["a", "b", "c", "d", "e"]
BOUNDS = {"a": [0.0, 5.0], "b": [-4.0, 4.0], "c": [2.0, 9.0], "d": [-1.0, 3.0], "e": [10.0, 20.0]}
SPEC = [("inversion", ["a"]), ("z-score", ["b"]), ("inversion-duplicate", ["c"]), ("inversion", ["d"]),
        ("rescaletobounds", ["e"])]
SEED = 20260930


def build(rng):
    combined = CombinedReparameterisation(initial_parameters=list(NAMES))
    for name, params in SPEC:
        cls, kw = get_reparameterisation(name)
        kw = dict(kw or {})
        kw["prior_bounds"] = {p: np.array(BOUNDS[p]) for p in params}
        combined.add_reparameterisations(cls(parameters=params, rng=rng, **kw))
    return combined


def main():
    data = np.random.default_rng(SEED)
    n = 400
    live = empty_structured_array(n, names=NAMES)
    live["a"] = np.abs(data.normal(0.0, 0.6, size=n))            # against the lower bound 0
    live["b"] = data.normal(0.3, 1.1, size=n)
    live["c"] = 9.0 - np.abs(data.normal(0.0, 0.8, size=n))      # against the upper bound 9
    live["d"] = np.clip(data.normal(1.0, 0.25, size=n), -1, 3)   # in the middle: no edge
    live["e"] = data.uniform(11.0, 19.0, size=n)
    combined = build(np.random.default_rng(SEED + 1))
    combined.update(live)

    x_prime = empty_structured_array(n, names=combined.prime_parameters)
    _, x_prime, log_j = combined.reparameterise(live.copy(), x_prime, np.zeros(n))
    edges = {}
    for r in combined.values():
        if getattr(r, "_edges", None):
            edges.update(r._edges)

    m = 64  # radius computation: every mirrored parameter duplicates (edges stay as detected)
    xr = empty_structured_array(m, names=combined.prime_parameters)
    _, xr, log_j_r = combined.reparameterise(live[:m].copy(), xr, np.zeros(m), compute_radius=True)

    k = 300
    xp_in = empty_structured_array(k, names=combined.prime_parameters)
    for pp in combined.prime_parameters:
        xp_in[pp] = data.normal(0.0, 0.8, size=k)
    x_out = empty_structured_array(k, names=NAMES)
    x_out, _, log_j_inv = combined.inverse_reparameterise(x_out, xp_in.copy(), np.zeros(k))

    out = {"n": n, "m": m, "k": k, "seed": SEED,
           "edges": np.array([str(edges.get(p)) for p in NAMES])}
    for p in NAMES:
        out[f"live/{p}"] = np.array(live[p], copy=True)
        out[f"x_prime/{p}_prime"] = np.array(x_prime[p + "_prime"], copy=True)
        out[f"x_radius/{p}_prime"] = np.array(xr[p + "_prime"], copy=True)
        out[f"xp_in/{p}_prime"] = np.array(xp_in[p + "_prime"], copy=True)
        out[f"x_out/{p}"] = np.array(x_out[p], copy=True)
    out["log_j"], out["log_j_radius"], out["log_j_inv"] = log_j, log_j_r, log_j_inv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "inversion_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "rows", x_prime.size, xr.size, "edges", out["edges"])


if __name__ == "__main__":
    main()
