#!/usr/bin/env python
"""Golden vectors for the per-parameter reparameterisations, from the REFERENCE's own classes.

Run in the build container only (needs /root/reference):

    python tests/golden/make_reparam_goldens.py

``nessai.reparameterisations`` imports without glasflow (the stubs of
``oracle.reference_harness`` only satisfy unrelated top-level imports).  Stores inputs and the
reference's outputs of ``CombinedReparameterisation.reparameterise`` / ``inverse_reparameterise``
for a 7-parameter model that mixes every registry entry the device epilogue supports.
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

NAMES = ["a", "b", "c", "d", "e", "f", "g"]
BOUNDS = {"a": [-3.0, 7.0], "b": [0.5, 2.5], "c": [-1.0, 1.0], "d": [10.0, 20.0], "e": [-4.0, 4.0],
          "f": [0.0, 5.0], "g": [1.0, 9.0]}
# (registry name, parameters, extra kwargs), in the order they are added
SPEC = [
    ("logit", ["b", "f"], {}),
    ("rescaletobounds", ["a"], {}),
    ("offset", ["d"], {}),
    ("null", ["c"], {}),
    ("log-rescale", ["g"], {}),
    ("z-score", ["e"], {}),
]


def build():
    combined = CombinedReparameterisation(initial_parameters=list(NAMES))
    for name, params, kwargs in SPEC:
        cls, kw = get_reparameterisation(name)
        kw = dict(kw or {})
        kw.update(kwargs)
        kw["prior_bounds"] = {p: np.array(BOUNDS[p]) for p in params}
        combined.add_reparameterisations(cls(parameters=params, **kw))
    return combined


def main():
    rng = np.random.default_rng(20260930)
    n = 257
    live = empty_structured_array(n, names=NAMES)
    for p in NAMES:
        lo, hi = BOUNDS[p]
        live[p] = lo + (hi - lo) * rng.beta(2.0, 3.0, size=n)
    combined = build()
    combined.update(live)

    x_prime = empty_structured_array(n, names=combined.prime_parameters)
    log_j = np.zeros(n)
    _, x_prime, log_j = combined.reparameterise(live.copy(), x_prime, log_j)

    # inverse on fresh x' values, including saturating logits and values outside [-1, 1]
    m = 301
    xp_in = empty_structured_array(m, names=combined.prime_parameters)
    for pp in combined.prime_parameters:
        xp_in[pp] = rng.normal(0.0, 2.0, size=m)
    xp_in["b_prime"][:6] = [-800.0, -40.0, 0.0, 37.5, 40.0, 800.0]
    x_out = empty_structured_array(m, names=NAMES)
    log_j_inv = np.zeros(m)
    x_out, _, log_j_inv = combined.inverse_reparameterise(x_out, xp_in.copy(), log_j_inv)

    out = {"n": n, "m": m}
    for p in NAMES:
        out[f"live/{p}"] = np.array(live[p], copy=True)
        out[f"x_out/{p}"] = np.array(x_out[p], copy=True)
    for p in NAMES:  # the null reparameterisation keeps the parameter's own name in the prime space
        pp = p + "_prime" if p + "_prime" in combined.prime_parameters else p
        out[f"x_prime/{p}_prime"] = np.array(x_prime[pp], copy=True)
        out[f"xp_in/{p}_prime"] = np.array(xp_in[pp], copy=True)
    out["log_j"] = log_j
    out["log_j_inv"] = log_j_inv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reparam_goldens.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: np.shape(v) for k, v in out.items() if k.startswith("log")})


if __name__ == "__main__":
    main()
