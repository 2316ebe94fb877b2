#!/usr/bin/env python
"""Golden vectors for the ``angle-pair`` reparameterisation, from the REFERENCE's own ``FlowProposal``
(``src/nessai/reparameterisations/angle.py:186-538``: two angles - a direction - become a point of space, its radius
an auxiliary parameter drawn from chi(3); registry name ``angle-pair``: ``reparameterisations/__init__.py``).

Run in the build container only (needs /root/reference):

    python tests/golden/make_angle_pair_goldens.py

A 6-parameter box model with two pairs and two ordinary parameters: (dec, ra) - given vertical angle first and not next
to each other in the model, right ascension on [0, 2 pi] (modulo 2 pi on the way back), declination on
[-pi/2, pi/2] (the ra-dec convention) - and (az, zen) with the azimuth on [-pi, pi] and the zenith on [0, pi] (az-zen).
The same quantities as ``make_angle_goldens.py`` stores, made by the same code.
"""
import numpy as np

import make_angle_goldens as base

NAMES = ["a", "dec", "c", "ra", "az", "zen"]
BOUNDS = {"a": [-3.0, 7.0], "dec": [-np.pi / 2.0, np.pi / 2.0], "c": [0.5, 2.5], "ra": [0.0, 2.0 * np.pi],
          "az": [-np.pi, np.pi], "zen": [0.0, np.pi]}
# keyed by parameters of the model, in model order of each group's first parameter
REPARAMETERISATIONS = {
    "a": "rescaletobounds",
    "dec": {"reparameterisation": "angle-pair", "parameters": ["dec", "ra"]},
    "c": "z-score",
    "az": {"reparameterisation": "angle-pair", "parameters": ["az", "zen"]},
}

if __name__ == "__main__":
    base.main(NAMES, BOUNDS, REPARAMETERISATIONS, "angle_pair_goldens.npz")
