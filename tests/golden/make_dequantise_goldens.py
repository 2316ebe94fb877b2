#!/usr/bin/env python
"""Golden vectors for the ``dequantise`` / ``dequantise-logit`` reparameterisations, from the REFERENCE's own
``FlowProposal`` (``src/nessai/reparameterisations/discrete.py:9-78``: uniform noise on a discrete parameter, then
``RescaleToBounds`` on [min, max + 1] - with a logit for the second registry name - and ``floor`` on the way back;
registry: ``reparameterisations/__init__.py:180-189``).

Run in the build container only (needs /root/reference):

    python tests/golden/make_dequantise_goldens.py

A 4-parameter box model: two continuous parameters and two discrete ones (0..7 with ``dequantise``, 2..5 with
``dequantise-logit``).  The same quantities as ``make_angle_goldens.py`` stores, made by the same code (the noise comes
from the proposal's generator: its state is pinned before ``rescale``); the live points' discrete columns are rounded
down before they go in.
"""
import numpy as np

import make_angle_goldens as base

NAMES = ["a", "k", "c", "m"]
BOUNDS = {"a": [-3.0, 7.0], "k": [0.0, 7.0], "c": [0.5, 2.5], "m": [2.0, 5.0]}
REPARAMETERISATIONS = {"a": "rescaletobounds", "k": "dequantise", "c": "z-score", "m": "dequantise-logit"}
DISCRETE = ("k", "m")

if __name__ == "__main__":
    base.main(NAMES, BOUNDS, REPARAMETERISATIONS, "dequantise_goldens.npz", discrete=DISCRETE)
