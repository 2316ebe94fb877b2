#!/usr/bin/env python
"""Golden vectors for the ``to-cartesian`` reparameterisation, from the REFERENCE's own ``FlowProposal``
(``src/nessai/reparameterisations/angle.py:189-232``: a bounded parameter mapped to [0, 1], mirrored - ``mode`` "split":
half of the rows, chosen by the proposal's generator; "duplicate": every row twice; "half": not at all - times pi,
then a point of the plane with an auxiliary chi(2) radius like an angle; registry name ``to-cartesian``).

Run in the build container only (needs /root/reference):

    python tests/golden/make_to_cartesian_goldens.py

A 4-parameter box model with one parameter per mode and a z-score parameter between them.  The same quantities as
``make_angle_goldens.py`` stores, made by the same code (``rescale`` returns twice the rows: the "duplicate" parameter).
"""
import make_angle_goldens as base

NAMES = ["s", "c", "d", "h"]
BOUNDS = {"s": [-3.0, 7.0], "c": [0.5, 2.5], "d": [2.0, 5.0], "h": [0.0, 1.5]}
REPARAMETERISATIONS = {
    "s": "to-cartesian",
    "c": "z-score",
    "d": {"reparameterisation": "to-cartesian", "mode": "duplicate"},
    "h": {"reparameterisation": "to-cartesian", "mode": "half"},
}

if __name__ == "__main__":
    base.main(NAMES, BOUNDS, REPARAMETERISATIONS, "to_cartesian_goldens.npz")
