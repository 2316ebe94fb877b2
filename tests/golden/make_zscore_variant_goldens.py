#!/usr/bin/env python
"""Golden vectors for the z-score reparameterisations with a rescaling in front of the affine map or behind it, from the
REFERENCE's own ``FlowProposal`` (registry names ``z-score-logit``, ``log-z-score`` / ``log-standardise``,
``z-score-inv-gaussian-cdf``, ``z-score-gaussian-cdf`` and their ``zscore-`` spellings:
``src/nessai/reparameterisations/__init__.py:99-162``; ``ScaleAndShift``: ``rescale.py:119-312``; the rescaling
functions: ``utils/rescaling.py:290-415``).

Run in the build container only (needs /root/reference):

    python tests/golden/make_zscore_variant_goldens.py

A 5-parameter box model, one parameter per variant and a plain z-score.  The same quantities as
``make_angle_goldens.py`` stores, made by the same code.
"""
import make_angle_goldens as base

NAMES = ["a", "b", "c", "d", "e"]
BOUNDS = {"a": [0.0, 1.0], "b": [0.1, 9.0], "c": [0.0, 1.0], "d": [-5.0, 5.0], "e": [0.5, 2.5]}
REPARAMETERISATIONS = {"a": "z-score-logit", "b": "log-z-score", "c": "zscore-inv-gaussian-cdf",
                       "d": "z-score-gaussian-cdf", "e": "z-score"}

if __name__ == "__main__":
    base.main(NAMES, BOUNDS, REPARAMETERISATIONS, "zscore_variant_goldens.npz", latent_scale=0.6)
