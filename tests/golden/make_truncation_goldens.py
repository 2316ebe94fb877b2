#!/usr/bin/env python
"""Golden vectors for the truncation rules, captured from the REFERENCE's own classes
(``/root/reference/src/nessai/proposal/flowproposal/truncation.py`` and
``FlowProposal.configure_truncation``, flowproposal.py:276-331).  Build container only.

    python -B tests/golden/make_truncation_goldens.py   ->   tests/golden/truncation_goldens.json

Cases: (1) argument resolution - for each combination of new / deprecated arguments, the ordered
rule names and the attributes of the latent-radius rule after ``configure`` on a D-dimensional
proposal; (2) ``LatentRadiusTruncation.prepare`` (radius, threshold) with a stand-in proposal whose
``forward_pass`` returns fixed latents; (3) the array-level ``apply_*`` of the three rules.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import reference_harness  # noqa: E402

reference_harness.install()
sys.path.insert(0, reference_harness.REFERENCE_SRC)

from nessai.livepoint import numpy_array_to_live_points  # noqa: E402
from nessai.proposal.flowproposal.flowproposal import FlowProposal  # noqa: E402
from nessai.proposal.flowproposal.truncation import (  # noqa: E402
    TruncationScheme,
    LatentRadiusTruncation,
    LikelihoodThresholdTruncation,
    MinLogQTruncation,
    get_deprecated_latent_radius_kwargs,
)

LEGACY = ("constant_volume_mode", "volume_fraction", "fuzz", "expansion_fraction", "fixed_radius",
          "radius_mode", "min_radius", "max_radius", "compute_radius_with_all")

CASES = [
    dict(),
    dict(radius_mode="adaptive"),
    dict(radius_mode="fixed", fixed_radius=3.0),
    dict(fixed_radius=2.0),
    dict(radius_mode="adaptive", expansion_fraction=1.5, min_radius=1.0, max_radius=4.0),
    dict(radius_mode="adaptive", expansion_fraction=0.0, fuzz=1.3),
    dict(constant_volume_mode=True, volume_fraction=0.8),
    dict(volume_fraction=0.5),
    dict(truncate_log_q=True),
    dict(enforce_likelihood_threshold=True),
    dict(truncation_method="min_log_q"),
    dict(truncation_methods=["likelihood_threshold", "min_log_q"]),
    dict(truncation_methods=["min_log_q", "latent_radius"], truncation_kwargs={"latent_radius": {"radius_mode": "adaptive"}}),
    dict(truncation_method="latent_radius", truncation_kwargs={"radius_mode": "fixed", "fixed_radius": 2.2}),
    dict(truncation_method="latent_radius", radius_mode="adaptive", truncation_kwargs={"latent_radius": {"max_radius": 7.0}}),
    dict(radius_mode="constant_volume", min_radius=9.0, max_radius=1.0, volume_fraction=0.95),
]


class Stand:
    """What the rules ask of a proposal."""

    def __init__(self, dims, latents=None, log_q=None):
        self.prime_dims = dims
        self.training_data = np.zeros(5)
        self._latents, self._log_q = latents, log_q

    def forward_pass(self, points, rescale=True, compute_radius=False):
        return self._latents, self._log_q


def resolve(case, dims):
    prop = FlowProposal.__new__(FlowProposal)
    prop._truncation_scheme = TruncationScheme([])
    legacy = {k: case.get(k) for k in LEGACY}
    prop.configure_truncation(
        truncation_method=case.get("truncation_method"), truncation_methods=case.get("truncation_methods"),
        truncation_kwargs=case.get("truncation_kwargs"), truncate_log_q=case.get("truncate_log_q", False),
        enforce_likelihood_threshold=case.get("enforce_likelihood_threshold", False),
        latent_radius_kwargs=get_deprecated_latent_radius_kwargs(**legacy) or None, default_latent_radius=True,
    )
    scheme = prop.truncation
    scheme.configure(Stand(dims))
    out = dict(names=list(scheme.rule_names))
    rule = scheme.get_rule("latent_radius")
    if rule is not None:
        out["latent_radius"] = {k: (None if v is False else v) if k.endswith("radius") else v
                                for k, v in rule.to_kwargs().items()}
    return out


def main():
    gold = dict(resolution=[], prepare=[], apply={})
    for case in CASES:
        for dims in (2, 16):
            gold["resolution"].append(dict(case=case, dims=dims, result=resolve(case, dims)))
    rng = np.random.default_rng(8)
    latents = rng.normal(size=(7, 4))
    for kw in (dict(radius_mode="adaptive"), dict(radius_mode="adaptive", max_radius=1.5),
               dict(radius_mode="adaptive", min_radius=6.0), dict(radius_mode="adaptive", compute_radius_with_all=True),
               dict(radius_mode="fixed", fixed_radius=2.5), dict(radius_mode="constant_volume", volume_fraction=0.9)):
        rule = LatentRadiusTruncation(**kw)
        stand = Stand(4, latents=latents if kw.get("compute_radius_with_all") else latents[:1])
        rule.configure(stand)
        rule.prepare(stand, worst_point=None, radius=None)
        gold["prepare"].append(dict(kwargs=kw, latents=stand._latents.tolist(), radius=rule.radius,
                                    threshold=rule.threshold, fuzz=rule.fuzz))
        rule.prepare(stand, worst_point=None, radius=1.25)
        gold["prepare"][-1]["threshold_with_r"] = rule.threshold
    z = rng.normal(size=(40, 4))
    rule = LatentRadiusTruncation(radius_mode="fixed", fixed_radius=1.8, expansion_fraction=0.5)
    stand = Stand(4)
    rule.configure(stand)
    rule.prepare(stand, None)
    gold["apply"]["latent"] = dict(z=z.tolist(), fixed_radius=1.8, expansion_fraction=0.5, threshold=rule.threshold,
                                   kept=rule.apply_latent(stand, z).tolist())
    log_q = rng.normal(size=40)
    x = numpy_array_to_live_points(rng.normal(size=(40, 2)), ["a", "b"])
    x["logL"] = rng.normal(size=40)
    rule = MinLogQTruncation()
    rule.prepare(Stand(2, latents=None, log_q=np.array([0.3, -0.2, 0.9])), None)
    kept = rule.apply_after_backward(None, x, log_q, z)
    gold["apply"]["min_log_q"] = dict(train_log_q=[0.3, -0.2, 0.9], min_log_q=rule.min_log_q, log_q=log_q.tolist(),
                                      kept_log_q=kept[1].tolist())
    worst = numpy_array_to_live_points(np.zeros((1, 2)), ["a", "b"])
    worst["logL"] = 0.1
    rule = LikelihoodThresholdTruncation()
    rule.prepare(None, worst[0])
    kept = rule.apply_after_likelihood(None, x, log_q, z)
    gold["apply"]["likelihood_threshold"] = dict(threshold=rule.threshold, logL=x["logL"].tolist(),
                                                 kept_logL=kept[0]["logL"].tolist())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "truncation_goldens.json")
    with open(path, "w") as f:
        json.dump(gold, f, indent=1, sort_keys=True)
    print("wrote", path, len(gold["resolution"]), "resolution cases")


if __name__ == "__main__":
    main()
