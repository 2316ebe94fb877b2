"""Truncation rules (``nessai_amd.truncation``) against vectors captured from the reference's own
classes by ``tests/golden/make_truncation_goldens.py`` (``FlowProposal.configure_truncation`` and
``proposal/flowproposal/truncation.py``): argument resolution incl. the deprecated latent-radius
spellings, ``fuzz = (1 + expansion_fraction) ** (1 / dims)`` outside constant-volume mode, the
adaptive radius with its clipping, and the array-level filters."""
import json
import os
import pickle

import numpy as np
import pytest

from nessai_amd import truncation as T

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "truncation_goldens.json")))
LEGACY = T.LEGACY_LATENT_RADIUS_ARGUMENTS


class Stand:
    def __init__(self, dims, latents=None, log_q=None):
        self.prime_dims = dims
        self.training_data = np.zeros(5)
        self._latents, self._log_q = latents, log_q

    def forward_pass(self, points, rescale=True, compute_radius=False):
        return self._latents, self._log_q


def scheme_for(case, dims):
    names, kwargs = T.resolve_truncation(
        truncation_method=case.get("truncation_method"), truncation_methods=case.get("truncation_methods"),
        truncation_kwargs=case.get("truncation_kwargs"), truncate_log_q=case.get("truncate_log_q", False),
        enforce_likelihood_threshold=case.get("enforce_likelihood_threshold", False),
        legacy={k: case.get(k) for k in LEGACY},
    )
    scheme = T.build_scheme(names, kwargs)
    scheme.configure(Stand(dims))
    return scheme


@pytest.mark.parametrize("entry", GOLD["resolution"], ids=lambda e: f"{sorted(e['case'].items())}-D{e['dims']}"[:80])
def test_argument_resolution_matches_the_reference(entry):
    scheme = scheme_for(entry["case"], entry["dims"])
    expected = entry["result"]
    assert scheme.rule_names == expected["names"]
    rule = scheme.get_rule("latent_radius")
    if "latent_radius" in expected:
        got = rule.to_kwargs()
        for key, value in expected["latent_radius"].items():
            mine = got[key]
            if key.endswith("radius") and mine is False:
                mine = None
            if isinstance(value, float):
                assert mine == pytest.approx(value, rel=0, abs=0), key  # bit-exact
            else:
                assert mine == value, key
    else:
        assert rule is None


@pytest.mark.parametrize("entry", GOLD["prepare"], ids=lambda e: str(sorted(e["kwargs"].items())))
def test_population_radius_and_threshold(entry):
    rule = T.LatentRadiusTruncation(**entry["kwargs"])
    stand = Stand(4, latents=np.array(entry["latents"]))
    rule.configure(stand)
    assert rule.fuzz == entry["fuzz"]
    rule.prepare(stand, None, radius=None)
    assert rule.radius == entry["radius"] and rule.threshold == entry["threshold"]
    rule.prepare(stand, None, radius=1.25)
    assert rule.threshold == entry["threshold_with_r"]
    rule.reset()
    assert np.isnan(rule.radius) and np.isnan(rule.threshold)


def test_array_level_filters():
    g = GOLD["apply"]["latent"]
    rule = T.LatentRadiusTruncation(radius_mode="fixed", fixed_radius=g["fixed_radius"],
                                    expansion_fraction=g["expansion_fraction"])
    stand = Stand(4)
    rule.configure(stand)
    rule.prepare(stand, None)
    assert rule.threshold == g["threshold"]
    assert np.array_equal(rule.apply_latent(stand, np.array(g["z"])), np.array(g["kept"]))

    g = GOLD["apply"]["min_log_q"]
    rule = T.MinLogQTruncation()
    rule.prepare(Stand(2, log_q=np.array(g["train_log_q"])), None)
    assert rule.min_log_q == g["min_log_q"]
    log_q = np.array(g["log_q"])
    kept = rule.apply_after_backward(None, np.arange(len(log_q)), log_q, None)
    assert np.array_equal(kept[1], np.array(g["kept_log_q"])) and kept[2] is None

    g = GOLD["apply"]["likelihood_threshold"]
    rule = T.LikelihoodThresholdTruncation()
    worst = np.zeros(1, dtype=[("a", "f8"), ("logL", "f8")])
    worst["logL"] = g["threshold"]
    rule.prepare(None, worst[0])
    assert rule.threshold == g["threshold"] and rule.requires_log_likelihood
    x = np.zeros(len(g["logL"]), dtype=[("a", "f8"), ("logL", "f8")])
    x["logL"] = g["logL"]
    kept = rule.apply_after_likelihood(None, x, np.zeros(len(x)), None)
    assert np.array_equal(kept[0]["logL"], np.array(g["kept_logL"]))


def test_scheme_limits_and_errors():
    scheme = scheme_for(dict(radius_mode="fixed", fixed_radius=2.0, truncate_log_q=True,
                             enforce_likelihood_threshold=True), 3)
    assert scheme.rule_names == ["latent_radius", "min_log_q", "likelihood_threshold"]
    assert scheme.requires_log_likelihood
    worst = np.zeros(1, dtype=[("a", "f8"), ("logL", "f8")])
    worst["logL"] = -3.5
    scheme.prepare(Stand(3, log_q=np.array([1.0, -2.0])), worst[0])
    r, q, l = scheme.device_limits()
    assert r == pytest.approx(2.0 * 5.0 ** (1 / 3)) and q == -2.0 and l == -3.5
    state = pickle.loads(pickle.dumps(scheme))  # transient numbers are not pickled
    assert np.isnan(state.get_rule("latent_radius").radius)
    none = T.build_scheme(["min_log_q"], {})
    assert none.device_limits()[0] == np.inf
    with pytest.raises(ValueError):
        T.build_scheme(["latent_radius", "latent_radius"], {})
    with pytest.raises(ValueError):
        T.resolve_truncation(truncation_method="a", truncation_methods=["b"])
    with pytest.raises(ValueError):
        T.build_scheme(["nope"], {})
    with pytest.raises(ValueError):
        T.LatentRadiusTruncation(radius_mode="fixed")
    with pytest.raises(RuntimeError):
        T.LatentRadiusTruncation(fixed_radius="2")
    # the reference's `value in (False, None)` coercion (truncation.py:244-249): 0 switches the rule off
    # (-> adaptive mode), True counts as 1.0
    assert T.LatentRadiusTruncation(fixed_radius=0).radius_mode == "adaptive"
    assert T.LatentRadiusTruncation(fixed_radius=0.0).fixed_radius is False
    one = T.LatentRadiusTruncation(fixed_radius=True)
    assert one.fixed_radius == 1.0 and one.radius_mode == "fixed"
