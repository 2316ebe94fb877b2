"""Truncation rules of ``FlowProposal.populate`` as device predicates.

Counterpart of ``src/nessai/proposal/flowproposal/truncation.py``.  The reference filters host
arrays three times per draw batch (``flowproposal.py:431-466``):

====================  =============================================  =====================================
rule                  reference                                      here
====================  =============================================  =====================================
``latent_radius``     ``apply_latent``: ``|z| <= fuzz * r``          compare inside the fused draw kernel
                      (``truncation.py:358-365``)                    (``radius`` argument, float64)
``min_log_q``         ``apply_after_backward``: ``log_q > min``      ``nsa_populate_truncate`` marks the row
                      (``truncation.py:385-392``)                    dropped (NaN ``log_q``)
``likelihood_...``    ``apply_after_likelihood``: ``logL > L*``      same kernel, on a device (or uploaded)
                      (``truncation.py:423-430``)                    ``logL`` column
====================  =============================================  =====================================

A rule therefore only has to produce *numbers* per ``populate`` call (``prepare``): the radius
threshold, the smallest ``log_q`` of the training data, the likelihood of the worst point.
``TruncationScheme.device_limits()`` hands them to the proposal.  The numpy ``apply_*`` methods
keep the reference's array-level API for callers that filter host arrays themselves; ``populate``
does not use them.

Constructor arguments, defaults, error messages and the legacy-argument resolution follow the
reference (``truncation.py:15-151``, ``flowproposal.py:246-331``) so that the same keyword
arguments configure the same rules.
"""

import copy
import logging
import math

import numpy as np

from .utils import compute_radius

logger = logging.getLogger(__name__)

#: ``FlowProposal`` arguments that configure the latent-radius rule (deprecated spelling)
LEGACY_LATENT_RADIUS_ARGUMENTS = (
    "constant_volume_mode", "volume_fraction", "fuzz", "expansion_fraction", "fixed_radius",
    "radius_mode", "min_radius", "max_radius", "compute_radius_with_all",
)
_LATENT_RADIUS_DEFAULTS = dict(radius_mode="constant_volume", volume_fraction=0.95)


class BaseTruncationRule:
    """A rule owns per-population ("transient") numbers that ``reset`` forgets."""

    name = "base"
    _transient = {}

    def __init__(self):
        self.reset()

    requires_log_likelihood = False

    def configure(self, proposal):
        """Once per proposal, after the rescaling is known (``set_rescaling``)."""

    def prepare(self, proposal, worst_point, radius=None):
        """Once per ``populate`` call."""

    def apply_latent(self, proposal, z):
        return z

    def apply_after_backward(self, proposal, x, log_q, z):
        return x, log_q, z

    def apply_after_likelihood(self, proposal, x, log_q, z):
        return x, log_q, z

    def reset(self):
        for key, value in self._transient.items():
            setattr(self, key, value)

    def __getstate__(self):
        state = dict(self.__dict__)
        state.update(self._transient)
        return state


def _subset(keep, *arrays):
    return tuple(None if a is None else a[keep] for a in arrays)


class LatentRadiusTruncation(BaseTruncationRule):
    """Keep latent draws inside a ball: ``|z|_2 <= fuzz * radius``.

    ``radius_mode``: ``"constant_volume"`` (radius holding ``volume_fraction`` of the latent
    Gaussian's mass, fuzz 1), ``"fixed"`` (``fixed_radius``) or ``"adaptive"`` (norm of the worst
    point - or of all the training data with ``compute_radius_with_all`` - mapped to the latent
    space, clipped to ``[min_radius, max_radius]``).  Outside constant-volume mode the ball is
    inflated by ``fuzz = (1 + expansion_fraction) ** (1 / dims)`` (``truncation.py:292-300``).
    """

    name = "latent_radius"
    _transient = {"_radius": math.nan, "_threshold": math.nan}

    def __init__(self, radius_mode=None, fixed_radius=False, min_radius=False, max_radius=50.0,
                 compute_radius_with_all=False, constant_volume_mode=False, volume_fraction=0.95, fuzz=1.0,
                 expansion_fraction=4.0):
        super().__init__()
        self.fixed_radius = self._radius_argument(fixed_radius, "fixed_radius")
        self.min_radius = self._radius_argument(min_radius, "min_radius")
        self.max_radius = self._radius_argument(max_radius, "max_radius")
        self.compute_radius_with_all = compute_radius_with_all
        self.volume_fraction = float(volume_fraction)
        self.fuzz = float(fuzz)
        self.expansion_fraction = expansion_fraction
        if radius_mode is None:
            radius_mode = ("constant_volume" if constant_volume_mode
                           else "fixed" if self.fixed_radius is not False else "adaptive")
        if radius_mode not in ("adaptive", "fixed", "constant_volume"):
            raise ValueError(f"Unknown radius mode: {radius_mode}. Choose from: adaptive, fixed, constant_volume")
        if radius_mode == "fixed" and self.fixed_radius is False:
            raise ValueError("Fixed radius mode requires `fixed_radius` to be specified")
        self.radius_mode = radius_mode

    @staticmethod
    def _radius_argument(value, name):
        # the reference's coercion to the letter (truncation.py:244-249): `value in (False, None)` is also
        # true for 0 and 0.0 (-> the rule is off), and True passes the isinstance test (-> 1.0)
        if value in (False, None):
            return False
        if not isinstance(value, (int, float)):
            raise RuntimeError(f"{name} must be an int or float")
        return float(value)

    radius = property(lambda self: self._radius)
    threshold = property(lambda self: self._threshold)
    constant_volume_mode = property(lambda self: self.radius_mode == "constant_volume")

    def to_kwargs(self):
        names = ("radius_mode", "fixed_radius", "min_radius", "max_radius", "compute_radius_with_all",
                 "constant_volume_mode", "volume_fraction", "fuzz", "expansion_fraction")
        return {n: getattr(self, n) for n in names}

    def configure(self, proposal):
        dims = proposal.prime_dims
        if self.expansion_fraction:
            self.fuzz = (1 + self.expansion_fraction) ** (1 / dims)
        if self.radius_mode != "constant_volume":
            return
        self.fixed_radius = compute_radius(dims, self.volume_fraction)
        self.fuzz = 1.0
        if self.max_radius and self.max_radius < self.fixed_radius:
            logger.warning("Max radius is less than the constant-volume radius. Disabling max radius.")
            self.max_radius = False
        if self.min_radius and self.min_radius > self.fixed_radius:
            logger.warning("Min radius is greater than the constant-volume radius. Disabling min radius.")
            self.min_radius = False

    def _population_radius(self, proposal, worst_point, radius):
        if radius is not None:
            return float(radius)
        if self.radius_mode != "adaptive":
            if self.fixed_radius is False:
                raise RuntimeError("Fixed radius mode requires `fixed_radius` to be specified")
            return float(self.fixed_radius)
        points = worst_point
        if self.compute_radius_with_all:
            points = proposal.training_data
            if points is None or not len(points):
                raise RuntimeError("compute_radius_with_all requires training_data to be set")
        latents = proposal.forward_pass(points, rescale=True, compute_radius=True)[0]
        r = float(np.sqrt(np.sum(latents**2.0, axis=-1)).max())
        if self.max_radius and r > self.max_radius:
            r = self.max_radius
        if self.min_radius and r < self.min_radius:
            r = self.min_radius
        return float(r)

    def prepare(self, proposal, worst_point, radius=None):
        self._radius = self._population_radius(proposal, worst_point, radius)
        self._threshold = self.fuzz * self._radius

    def apply_latent(self, proposal, z):
        return z[np.sqrt(np.sum(z**2.0, axis=-1)) <= self.threshold]


class MinLogQTruncation(BaseTruncationRule):
    """Drop candidates whose proposal density is not above the smallest ``log_q`` of the
    training data (``truncation.py:368-392``)."""

    name = "min_log_q"
    _transient = {"_min_log_q": math.nan}

    min_log_q = property(lambda self: self._min_log_q)

    def prepare(self, proposal, worst_point, radius=None):
        data = proposal.training_data
        if data is None or not len(data):
            raise RuntimeError("min_log_q truncation requires training_data to be set")
        self._min_log_q = float(proposal.forward_pass(data)[1].min())

    def apply_after_backward(self, proposal, x, log_q, z):
        return _subset(log_q > self.min_log_q, x, log_q, z)


class LikelihoodThresholdTruncation(BaseTruncationRule):
    """Drop candidates at or below the likelihood of the worst live point before the rejection
    step (``truncation.py:398-430``); needs ``logL`` of every candidate."""

    name = "likelihood_threshold"
    _transient = {"_threshold": math.nan}
    requires_log_likelihood = True

    threshold = property(lambda self: self._threshold)

    def prepare(self, proposal, worst_point, radius=None):
        value = float(np.asarray(worst_point["logL"], dtype=float).reshape(-1)[0])
        self._threshold = value if math.isfinite(value) else -math.inf

    def apply_after_likelihood(self, proposal, x, log_q, z):
        return _subset(x["logL"] > self.threshold, x, log_q, z)


TRUNCATION_REGISTRY = {
    rule.name: rule for rule in (LatentRadiusTruncation, MinLogQTruncation, LikelihoodThresholdTruncation)
}


def get_truncation_rule_class(name):
    if name not in TRUNCATION_REGISTRY:
        raise ValueError(f"Unknown truncation method: {name}")
    return TRUNCATION_REGISTRY[name]


class TruncationScheme:
    """Ordered, duplicate-free list of rules (``truncation.py:447-508``)."""

    def __init__(self, rules=None):
        self.rules = []
        for rule in rules or ():
            self.add_rule(rule)

    rule_names = property(lambda self: [r.name for r in self.rules])
    requires_log_likelihood = property(lambda self: any(r.requires_log_likelihood for r in self.rules))

    def get_rule(self, name):
        return next((r for r in self.rules if r.name == name), None)

    def has_rule(self, name):
        return self.get_rule(name) is not None

    def add_rule(self, rule, index=None):
        if self.has_rule(rule.name):
            raise ValueError(f"Duplicate truncation rule: {rule.name}")
        self.rules.insert(len(self.rules) if index is None else index, rule)

    def configure(self, proposal):
        for rule in self.rules:
            rule.configure(proposal)

    def prepare(self, proposal, worst_point, radius=None):
        self.reset()
        for rule in self.rules:
            rule.prepare(proposal, worst_point, radius=radius)

    def reset(self):
        for rule in self.rules:
            rule.reset()

    def apply_latent(self, proposal, z):
        for rule in self.rules:
            z = rule.apply_latent(proposal, z)
        return z

    def apply_after_backward(self, proposal, x, log_q, z):
        for rule in self.rules:
            x, log_q, z = rule.apply_after_backward(proposal, x, log_q, z)
        return x, log_q, z

    def apply_after_likelihood(self, proposal, x, log_q, z):
        for rule in self.rules:
            x, log_q, z = rule.apply_after_likelihood(proposal, x, log_q, z)
        return x, log_q, z

    def device_limits(self):
        """``(radius threshold, min log_q, logL threshold)`` of the prepared rules for the device
        predicates: ``inf`` / ``-inf`` / ``None`` where a rule is absent."""
        latent = self.get_rule("latent_radius")
        log_q = self.get_rule("min_log_q")
        log_l = self.get_rule("likelihood_threshold")
        return (
            math.inf if latent is None else float(latent.threshold),
            -math.inf if log_q is None else float(log_q.min_log_q),
            None if log_l is None else float(log_l.threshold),
        )


def resolve_truncation(truncation_method=None, truncation_methods=None, truncation_kwargs=None,
                       truncate_log_q=False, enforce_likelihood_threshold=False, legacy=None):
    """Rule names and per-rule keyword arguments from the new and the deprecated arguments.

    The outcome equals the reference's ``configure_truncation(..., default_latent_radius=True)``
    (``flowproposal.py:246-331``): the latent-radius rule leads when no method is named or a
    legacy latent-radius argument is set; ``truncate_log_q`` / ``enforce_likelihood_threshold``
    append their rules; bare ``truncation_kwargs`` belong to a single named ``truncation_method``;
    legacy arguments are overridden by explicit ``truncation_kwargs["latent_radius"]``; the
    latent-radius rule defaults to constant-volume mode with a volume fraction of 0.95.
    """
    if truncation_method is not None and truncation_methods is not None:
        raise ValueError("Specify only one of truncation_method or truncation_methods")
    named = truncation_methods if truncation_methods is not None else truncation_method
    if named is None:
        names = []
    elif isinstance(named, str):
        names = [named]
    else:
        names = list(dict.fromkeys(named))
    legacy = {k: v for k, v in (legacy or {}).items() if v is not None}
    nothing_named = truncation_method is None and truncation_methods is None
    if (nothing_named or legacy) and "latent_radius" not in names:
        names.insert(0, "latent_radius")
    if truncate_log_q and "min_log_q" not in names:
        names.append("min_log_q")
    if enforce_likelihood_threshold and "likelihood_threshold" not in names:
        names.append("likelihood_threshold")

    kwargs = copy.deepcopy(truncation_kwargs) if truncation_kwargs is not None else {}
    bare = (isinstance(truncation_method, str) and truncation_methods is None and kwargs
            and truncation_method not in kwargs and not any(isinstance(v, dict) for v in kwargs.values()))
    if bare:
        kwargs = {truncation_method: kwargs}
    if legacy and "latent_radius" in names:
        kwargs["latent_radius"] = {**legacy, **kwargs.get("latent_radius", {})}
    if "latent_radius" in names:
        latent = kwargs.setdefault("latent_radius", {})
        if isinstance(latent, dict):
            for key, value in _LATENT_RADIUS_DEFAULTS.items():
                latent.setdefault(key, value)
    return names, kwargs


def build_scheme(names, kwargs):
    rules = []
    for name in names:
        options = kwargs.get(name, {})
        if not isinstance(options, dict):
            raise TypeError(f"Truncation kwargs for {name} must be a dictionary")
        rules.append(get_truncation_rule_class(name)(**options))
    return TruncationScheme(rules)
