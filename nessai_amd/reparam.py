"""Per-parameter reparameterisations of the proposal, host side and device tables.

Counterpart of the part of the reference's reparameterisation registry
(``src/nessai/reparameterisations/__init__.py:28-110``) that is a per-parameter, one-to-one
map and therefore fits the epilogue of the fused draw kernel:

====================================  =========================================================
registry name                         map (x -> x')
====================================  =========================================================
``z-score`` ``zscore`` ``standardize``  ``ScaleAndShift`` with estimated mean / std
                                      (``rescale.py:119-312``)
``scaleandshift`` ``scale`` ``rescale``  ``ScaleAndShift`` / ``Rescale`` with fixed constants
``z-score-logit`` ``log-z-score``       the z-score of ``logit(x)`` / ``log(x)`` / the inverse Gaussian CDF of x (a
``log-standardise``                   rescaling in front of the affine map, statistics of the rescaled values), and
``z-score-inv-gaussian-cdf``          ``z-score-gaussian-cdf``: the Gaussian CDF of the z-score (behind it)
``z-score-gaussian-cdf`` (+ ``zscore-``)  (``reparameterisations/__init__.py:99-162``, ``utils/rescaling.py:290-415``)
``default`` ``rescaletobounds``        ``RescaleToBounds`` to [-1, 1], bounds updated from the live
``rescale-to-bounds`` ``angle-sine``   points (``rescale.py:321-731``)
``angle-cosine``
``offset``                            the same with the prior mid-point subtracted first
``logit``                             ``RescaleToBounds`` to [0, 1] with the prior bounds, then
                                      ``logit`` (``utils/rescaling.py:290-337``)
``log-rescale``                       the same followed by ``log`` (``utils/rescaling.py:359-395``)
``inversion`` ``inversion-duplicate``   ``RescaleToBounds`` with boundary inversion (``rescale.py:541-612``):
                                      the edge the live points pile up at is detected
                                      (``utils/rescaling.py:106-193``), the parameter is mapped to [0, 1]
                                      measured from that edge and mirrored about it - half of the points
                                      flipped (``split``) or every point doubled (``duplicate``)
``null`` ``none``                     ``NullReparameterisation`` (``null.py``)
``angle`` ``angle-2pi`` ``angle-pi``   ``Angle`` (``angle.py:17-183``): the angle times ``scale`` (1, 1, 2; ``periodic``:
``periodic``                          2 pi / prior range) becomes a point ``r (cos, sin)`` of the plane, ``r`` an
                                      AUXILIARY radius drawn from chi(2) (the proposal's generator through scipy, as the
                                      reference draws it); back: ``atan2`` (modulo 2 pi when the prior range starts at 0)
                                      / ``scale`` and the radius, which stays in x and carries a chi(2) prior
``dequantise`` ``dequantise-logit``   ``Dequantise`` (``discrete.py:9-78``): a discrete parameter plus uniform noise (the
                                      proposal's generator), rescaled on [min, max + 1] (then ``logit``); back:
                                      the inverse map, then ``floor``
``to-cartesian``                      ``ToCartesian`` (``angle.py:189-232``): a bounded parameter on [0, 1], mirrored
                                      (``mode`` split: half of the rows; duplicate: every row twice; half: none),
                                      times pi, then as an angle; back: ``|angle| / pi`` onto the bounds
``angle-pair``                        ``AnglePair`` (``angle.py:186-538``): two angles -> ``r`` times the unit vector they
                                      point along, ``r`` an auxiliary radius from chi(3); conventions ``ra-dec``
                                      (vertical angle on [-pi/2, pi/2]) and ``az-zen`` ([0, pi]) found from the bounds
====================================  =========================================================

An angle changes the number of columns (round 6): x' has TWO columns for it (``<name>_x``, ``<name>_y``), x gains
``<name>_radial``, the flow has D' = D + (number of angles) inputs and the draw kernel writes D' output columns - the
angle where the x coordinate was, the radius where the y coordinate was (``device_names``).

The mirrored FORWARD map changes the number of rows (training data, host side, numpy as in the
reference); its INVERSE - fold ``|x'|`` back, undo the [0, 1] map - is one-to-one per column and
runs in the draw kernel's epilogue like the others.

A PAIR of angles (``angle-pair``, angle.py:186-538: a direction - right ascension / declination or azimuth / zenith)
becomes a point of space: x' has THREE columns (``<h>_<v>_x``, ``_y``, ``_z``) where the pair's first parameter sat,
x gains ``<h>_<v>_radial`` with a chi(3) prior, and the device columns are (horizontal, vertical, radius).

An angle or a pair with a radial parameter of the model and user-defined classes raise ``NotImplementedError`` here.

The configuration forms are the reference's (``reparameterisations/utils.py:120-268``):
``None`` | ``"name"`` | ``{parameter: "name"}`` | ``{parameter: {"reparameterisation": "name",
**kwargs}}`` | ``{"name": {"parameters": [...], **kwargs}}`` | ``{"name": [patterns]}``;
parameters without an entry get the fallback (``z-score``, ``proposal/flowproposal/base.py:115``).

The x'-columns keep the order of ``model.names`` (the reference orders them by group, which is
not observable by a flow trained from scratch).  The log-Jacobian is accumulated group by group
and parameter by parameter in the reference's order (``combined.py:154-192``).
"""

import re

import numpy as np

from .livepoint import empty_structured_array

# kinds of nsa_populate_draw's rp_kind table (nsa_flow_kernel.h NSA_RP_*)
RP_SCALE_SHIFT, RP_NULL, RP_TO_BOUNDS, RP_TO_BOUNDS_LOGIT, RP_TO_BOUNDS_LOG = 0, 1, 2, 3, 4
RP_INVERT, RP_MINUS_ONE_ONE = 5, 6  # boundary inversion: folded [0, 1] map about an edge / plain [-1, 1] map
# an angle as a point of the plane (reparameterisations/angle.py, round 6): the x' column of its x coordinate (-> the
# angle) and, right behind it, the column of its y coordinate (-> the auxiliary radius)
RP_ANGLE, RP_ANGLE_RADIAL = 7, 8
# a pair of angles (a direction: azimuth / zenith or right ascension / declination) as a point of space
# (angle.py:186-538): three x' columns in a row -> the horizontal angle, the vertical angle, the auxiliary radius
RP_PAIR_H, RP_PAIR_V, RP_PAIR_R = 9, 10, 11
# a discrete parameter (discrete.py:9-78): the three rescale-to-bounds kinds followed by floor()
RP_DEQUANTISE, RP_DEQUANTISE_LOGIT, RP_DEQUANTISE_LOG = 12, 13, 14
# scale-and-shift with a rescaling in front of it (x' * scale + shift, then sigmoid / exp / Gaussian CDF on the way back)
# or behind it (inverse Gaussian CDF of x' first): rescale.py:119-312, utils/rescaling.py:290-415
RP_SS_SIGMOID, RP_SS_EXP, RP_SS_GAUSSIAN_CDF, RP_INV_GAUSSIAN_CDF_SS = 15, 16, 17, 18

_KNOWN = {
    "z-score": ("scale_shift", dict(estimate_scale=True, estimate_shift=True)),
    "zscore": ("scale_shift", dict(estimate_scale=True, estimate_shift=True)),
    "standardize": ("scale_shift", dict(estimate_scale=True, estimate_shift=True)),
    # (reparameterisations/__init__.py:99-162: the z-score with a rescaling in front of it or behind it)
    "zscore-gaussian-cdf": ("scale_shift", dict(estimate_scale=True, estimate_shift=True, post_rescaling="gaussian_cdf")),
    "z-score-gaussian-cdf": ("scale_shift", dict(estimate_scale=True, estimate_shift=True, post_rescaling="gaussian_cdf")),
    "z-score-logit": ("scale_shift", dict(estimate_scale=True, estimate_shift=True, pre_rescaling="logit")),
    "zscore-logit": ("scale_shift", dict(estimate_scale=True, estimate_shift=True, pre_rescaling="logit")),
    "z-score-inv-gaussian-cdf": ("scale_shift", dict(estimate_scale=True, estimate_shift=True,
                                                     pre_rescaling="inv_gaussian_cdf")),
    "zscore-inv-gaussian-cdf": ("scale_shift", dict(estimate_scale=True, estimate_shift=True,
                                                    pre_rescaling="inv_gaussian_cdf")),
    "log-z-score": ("scale_shift", dict(estimate_scale=True, estimate_shift=True, pre_rescaling="log")),
    "log-standardise": ("scale_shift", dict(estimate_scale=True, estimate_shift=True, pre_rescaling="log")),
    "scaleandshift": ("scale_shift", {}),
    "scale": ("scale_shift", {}),
    "rescale": ("scale_shift", {}),
    "default": ("to_bounds", {}),
    "rescaletobounds": ("to_bounds", {}),
    "rescale-to-bounds": ("to_bounds", {}),
    # (the reference registers these two names for the same class: reparameterisations/__init__.py:168-173 - the
    # bounded inclination / declination angles of the GW parameter sets)
    "angle-sine": ("to_bounds", {}),
    "angle-cosine": ("to_bounds", {}),
    "offset": ("to_bounds", dict(offset=True)),
    "inversion": ("to_bounds", dict(detect_edges=True, boundary_inversion=True, inversion_type="split")),
    "inversion-duplicate": ("to_bounds", dict(detect_edges=True, boundary_inversion=True, inversion_type="duplicate")),
    "logit": ("to_bounds", dict(rescale_bounds=[0.0, 1.0], update_bounds=False, post_rescaling="logit")),
    "log-rescale": ("to_bounds", dict(rescale_bounds=[0.0, 1.0], update_bounds=False, post_rescaling="log")),
    "null": ("null", {}),
    "none": ("null", {}),
    # Angle with an auxiliary chi(2)-distributed radius (reparameterisations/__init__.py:158-177, angle.py:17-183)
    "angle": ("angle", dict(scale=1.0)),
    "angle-pi": ("angle", dict(scale=2.0)),
    "angle-2pi": ("angle", dict(scale=1.0)),
    "periodic": ("angle", dict(scale=None)),
    # ToCartesian (reparameterisations/__init__.py:179, angle.py:189-232): any bounded parameter to [0, 1], mirrored,
    # times pi: an angle of the half plane (split / half) or of the plane
    "to-cartesian": ("angle", dict(scale=np.pi, cartesian=True)),
    "angle-pair": ("angle_pair", {}),
    # Dequantise (reparameterisations/__init__.py:180-189, discrete.py:9-78): uniform noise on a discrete parameter,
    # then RescaleToBounds on [min, max + 1]; floor() on the way back
    "dequantise": ("to_bounds", dict(dequantise=True, update_bounds=False)),
    "dequantise-logit": ("to_bounds", dict(dequantise=True, rescale_bounds=[0.0, 1.0], update_bounds=False,
                                           post_rescaling="logit")),
}
_NON_SAMPLING = ("logP", "logL", "it")


def _logit(x):
    """(``utils/rescaling.py:290-315``, no clamping: the registry passes no eps)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        log_j = -np.log(x) - np.log1p(-x)
        return np.log(x) - np.log1p(-x), log_j


def _gaussian_cdf(x):
    """(``utils/rescaling.py:398-402``)."""
    from scipy.special import erfc

    log_j = -0.5 * np.log(2 * np.pi) - 0.5 * (x ** 2.0)
    return 0.5 * erfc(-x / np.sqrt(2.0)), log_j


def _inverse_gaussian_cdf(x):
    """(``utils/rescaling.py:405-409``)."""
    from scipy.special import erfcinv

    x = -np.sqrt(2) * erfcinv(2.0 * x)
    return x, 0.5 * np.log(2 * np.pi) + 0.5 * (x ** 2.0)


def _log(x):
    """(``utils/rescaling.py:359-375``)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        y = np.log(x)
    return y, -y


def _exp(x):
    """(``utils/rescaling.py:378-395``)."""
    with np.errstate(over="ignore"):
        return np.exp(x), np.array(x, dtype=np.float64, copy=True)


def _sigmoid(x):
    """(``utils/rescaling.py:318-337``)."""
    with np.errstate(over="ignore", divide="ignore"):
        x = np.divide(1, 1 + np.exp(-x))
        log_j = np.log(x) + np.log1p(-x)
    return x, log_j


def _auto_bins(x, max_bins=50):
    """Histogram bin count: the narrower of the Freedman-Diaconis and Sturges widths, at most
    ``max_bins`` (``utils/hist.py:9-76``)."""
    x = np.asarray(x)
    if not x.size:
        raise RuntimeError("Input array is empty!")
    q75, q25 = np.percentile(x, [75, 25])
    fd = 2.0 * (q75 - q25) * x.size ** (-1.0 / 3.0)
    sturges = np.ptp(x) / (np.log2(x.size) + 1.0)
    width = min(fd, sturges) if fd else sturges
    return min(int(np.ceil(np.ptp(x)) / width) if width else 1, max_bins)


def detect_edge(x, x_range=None, percent=0.1, cutoff=0.5, nbins="auto", allow_both=False, allow_none=False,
                allowed_bounds=("lower", "upper"), test=None):
    """Which bound of a parameter the samples pile up at: ``"lower"``, ``"upper"``, ``"both"`` or
    ``False`` (``utils/rescaling.py:106-193``).  The density histogram's mass in the outer ``percent``
    of the bins on each side is compared with ``cutoff`` times the mass of the fullest bin; the mode
    inside an outer band decides first, otherwise the heavier side."""
    sides = ["lower", "upper"]
    if test is not None:  # the reference's hook for exercising every branch
        return False if (test in sides and test not in allowed_bounds) else test
    if not all(b in sides for b in allowed_bounds):
        raise RuntimeError(f"Unknown allowed bounds: {allowed_bounds}")
    if nbins == "auto":
        nbins = _auto_bins(x)
    hist, edges = np.histogram(x, bins=nbins, density=True, range=x_range)
    width = edges[1] - edges[0]
    n = max(int(len(edges) * percent), 1)
    mass = np.array([np.sum(hist[:n]), np.sum(hist[-n:])]) * width
    top = int(np.argmax(hist))
    top_mass = hist[top] * width
    keep = [i for i, b in enumerate(sides) if b in allowed_bounds]
    sides, mass = [sides[i] for i in keep], mass[keep]
    if np.all(mass > cutoff * top_mass) and allow_both and len(sides) > 1:
        return "both"
    if top < n and "lower" in sides:
        return sides[0]
    if top >= (len(edges) - n) and "upper" in sides:
        return sides[-1]
    if not np.any(mass > cutoff * top_mass) and allow_none:
        return False
    return sides[int(np.argmax(mass))]


def _concat_structured(a, b):
    return np.concatenate([a, b])


class _Group:
    """One reparameterisation instance of the reference: a kind and its parameters."""

    def __init__(self, name, kind, parameters, prior_bounds, kwargs, rng=None):
        self.rng = rng
        self.name = name
        self.kind = kind
        self.parameters = list(parameters)
        kw = dict(kwargs)
        if kind == "scale_shift":
            self.estimate_scale = bool(kw.pop("estimate_scale", False))
            self.estimate_shift = bool(kw.pop("estimate_shift", False))
            scale, shift = kw.pop("scale", None), kw.pop("shift", None)
            if scale is None and not self.estimate_scale:
                raise RuntimeError("Must specify a scale or enable estimate_scale")  # rescale.py:176
            self.scale = {p: 1.0 for p in self.parameters} if self.estimate_scale else self._values(scale, "scale")
            if self.estimate_shift:
                self.shift = {p: 0.0 for p in self.parameters}
            elif shift:
                self.shift = self._values(shift, "shift")
            else:
                self.shift = None
            self.update_enabled = self.estimate_scale or self.estimate_shift
            # a rescaling in front of the affine map or behind it (rescale.py:180-181, base.py): the ones with a
            # device map - logit / log / inverse Gaussian CDF in front, Gaussian CDF behind, not both
            pre, post = kw.pop("pre_rescaling", None), kw.pop("post_rescaling", None)
            pre = "inv_gaussian_cdf" if pre == "inverse_gaussian_cdf" else pre
            if pre not in (None, "logit", "log", "inv_gaussian_cdf") or post not in (None, "gaussian_cdf") or (pre and post):
                raise NotImplementedError(f"pre_rescaling={pre!r}, post_rescaling={post!r} has no device map")
            self.ss_pre, self.ss_post = pre, post
            self._pre = {None: None, "logit": (_logit, _sigmoid), "log": (_log, _exp),
                         "inv_gaussian_cdf": (_inverse_gaussian_cdf, _gaussian_cdf)}[pre]
            self._post = {None: None, "gaussian_cdf": (_gaussian_cdf, _inverse_gaussian_cdf)}[post]
        elif kind == "to_bounds":
            missing = [p for p in self.parameters if p not in prior_bounds or not np.all(np.isfinite(prior_bounds[p]))]
            if missing:
                raise RuntimeError(f"{name} requires finite prior bounds, missing for {missing}")
            self.prior_bounds = {p: np.asarray(prior_bounds[p], dtype=np.float64) for p in self.parameters}
            rb = kw.pop("rescale_bounds", None)
            if rb is None:
                self.rescale_bounds = {p: [-1, 1] for p in self.parameters}
            elif isinstance(rb, dict):
                self.rescale_bounds = {p: list(rb[p]) for p in self.parameters}
            else:
                self.rescale_bounds = {p: list(rb) for p in self.parameters}
            self.update_enabled = bool(kw.pop("update_bounds", True))
            self.dequantise = bool(kw.pop("dequantise", False))
            if self.dequantise:
                if self.rng is None:
                    raise RuntimeError("the dequantise reparameterisation needs the proposal's random generator")
                if self.update_enabled:
                    # (accepted by the reference's constructor "but not recommended" - its update then calls len() on
                    # a scalar minimum: discrete.py:73-75, rescale.py:686-700)
                    raise NotImplementedError("dequantise with update_bounds=True")
                extra = [k for k in ("boundary_inversion", "inversion_type", "detect_edges", "detect_edges_kwargs",
                                     "offset") if k in kw]
                if extra:
                    raise TypeError(f"Unknown arguments for reparameterisation {name}: {sorted(extra)}")
            self.post = kw.pop("post_rescaling", None)
            if self.post not in (None, "logit", "log"):
                raise NotImplementedError(f"post_rescaling={self.post!r} has no device map")
            if self.post in ("logit", "log"):
                if self.update_enabled:
                    raise RuntimeError("Cannot use log or logit with update bounds")  # rescale.py:491-495
                self.rescale_bounds = {p: [0, 1] for p in self.parameters}
            # boundary inversion (rescale.py:425-463): {parameter: "split" | "duplicate"} or False
            inversion, itype = kw.pop("boundary_inversion", None), kw.pop("inversion_type", "split")
            detect_edges, edge_kwargs = bool(kw.pop("detect_edges", False)), kw.pop("detect_edges_kwargs", None)
            if inversion is None or inversion is False:
                self.inversion = False
            elif inversion is True:
                self.inversion = {p: itype for p in self.parameters}
            elif isinstance(inversion, list):
                self.inversion = {p: itype for p in inversion}
            elif isinstance(inversion, dict):
                self.inversion = dict(inversion)
            else:
                raise TypeError(f"boundary_inversion must be a list, dict or bool. Got type: {type(inversion).__name__}")
            if detect_edges and not self.inversion:
                raise RuntimeError("Must enable boundary inversion to use detect edges")
            self.edges = None
            if self.inversion:
                if self.post is not None:
                    raise NotImplementedError("boundary inversion with a post-rescaling has no device map")
                bad = [t for t in self.inversion.values() if t not in ("split", "duplicate")]
                if bad or any(p not in self.parameters for p in self.inversion):
                    raise ValueError(f"unknown boundary inversion configuration: {self.inversion}")
                for p in self.inversion:
                    self.rescale_bounds[p] = [0, 1]
                self.edges = {p: None for p in self.parameters}
                # utils/rescaling.py:196-222: with edge detection "no edge" is allowed, without it the
                # heavier side is always taken
                self.edge_kwargs = dict(cutoff=0.5)
                given = dict(edge_kwargs or {})
                given["allow_none"] = detect_edges
                if not detect_edges:
                    given["cutoff"] = 0.0
                self.edge_kwargs.update(given)
                if detect_edges:
                    self.update_enabled = True  # rescale.py:451
            if kw.pop("offset", False):  # rescale.py:467-475
                self.offsets = {p: b[0] + np.ptp(b) / 2 for p, b in self.prior_bounds.items()}
            else:
                self.offsets = {p: 0.0 for p in self.parameters}
            self.set_bounds()
        elif kind == "null":
            self.update_enabled = False
        elif kind == "angle":
            # ONE angle -> Cartesian coordinates with an auxiliary radius drawn from chi(2) (angle.py:57-71).  The
            # reference's two-parameter form (an angle and a radial parameter OF THE MODEL) is not built.
            if len(self.parameters) != 1:
                raise NotImplementedError(
                    f"{name}: one angle per entry (the angle + radial-parameter form of the reference's Angle couples "
                    f"two model parameters and has no device map); got {self.parameters}")
            p = self.parameters[0]
            if p not in prior_bounds or not np.all(np.isfinite(prior_bounds[p])):
                raise RuntimeError(f"{name} requires finite prior bounds, missing for ['{p}']")
            self.prior_bounds = {p: np.asarray(prior_bounds[p], dtype=np.float64)}
            scale = kw.pop("scale", 1.0)
            self.angle_scale = float(2.0 * np.pi / np.ptp(self.prior_bounds[p])) if scale is None else float(scale)
            self.zero_bound = bool(self.prior_bounds[p][0] == 0)  # angle.py:79-82: modulo 2 pi on the way back
            # ToCartesian (angle.py:189-232): the parameter on [0, 1] first, mirrored - "split": half of the rows,
            # "duplicate": every row twice, "half": not at all - then times the scale; back: |angle| / scale on [0, 1]
            self.cartesian = bool(kw.pop("cartesian", False))
            if self.cartesian:
                self.mode = kw.pop("mode", "split")
                if self.mode not in ("duplicate", "split", "half"):
                    raise RuntimeError(f"Unknown mode: {self.mode}")
                self.zero_bound = False  # angle.py:200
            self.update_enabled = False
        elif kind == "angle_pair":
            # two angles -> Cartesian coordinates with an auxiliary radius drawn from chi(3) (angle.py:214-376); the
            # three-parameter form (a radial parameter OF THE MODEL) is not built
            if len(self.parameters) != 2:
                raise NotImplementedError(
                    f"{name}: a pair of angles (the reference's angles + radial-parameter form has no device map); "
                    f"got {self.parameters}")
            missing = [p for p in self.parameters if p not in prior_bounds or not np.all(np.isfinite(prior_bounds[p]))]
            if missing:
                raise RuntimeError(f"{name} requires finite prior bounds, missing for {missing}")
            self.prior_bounds = {p: np.asarray(prior_bounds[p], dtype=np.float64) for p in self.parameters}
            widths = np.ptp([self.prior_bounds[p] for p in self.parameters], axis=1)
            try:  # which angle runs along the horizon (range 2 pi), which is the vertical one (range pi)
                hz = int(np.where(widths == (2 * np.pi))[0][0])
                vt = int(np.where(widths == np.pi)[0][0])
            except IndexError:
                raise ValueError(f"Invalid prior ranges: {self.prior_bounds}. "
                                 "Parameters must be defined over a range of pi and 2 pi!")
            self.parameters = [self.parameters[hz], self.parameters[vt]]
            h_bounds, v_bounds = self.prior_bounds[self.parameters[0]], self.prior_bounds[self.parameters[1]]
            if np.array_equal(h_bounds, [0, 2 * np.pi]):
                self.modulo_2pi = True
            elif np.array_equal(h_bounds, [-np.pi, np.pi]):
                self.modulo_2pi = False
            else:
                raise ValueError(f"Prior bounds for {self.parameters[0]} must be [0, 2pi] or [-pi, pi]. "
                                 f"Received: {h_bounds}")
            conventions = {"az-zen": [0, np.pi], "ra-dec": [-np.pi / 2, np.pi / 2]}
            convention = kw.pop("convention", None)
            if convention is None:
                found = [c for c, b in conventions.items() if np.array_equal(v_bounds, b)]
                if not found:
                    raise RuntimeError(f"Could not determine convention for: {self.parameters}!")
                convention = found[0]
            elif convention not in conventions:
                raise ValueError(f"Unknown convention: `{convention}`. Choose from: {list(conventions)}.")
            elif not np.array_equal(v_bounds, conventions[convention]):
                raise ValueError(f"Prior bounds for {self.parameters[1]} must be {conventions[convention]} for the "
                                 f"{convention} convention. Received: {v_bounds}.")
            self.convention = convention
            self.update_enabled = False
        if kw:
            raise TypeError(f"Unknown arguments for reparameterisation {name}: {sorted(kw)}")

    def _values(self, value, what):
        """(``rescale.py:206-231``)."""
        if isinstance(value, (int, float)):
            return {p: float(value) for p in self.parameters}
        if isinstance(value, list):
            if len(value) != len(self.parameters):
                raise RuntimeError(f"{what} list is a different length to the parameters.")
            return {p: float(v) for p, v in zip(self.parameters, value)}
        if isinstance(value, dict):
            if set(value) != set(self.parameters):
                raise RuntimeError(f"Mismatched parameters with {what} dict")
            return {p: float(v) for p, v in value.items()}
        raise TypeError(f"{what} input must be an instance of int, list or dict")

    def set_bounds(self):
        """(``rescale.py:702-718``)."""
        self.factor = {p: np.ptp(self.rescale_bounds[p]) for p in self.parameters}
        self.rshift = {p: self.rescale_bounds[p][0] for p in self.parameters}
        self.bounds = {p: self.prior_bounds[p] - self.offsets[p] for p in self.parameters}
        if getattr(self, "dequantise", False):
            # discrete.py:66-71: RescaleToBounds.set_bounds passes the prior bounds through the pre-rescaling first -
            # two draws of two uniforms per parameter that are thrown away (the generator moves as the reference's
            # does) - then the bounds become [min, max + 1]
            for _ in range(2 * len(self.parameters)):
                self.rng.random(2)
            self.bounds = {p: np.array([b[0], b[1] + 1]) for p, b in self.prior_bounds.items()}

    def update(self, x):
        if self.kind == "to_bounds":
            self.reset_inversion()  # RescaleToBounds.update always forgets the edges (rescale.py:718-725)
        if not self.update_enabled:
            return
        if self.kind == "scale_shift":  # rescale.py:293-304
            for p in self.parameters:
                x_pre = self._pre[0](x[p])[0] if self._pre else x[p]
                if self.estimate_scale:
                    self.scale[p] = np.std(x_pre)
                if self.estimate_shift:
                    self.shift[p] = np.mean(x_pre)
        elif self.kind == "to_bounds":  # rescale.py:720-741
            self.bounds = {
                p: [np.min(x[p]) - self.offsets[p], np.max(x[p]) - self.offsets[p]] for p in self.parameters
            }

    def reset_inversion(self):
        """Forget the detected edges (``rescale.py:673-676``): the next forward pass detects them again."""
        if getattr(self, "edges", None):
            self.edges = {p: None for p in self.parameters}

    def reset(self):
        if self.kind == "scale_shift":  # rescale.py:306-312
            for p in self.parameters:
                if self.estimate_scale:
                    self.scale[p] = 1.0
                if self.estimate_shift:
                    self.shift[p] = 0.0
        elif self.kind == "to_bounds":
            self.reset_inversion()
            self.set_bounds()

    def _mirror(self, x, x_prime, log_j, p, compute_radius, test):
        """Forward map of a parameter with boundary inversion (``rescale.py:541-583``): [0, 1] from the
        detected edge, then mirrored about it - half of the rows flipped, or every row doubled (also
        whenever the radius of the latent contour is being computed)."""
        pp = p + "_prime"
        if self.edges[p] is None:
            self.edges[p] = detect_edge(x_prime[pp], test=test, **self.edge_kwargs)
        lo, hi = self.bounds[p]
        if self.edges[p]:
            x_prime[pp] = (x_prime[pp] - self.offsets[p] - lo) / (hi - lo)
            log_j += -np.log(hi - lo)
            if self.edges[p] == "upper":
                x_prime[pp] = 1 - x_prime[pp]
            if self.inversion[p] == "duplicate" or compute_radius:
                mirrored = x_prime.copy()
                mirrored[pp] *= -1
                x_prime = _concat_structured(x_prime, mirrored)
                x = _concat_structured(x, x)
                log_j = np.concatenate([log_j, log_j])
            else:
                if self.rng is None:
                    raise RuntimeError("split boundary inversion needs the proposal's random generator")
                flip = self.rng.choice(x_prime.size, x_prime.size // 2, replace=False)
                x_prime[pp][flip] *= -1
        else:
            x_prime[pp] = (2.0 * (x_prime[pp] - self.offsets[p] - lo) / (hi - lo)) - 1
            log_j += np.log(2) - np.log(hi - lo)
        return x, x_prime, log_j

    # x -> x'
    @property
    def prime_names(self):
        """The x' columns of the group, in order."""
        if self.kind == "angle":
            return [self.parameters[0] + "_x", self.parameters[0] + "_y"]
        if self.kind == "angle_pair":
            m = "_".join(self.parameters)
            return [f"{m}_x", f"{m}_y", f"{m}_z"]
        return [p + "_prime" for p in self.parameters]

    @property
    def aux_names(self):
        """Auxiliary parameters the inverse map leaves in x (the radius of an angle / a pair: angle.py:61-63, 308-312)."""
        if self.kind == "angle":
            return [self.parameters[0] + "_radial"]
        if self.kind == "angle_pair":
            return ["_".join(self.parameters) + "_radial"]
        return []

    def log_prior(self, x):
        """Prior of the auxiliary parameters (angle.py:181-183, 529-537: chi(2) / chi(3) on the radius); 0 otherwise."""
        if self.kind not in ("angle", "angle_pair"):
            return 0.0
        from scipy import stats

        return stats.chi(2 if self.kind == "angle" else 3).logpdf(x[self.aux_names[0]])

    def forward(self, x, x_prime, log_j, compute_radius=False, test=None):
        """Returns ``(x, x_prime, log_j)``: a duplicating boundary inversion replaces all three."""
        if self.kind == "angle":  # angle.py:134-151: the radius is drawn from the proposal's generator
            from scipy import stats

            if self.rng is None:
                raise RuntimeError("the angle reparameterisation needs the proposal's random generator")
            p = self.parameters[0]
            if self.cartesian:  # angle.py:203-221
                lo, hi = self.prior_bounds[p]
                angle = (x[p] - lo) / (hi - lo)
                log_j = log_j - np.log(hi - lo)
                if self.mode == "duplicate" or compute_radius:
                    angle = np.concatenate([angle, -angle])
                    x = _concat_structured(x, x)
                    x_prime = _concat_structured(x_prime, x_prime)
                    log_j = np.concatenate([log_j, log_j])
                elif self.mode == "split":
                    neg = self.rng.choice(angle.size, angle.size // 2, replace=False)
                    angle[neg] *= -1
                angle = angle * self.angle_scale
            else:
                angle = x[p] * self.angle_scale
            r = stats.chi(2).rvs(size=x.size, random_state=self.rng)
            x_prime[p + "_x"] = r * np.cos(angle)
            x_prime[p + "_y"] = r * np.sin(angle)
            log_j += np.log(r)
            return x, x_prime, log_j
        if self.kind == "angle_pair":  # angle.py:409-432, 509-521
            from scipy import stats

            if self.rng is None:
                raise RuntimeError("the angle-pair reparameterisation needs the proposal's random generator")
            r = stats.chi(3).rvs(size=x.size, random_state=self.rng)
            horizontal, vertical = x[self.parameters[0]], x[self.parameters[1]]
            px, py, pz = self.prime_names
            if self.convention == "az-zen":
                x_prime[px] = r * np.sin(vertical) * np.cos(horizontal)
                x_prime[py] = r * np.sin(vertical) * np.sin(horizontal)
                x_prime[pz] = r * np.cos(vertical)
                log_j += 2 * np.log(r) + np.log(np.sin(vertical))
            else:
                x_prime[px] = r * np.cos(vertical) * np.cos(horizontal)
                x_prime[py] = r * np.cos(vertical) * np.sin(horizontal)
                x_prime[pz] = r * np.sin(vertical)
                log_j += 2 * np.log(r) + np.log(np.cos(vertical))
            return x, x_prime, log_j
        for p in self.parameters:
            pp = p + "_prime"
            if self.kind == "to_bounds" and self.inversion and p in self.inversion:
                x_prime[pp] = x[p]
                x, x_prime, log_j = self._mirror(x, x_prime, log_j, p, compute_radius, test)
                continue
            if self.kind == "null":
                x_prime[pp] = x[p]
            elif self.kind == "scale_shift":  # rescale.py:233-261
                v = x[p]
                if self._pre:
                    v, lj = self._pre[0](v)
                    log_j += lj
                if self.shift:
                    v = (v - self.shift[p]) / self.scale[p]
                else:
                    v = v / self.scale[p]
                log_j -= np.log(np.abs(self.scale[p]))
                if self._post:
                    v, lj = self._post[0](v)
                    log_j += lj
                x_prime[pp] = v
            else:  # rescale.py:515-535, 639-671
                lo, hi = self.bounds[p]
                xin = x[p] + self.rng.random(len(x)) if self.dequantise else x[p]  # discrete.py:73-75
                v = self.factor[p] * ((xin - self.offsets[p] - lo) / (hi - lo)) + self.rshift[p]
                log_j += -np.log(hi - lo) + np.log(self.factor[p])
                if self.post == "logit":
                    v, lj = _logit(v)
                    log_j += lj
                elif self.post == "log":  # utils/rescaling.py:359-375
                    with np.errstate(divide="ignore", invalid="ignore"):
                        v = np.log(v)
                    log_j += -v
                x_prime[pp] = v
        return x, x_prime, log_j

    # x' -> x
    def inverse(self, x_prime, x, log_j):
        if self.kind == "angle":  # angle.py:153-179
            p = self.parameters[0]
            cx, cy = x_prime[p + "_x"], x_prime[p + "_y"]
            radial = np.sqrt(cx ** 2 + cy ** 2)
            if self.zero_bound:
                angle = np.arctan2(cy, cx) % (2.0 * np.pi) / self.angle_scale
            else:
                angle = np.arctan2(cy, cx) / self.angle_scale
            log_j -= np.log(radial)
            x[p + "_radial"] = radial
            if self.cartesian:  # angle.py:223-232
                lo, hi = self.prior_bounds[p]
                angle = (hi - lo) * np.abs(angle) + lo
                log_j += np.log(hi - lo)
            x[p] = angle
            return log_j
        if self.kind == "angle_pair":  # angle.py:434-507
            px, py, pz = (x_prime[n] for n in self.prime_names)
            radial = np.sqrt(px ** 2.0 + py ** 2.0 + pz ** 2.0)
            horizontal = np.arctan2(py, px)
            if self.modulo_2pi:
                horizontal = horizontal % (2.0 * np.pi)
            if self.convention == "az-zen":
                vertical = np.arctan2(np.sqrt(px ** 2.0 + py ** 2.0), pz)
                log_j += -2 * np.log(radial) - np.log(np.sin(vertical))
            else:
                vertical = np.arctan2(pz, np.sqrt(px ** 2.0 + py ** 2.0))
                log_j += -2 * np.log(radial) - np.log(np.cos(vertical))
            x[self.aux_names[0]] = radial
            x[self.parameters[0]] = horizontal
            x[self.parameters[1]] = vertical
            return log_j
        if self.kind == "to_bounds":
            params = reversed(self.parameters)  # rescale.py:674-677
        else:
            params = self.parameters
        for p in params:
            pp = p + "_prime"
            if self.kind == "null":
                x[p] = x_prime[pp]
            elif self.kind == "scale_shift":  # rescale.py:263-291
                v = x_prime[pp]
                if self._post:
                    v, lj = self._post[1](v)
                    log_j += lj
                if self.shift:
                    v = (v * self.scale[p]) + self.shift[p]
                else:
                    v = v * self.scale[p]
                log_j += np.log(np.abs(self.scale[p]))
                if self._pre:
                    v, lj = self._pre[1](v)
                    log_j += lj
                x[p] = v
            elif self.inversion and p in self.inversion:  # rescale.py:585-612
                v = x_prime[pp]
                lo, hi = self.bounds[p]
                if self.edges[p]:
                    v = np.where(v < 0.0, -v, v)
                    if self.edges[p] == "upper":
                        v = 1 - v
                    v = (hi - lo) * v + lo
                    log_j += np.log(hi - lo)
                else:
                    v = (hi - lo) * ((v + 1) / 2.0) + lo
                    log_j += np.log(hi - lo) - np.log(2)
                x[p] = v + self.offsets[p]
            else:  # rescale.py:537-547, 673-700
                v = x_prime[pp]
                if self.post == "logit":
                    v, lj = _sigmoid(v)
                    log_j += lj
                elif self.post == "log":  # utils/rescaling.py:378-395
                    with np.errstate(over="ignore"):
                        lj = v
                        v = np.exp(v)
                    log_j += lj
                lo, hi = self.bounds[p]
                v = (hi - lo) * (v - self.rshift[p]) / self.factor[p] + lo
                log_j += np.log(hi - lo) - np.log(self.factor[p])
                x[p] = np.floor(v + self.offsets[p]) if self.dequantise else v + self.offsets[p]  # discrete.py:77-78
        return log_j


def _resolve(patterns, names):
    """Exact names or full-match regular expressions (``utils.py:271-299``)."""
    if isinstance(patterns, str):
        patterns = [patterns]
    out = []
    for pat in patterns:
        if pat in names:
            out.append(pat)
        else:
            out.extend(n for n in names if re.fullmatch(pat, n))
    return list(dict.fromkeys(out))


def _lookup(name):
    if not isinstance(name, str) or name.lower() not in _KNOWN:
        raise NotImplementedError(
            f"Reparameterisation {name!r} is not a per-parameter device map; supported: {sorted(_KNOWN)}"
        )
    kind, kwargs = _KNOWN[name.lower()]
    return kind, dict(kwargs)


class ReparameterisationTable:
    """The proposal's x <-> x' map: ordered groups + the tables the draw kernel reads."""

    def __init__(self, names, bounds, reparameterisations=None, fallback="z-score", rng=None, reverse_order=False):
        self.rng = rng  # split boundary inversion draws the rows it flips from the proposal's generator
        #: ``reverse_reparameterisations`` of the proposal (``proposal/flowproposal/base.py:120,589``;
        #: ``reparameterisations/combined.py:58-72``): the groups are applied last-to-first on the way to x' and
        #: first-to-last on the way back - for the one-to-one maps here that is the order in which the log-Jacobian
        #: is summed and in which a boundary inversion meets the groups around it
        self.reverse_order = bool(reverse_order)
        self.names = list(names)
        bounds = {n: np.asarray(bounds[n], dtype=np.float64) for n in self.names} if bounds else {}
        self.groups = []
        cfg = reparameterisations
        if cfg is None:
            cfg = {}
        elif isinstance(cfg, str):
            cfg = {cfg: {"parameters": list(self.names)}}
        elif not isinstance(cfg, dict):
            raise TypeError(
                f"Reparameterisations must be a dictionary, string or None, received {type(cfg).__name__}"
            )
        for key, spec in cfg.items():
            if key in self.names:  # {parameter: name | dict}
                if isinstance(spec, str) or spec is None:
                    name, kwargs, params = spec, {}, [key]
                elif isinstance(spec, dict):
                    spec = dict(spec)
                    name = spec.pop("reparameterisation", None)
                    if name is None:
                        raise RuntimeError(f"No reparameterisation found for {key}.")
                    params = spec.pop("input_parameters", spec.pop("parameters", [key]))
                    params, kwargs = _resolve(params, self.names), spec
                else:
                    raise TypeError(f"Unknown config type for: {key}.")
                if name is None:
                    name = "null"
            else:  # {name: parameter | [patterns] | dict}
                name = key
                if isinstance(spec, (str, list)):
                    params, kwargs = _resolve(spec, self.names), {}
                elif isinstance(spec, dict):
                    spec = dict(spec)
                    spec.pop("reparameterisation", None)
                    params = _resolve(spec.pop("input_parameters", spec.pop("parameters", [])), self.names)
                    kwargs = spec
                else:
                    raise TypeError(f"Unknown config type for: {key}.")
            self._add(name, params, bounds, kwargs)
        others = [n for n in self.names if n not in self.parameters]
        if others:
            self._add(fallback, others, bounds, {})
        # Columns.  x' keeps the order of model.names (an angle: its two Cartesian columns side by side); x gains the
        # auxiliary parameters (the radius of every angle: angle.py:61-63).  `device_names` is what the D' output
        # columns of the draw kernel hold: a model parameter per x' column, the auxiliary radius in the column of an
        # angle's y coordinate.
        wide = [g for g in self.groups if g.kind in ("angle", "angle_pair")]
        first = {min(g.parameters, key=self.names.index): g for g in wide}  # the group's columns sit where its first
        covered = {p for g in wide for p in g.parameters}                   # parameter (in model order) sits
        self.prime_names, self.device_names, self.aux_names = [], [], []
        # degrees of freedom of the chi prior each auxiliary radius carries (angle.py:181-183, 529-537)
        self.aux_dof = {a: (2 if g.kind == "angle" else 3) for g in wide for a in g.aux_names}
        for n in self.names:
            if n in first:
                g = first[n]
                self.prime_names += g.prime_names
                self.device_names += list(g.parameters) + g.aux_names
                self.aux_names += g.aux_names
            elif n not in covered:
                self.prime_names.append(n + "_prime")
                self.device_names.append(n)
        self.x_names = self.names + self.aux_names

    @property
    def to_prime_order(self):
        """Groups in the order they act on the way to x' (``combined.py:58-64``)."""
        return list(reversed(self.groups)) if self.reverse_order else list(self.groups)

    @property
    def from_prime_order(self):
        """... and on the way back (``combined.py:66-72``)."""
        return list(self.groups) if self.reverse_order else list(reversed(self.groups))

    @property
    def parameters(self):
        return [p for g in self.groups for p in g.parameters]

    def _add(self, name, params, bounds, kwargs):
        kind, defaults = _lookup(name)
        taken = set(self.parameters)
        clash = [p for p in params if p in taken]
        if clash:
            raise RuntimeError(f"Parameters {clash} already have a reparameterisation")
        if not params:
            return
        defaults.update(kwargs)
        self.groups.append(_Group(name, kind, params, bounds, defaults, rng=self.rng))

    @property
    def has_aux(self):
        """A map that changes the number of columns is present (angles)."""
        return bool(self.aux_names)

    def log_prior(self, x):
        """Prior of the auxiliary parameters (``combined.py``: the sum over the reparameterisations with
        ``has_prior``; ``proposal/flowproposal/base.py:1046-1049`` adds it to the model's)."""
        total = np.zeros(x.size)
        for g in self.groups:
            if g.kind in ("angle", "angle_pair"):
                total = total + g.log_prior(x)
        return total

    @property
    def should_update(self):
        return any(g.update_enabled for g in self.groups) or self.has_inversion

    def update(self, x):
        """Update the data-dependent constants from a set of points (``combined.py:219-230``): every group
        is updated and then APPLIED to the points before the next one is updated.  For the one-to-one maps
        that chain changes nothing; a boundary inversion detects its edges here (on the live points), draws
        the rows a split inversion flips from the generator - once here, once more in the ``rescale`` of the
        training data, as the reference does - and hands doubled arrays to the groups behind it."""
        if not self.has_inversion:
            for g in self.to_prime_order:
                g.update(x)
            return
        current = x.copy()
        x_prime = empty_structured_array(current.size, names=self.prime_names)
        log_j = np.zeros(current.size)
        for g in self.to_prime_order:
            g.update(current)
            current, x_prime, log_j = g.forward(current, x_prime, log_j)

    def reset(self):
        for g in self.groups:
            g.reset()

    @property
    def has_inversion(self):
        return any(g.kind == "to_bounds" and g.inversion for g in self.groups)

    def rescale(self, x, compute_radius=False, test=None):
        """x -> (x', log|J|)  (``proposal/flowproposal/base.py:996-1030``).  With a duplicating boundary
        inversion (or ``compute_radius``) x' has twice the rows of x per mirrored parameter."""
        x_prime = empty_structured_array(x.size, names=self.prime_names)
        log_j = np.zeros(x_prime.size)
        for g in self.to_prime_order:
            x, x_prime, log_j = g.forward(x, x_prime, log_j, compute_radius=compute_radius, test=test)
        for f in _NON_SAMPLING:
            x_prime[f] = x[f]
        return x_prime, log_j

    def inverse_rescale(self, x_prime):
        x = empty_structured_array(x_prime.size, names=self.x_names)
        log_j = np.zeros(x.size)
        for g in self.from_prime_order:
            log_j = g.inverse(x_prime, x, log_j)
        for f in _NON_SAMPLING:
            x[f] = x_prime[f]
        return x, log_j

    # -- device tables ------------------------------------------------------------

    @property
    def all_scale_shift(self):
        return all(g.kind == "scale_shift" and not (g.ss_pre or g.ss_post) for g in self.groups)

    def device_bounds(self, lower, upper):
        """Per device column: the model's bounds of its parameter; [0, inf) for an auxiliary radius."""
        lo = dict(zip(self.names, lower))
        hi = dict(zip(self.names, upper))
        return (np.array([lo.get(n, 0.0) for n in self.device_names], dtype=np.float64),
                np.array([hi.get(n, np.inf) for n in self.device_names], dtype=np.float64))

    def device_tables(self):
        """(kind int32[D], par float64[D,5], scale[D], shift[D], constant log|J|).

        The constant is the row-independent part of ``inverse_rescale``'s log-Jacobian,
        accumulated in the same order; the sigmoid terms are added per row by the kernel.
        """
        D = len(self.device_names)
        kind = np.zeros(D, dtype=np.int32)
        par = np.zeros((D, 5), dtype=np.float64)
        scale = np.ones(D)
        shift = np.zeros(D)
        col = {n: i for i, n in enumerate(self.device_names)}
        log_j = np.float64(0.0)
        for g in self.from_prime_order:
            params = list(reversed(g.parameters)) if g.kind == "to_bounds" else g.parameters
            if g.kind == "angle_pair":
                # columns i, i + 1, i + 2 hold x, y, z and give the horizontal angle, the vertical angle, the radius;
                # par = {modulo 2 pi?, 1 for ra-dec / 0 for az-zen}
                i = col[g.parameters[0]]
                kind[i], kind[i + 1], kind[i + 2] = RP_PAIR_H, RP_PAIR_V, RP_PAIR_R
                flags = [1.0 if g.modulo_2pi else 0.0, 1.0 if g.convention == "ra-dec" else 0.0, 0.0, 0.0, 0.0]
                par[i] = par[i + 1] = par[i + 2] = flags
                continue
            for p in params:
                i = col[p]
                if g.kind == "angle":
                    # column i: the x coordinate -> the angle, par = {scale, modulo 2 pi?}; column i + 1: the y
                    # coordinate -> the radius (and the row's -log(radius) of the Jacobian)
                    kind[i], kind[i + 1] = RP_ANGLE, RP_ANGLE_RADIAL
                    par[i] = [g.angle_scale, 1.0 if g.zero_bound else 0.0, 0.0, 0.0, 0.0]
                    if g.cartesian:  # {.., fold |angle| / scale onto [lo, lo + width]: flag, lo, width}
                        lo, hi = g.prior_bounds[p]
                        par[i][2:] = [1.0, lo, hi - lo]
                        log_j += np.log(hi - lo)
                elif g.kind == "null":
                    kind[i] = RP_NULL
                elif g.kind == "scale_shift":
                    kind[i] = {None: RP_SCALE_SHIFT, "logit": RP_SS_SIGMOID, "log": RP_SS_EXP,
                               "inv_gaussian_cdf": RP_SS_GAUSSIAN_CDF}[g.ss_pre]
                    if g.ss_post:
                        kind[i] = RP_INV_GAUSSIAN_CDF_SS
                    scale[i] = g.scale[p]
                    shift[i] = g.shift[p] if g.shift else 0.0
                    log_j += np.log(np.abs(g.scale[p]))
                elif g.inversion and p in g.inversion:
                    lo, hi = g.bounds[p]
                    if g.edges[p]:  # fold |x'|, measure from the edge: par = {lo, hi, upper?, -, offset}
                        kind[i] = RP_INVERT
                        par[i] = [lo, hi, 1.0 if g.edges[p] == "upper" else 0.0, 0.0, g.offsets[p]]
                        log_j += np.log(hi - lo)
                    else:  # no edge detected (yet): the plain [-1, 1] map
                        kind[i] = RP_MINUS_ONE_ONE
                        par[i] = [lo, hi, 0.0, 0.0, g.offsets[p]]
                        log_j += np.log(hi - lo) - np.log(2)
                else:
                    kind[i] = {None: RP_TO_BOUNDS, "logit": RP_TO_BOUNDS_LOGIT, "log": RP_TO_BOUNDS_LOG}[g.post]
                    if g.dequantise:
                        kind[i] += RP_DEQUANTISE - RP_TO_BOUNDS
                    lo, hi = g.bounds[p]
                    par[i] = [lo, hi, g.rshift[p], g.factor[p], g.offsets[p]]
                    log_j += np.log(hi - lo) - np.log(g.factor[p])
        return kind, par, scale, shift, float(log_j)
