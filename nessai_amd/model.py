"""Minimal user-model interface needed by the proposal path.

The reference's ``Model`` (``src/nessai/model.py:53``) stays user code and is
out of the MI355X scope.  The proposal only relies on
``names``, ``bounds``, ``in_bounds``, ``log_prior`` and ``log_likelihood``
(batch-evaluated); this module provides that surface plus the analytic
benchmark models of BASELINE.json's configs.

A model whose prior is uniform inside its bounds can say so with
``uniform_prior = True``; the proposal then evaluates ``log_prior`` as a
constant on the device instead of copying every candidate to the host.  A model
with a *separable* analytic prior can declare it per parameter in
``device_prior`` (``{"name": ("uniform",) | ("sine",) | ("cosine",) |
("power_law", alpha) | ("gaussian", mu, sigma)}``, densities normalised on the
parameter's bounds as bilby's priors are); the fused populate kernel then
evaluates it in float64 in the same pass.
"""

import math

import numpy as np

from .livepoint import copy_points, live_points_to_array


PRIOR_KINDS = {"uniform": 0, "sine": 1, "cosine": 2, "power_law": 3, "gaussian": 4}


def prior_term(spec, lo, hi):
    """(kind code, [a, b, log normalisation]) of one separable prior term on [lo, hi]."""
    kind = spec[0]
    if kind == "uniform":
        return 0, [0.0, 0.0, -math.log(hi - lo)]
    if kind == "sine":
        return 1, [0.0, 0.0, -math.log(math.cos(lo) - math.cos(hi))]
    if kind == "cosine":
        return 2, [0.0, 0.0, -math.log(math.sin(hi) - math.sin(lo))]
    if kind == "power_law":
        alpha = float(spec[1])
        if alpha == -1.0:
            norm = math.log(hi / lo)
        else:
            norm = (hi ** (alpha + 1) - lo ** (alpha + 1)) / (alpha + 1)
        return 3, [alpha, 0.0, -math.log(norm)]
    if kind == "gaussian":
        mu, sigma = float(spec[1]), float(spec[2])
        return 4, [mu, sigma, -math.log(sigma * math.sqrt(2 * math.pi))]
    raise ValueError(f"unknown device prior {kind!r}; choose from {sorted(PRIOR_KINDS)}")


def log_prior_term(spec, x, lo, hi):
    """numpy evaluation of the same term (host reference of the device formula)."""
    kind, par = prior_term(spec, lo, hi)
    with np.errstate(divide="ignore", invalid="ignore"):
        if kind == 1:
            return np.log(np.sin(x)) + par[2]
        if kind == 2:
            return np.log(np.cos(x)) + par[2]
        if kind == 3:
            return par[0] * np.log(x) + par[2]
        if kind == 4:
            return -0.5 * ((x - par[0]) / par[1]) ** 2 + par[2]
        return np.full_like(np.asarray(x, dtype=np.float64), par[2])


class Model:
    names = None
    bounds = None
    uniform_prior = False
    device_prior = None

    def device_prior_tables(self):
        """(kinds int32 [D], params float64 [D][3]) or None when no device prior is declared."""
        if self.device_prior is None:
            return None
        kinds, pars = [], []
        for n in self.names:
            k, par = prior_term(self.device_prior[n], *self.bounds[n])
            kinds.append(k)
            pars.append(par)
        return np.asarray(kinds, dtype=np.int32), np.asarray(pars, dtype=np.float64)

    @property
    def dims(self):
        return len(self.names)

    @property
    def lower_bounds(self):
        return np.array([self.bounds[n][0] for n in self.names], dtype=np.float64)

    @property
    def upper_bounds(self):
        return np.array([self.bounds[n][1] for n in self.names], dtype=np.float64)

    def _block(self, x):
        """(N, D) view of the parameter fields of ``x`` (None if they are not one f8 block)."""
        from .livepoint import parameter_block

        return parameter_block(x, self.names) if isinstance(x, np.ndarray) else None

    def in_bounds(self, x):
        """True where every parameter is inside its prior bounds (``model.py:497-518``)."""
        v = self._block(x)
        if v is not None:
            return ~((v < self.lower_bounds) | (v > self.upper_bounds)).any(axis=1)
        return ~np.any(
            [(x[n] < self.bounds[n][0]) | (x[n] > self.bounds[n][1]) for n in self.names],
            axis=0,
        )

    def unstructured_view(self, x):
        return live_points_to_array(x, self.names)

    def log_prior(self, x):
        raise NotImplementedError

    def log_likelihood(self, x):
        raise NotImplementedError

    def batch_evaluate_log_prior(self, x):
        return self.log_prior(x)

    def batch_evaluate_log_likelihood(self, x):
        return np.asarray(self.log_likelihood(x), dtype=np.float64)

    # -- unit hypercube (importance sampler; ``src/nessai/model.py:520-620``) ---------------

    def in_unit_hypercube(self, x):
        """True where every parameter is in [0, 1]."""
        v = self._block(x)
        if v is not None:
            return ~((v < 0.0) | (v > 1.0)).any(axis=1)
        return ~np.any([(x[n] < 0.0) | (x[n] > 1.0) for n in self.names], axis=0)

    def from_unit_hypercube(self, x):
        """Unit hypercube -> physical space (uniform priors: an affine map)."""
        if not self.uniform_prior:
            raise NotImplementedError("from_unit_hypercube needs a user implementation")
        out = copy_points(x)
        v, w = self._block(x), self._block(out)
        if v is not None and w is not None:
            lo, hi = self.lower_bounds, self.upper_bounds
            w[:] = lo + (hi - lo) * v
            return out
        for n in self.names:
            lo, hi = self.bounds[n]
            out[n] = lo + (hi - lo) * x[n]
        return out

    def to_unit_hypercube(self, x):
        if not self.uniform_prior:
            raise NotImplementedError("to_unit_hypercube needs a user implementation")
        out = copy_points(x)
        v, w = self._block(x), self._block(out)
        if v is not None and w is not None:
            lo, hi = self.lower_bounds, self.upper_bounds
            w[:] = (v - lo) / (hi - lo)
            return out
        for n in self.names:
            lo, hi = self.bounds[n]
            out[n] = (x[n] - lo) / (hi - lo)
        return out

    def log_prior_unit_hypercube(self, x):
        """Log-prior density in the unit hypercube: log 1 on [0, 1)^D, -inf outside - the reference's default, with
        its half-open support (``src/nessai/model.py:593-601``: ``(x < 0) | (x >= 1)``)."""
        v = self._block(x)
        if v is None:
            v = np.stack([np.asarray(x[n], dtype=np.float64) for n in self.names], axis=-1)
        with np.errstate(divide="ignore"):
            return np.log(~np.any((v < 0.0) | (v >= 1.0), axis=-1), dtype="float")

    def batch_evaluate_log_prior_unit_hypercube(self, x):
        """(``src/nessai/model.py:679-700``: the vectorised form of ``log_prior_unit_hypercube``.)"""
        return self.log_prior_unit_hypercube(x)

    def new_point(self, N=1, rng=None):
        """Draw from the prior box (uniform models)."""
        rng = rng if rng is not None else np.random.default_rng()
        from .livepoint import numpy_array_to_live_points

        lo, hi = self.lower_bounds, self.upper_bounds
        pts = numpy_array_to_live_points(rng.uniform(lo, hi, size=(N, self.dims)), self.names)
        pts["logP"] = self.log_prior(pts)
        return pts


class UniformBoxModel(Model):
    """Uniform prior on a box: ``log_prior = log(in_bounds) - sum log(width)``."""

    uniform_prior = True

    def __init__(self, dims, lower=-5.0, upper=5.0):
        self.names = [f"x_{d}" for d in range(dims)]
        self.bounds = {n: [float(lower), float(upper)] for n in self.names}

    def log_prior(self, x):
        with np.errstate(divide="ignore"):
            log_p = np.log(self.in_bounds(x), dtype="float")
        # -sum log(width), subtracted bound by bound like examples/rosenbrock.py:28-31 - on a scalar: 0 - c1 - c2
        # ... is the same float64 sequence for every in-bounds point (and -inf stays -inf), one array pass
        # instead of one per parameter
        total = np.float64(0.0)
        for bounds in self.bounds.values():
            total -= np.log(bounds[1] - bounds[0])
        log_p += total
        return log_p


class RosenbrockModel(UniformBoxModel):
    """n-D Rosenbrock likelihood on [-5, 5]^n (``examples/rosenbrock.py:16-45``)."""

    device_likelihood = ("rosenbrock",)  # evaluated on the accepted rows by nsa_log_likelihood

    def log_likelihood(self, x):
        a = self.unstructured_view(x)
        return -np.sum(
            100.0 * (a[..., 1:] - a[..., :-1] ** 2.0) ** 2.0 + (1.0 - a[..., :-1]) ** 2.0,
            axis=-1,
        )


class GaussianModel(UniformBoxModel):
    """Unit-Gaussian likelihood on [-10, 10]^n (style of ``examples/2d_gaussian.py:28-61``)."""

    def __init__(self, dims, lower=-10.0, upper=10.0):
        super().__init__(dims, lower, upper)
        self.device_likelihood = ("gaussian", None, None, 0.5 * self.dims * np.log(2 * np.pi))

    def log_likelihood(self, x):
        a = self.unstructured_view(x)
        return -0.5 * np.sum(a**2, axis=-1) - 0.5 * self.dims * np.log(2 * np.pi)


class CorrelatedGaussianModel(UniformBoxModel):
    """Zero-mean Gaussian likelihood with the Toeplitz covariance ``Sigma_ij = rho ** |i - j|`` on
    ``[-bound, bound]^n`` - the 64-D, rho = 0.5 workload of BASELINE config 4 (SURVEY.md section
    8d).  The inverse of an AR(1) covariance is tridiagonal, so the quadratic form is
    ``(sum x_i^2 + rho^2 sum_{interior} x_i^2 - 2 rho sum x_i x_{i+1}) / (1 - rho^2)`` and
    ``log|Sigma| = (n - 1) log(1 - rho^2)``.  The importance sampler works in the unit hypercube
    through the affine maps of ``Model.to_unit_hypercube`` / ``from_unit_hypercube``."""

    def __init__(self, dims=64, rho=0.5, bound=10.0):
        super().__init__(dims, -float(bound), float(bound))
        if not -1.0 < rho < 1.0:
            raise ValueError("rho must be in (-1, 1)")
        self.rho = float(rho)
        self._log_norm = -0.5 * (dims * np.log(2 * np.pi) + (dims - 1) * np.log1p(-self.rho**2))

    def covariance(self):
        idx = np.arange(self.dims)
        return self.rho ** np.abs(idx[:, None] - idx[None, :])

    def log_likelihood(self, x):
        a = self.unstructured_view(x)
        rho = self.rho
        quad = np.sum(a**2, axis=-1) + rho**2 * np.sum(a[..., 1:-1] ** 2, axis=-1)
        quad = quad - 2.0 * rho * np.sum(a[..., 1:] * a[..., :-1], axis=-1)
        return -0.5 * quad / (1.0 - rho**2) + self._log_norm


class SeparablePriorModel(Model):
    """Model whose prior is a product of the 1-D densities declared in ``device_prior``."""

    def log_prior(self, x):
        with np.errstate(divide="ignore"):
            log_p = np.log(self.in_bounds(x), dtype="float")
        for n in self.names:
            log_p = log_p + log_prior_term(self.device_prior[n], x[n], *self.bounds[n])
        return log_p


class GWSurrogateModel(SeparablePriorModel):
    """15-D analytic surrogate of a compact-binary likelihood (BASELINE.json config 5).

    Gaussian log-likelihood around a fiducial binary in bilby-style parameters with
    bilby-style priors: uniform (masses, phases, time), power-law luminosity distance,
    sine inclinations/tilts, cosine declination.  A stand-in for the reference's GW examples
    (``examples/gw/*.py``), whose bilby / lalsuite dependencies are not available.
    """

    def __init__(self):
        spec = [
            ("chirp_mass", (25.0, 35.0), ("uniform",), 30.0, 0.3),
            ("mass_ratio", (0.125, 1.0), ("uniform",), 0.8, 0.08),
            ("a_1", (0.0, 0.99), ("uniform",), 0.3, 0.15),
            ("a_2", (0.0, 0.99), ("uniform",), 0.3, 0.2),
            ("tilt_1", (0.0, np.pi), ("sine",), 1.0, 0.4),
            ("tilt_2", (0.0, np.pi), ("sine",), 1.5, 0.5),
            ("phi_12", (0.0, 2 * np.pi), ("uniform",), 3.0, 1.0),
            ("phi_jl", (0.0, 2 * np.pi), ("uniform",), 3.0, 1.0),
            ("luminosity_distance", (100.0, 5000.0), ("power_law", 2.0), 1500.0, 300.0),
            ("dec", (-np.pi / 2, np.pi / 2), ("cosine",), -0.3, 0.15),
            ("ra", (0.0, 2 * np.pi), ("uniform",), 1.4, 0.2),
            ("theta_jn", (0.0, np.pi), ("sine",), 0.6, 0.25),
            ("psi", (0.0, np.pi), ("uniform",), 1.5, 0.5),
            ("phase", (0.0, 2 * np.pi), ("uniform",), 3.0, 1.2),
            ("geocent_time", (-0.1, 0.1), ("uniform",), 0.0, 0.005),
        ]
        self.names = [s[0] for s in spec]
        self.bounds = {s[0]: [float(s[1][0]), float(s[1][1])] for s in spec}
        self.device_prior = {s[0]: s[2] for s in spec}
        self._mu = np.array([s[3] for s in spec])
        self._sigma = np.array([s[4] for s in spec])
        self.device_likelihood = ("gaussian", self._mu, self._sigma, 0.0)

    def log_likelihood(self, x):
        a = self.unstructured_view(x)
        return -0.5 * np.sum(((a - self._mu) / self._sigma) ** 2, axis=-1)

    def new_point(self, N=1, rng=None):
        """Rejection-sample the separable prior inside the bounds."""
        rng = rng if rng is not None else np.random.default_rng()
        from .livepoint import numpy_array_to_live_points

        lo, hi = self.lower_bounds, self.upper_bounds
        out = np.empty((0, self.dims))
        while len(out) < N:
            u = rng.uniform(lo, hi, size=(4 * N, self.dims))
            pts = numpy_array_to_live_points(u, self.names)
            lp = self.log_prior(pts)
            lp_max = sum(
                np.max(log_prior_term(self.device_prior[n], np.linspace(*self.bounds[n], 513)[1:-1], *self.bounds[n]))
                for n in self.names
            )
            keep = np.log(rng.random(len(u))) < lp - lp_max
            out = np.concatenate([out, u[keep]])
        pts = numpy_array_to_live_points(out[:N], self.names)
        pts["logP"] = self.log_prior(pts)
        return pts
