"""Minimal user-model interface needed by the proposal path.

The reference's ``Model`` (``src/nessai/model.py:53``) stays user code and is
out of the MI355X scope.  The proposal only relies on
``names``, ``bounds``, ``in_bounds``, ``log_prior`` and ``log_likelihood``
(batch-evaluated); this module provides that surface plus the analytic
benchmark models of BASELINE.json's configs.

A model whose prior is uniform inside its bounds can say so with
``uniform_prior = True``; the proposal then evaluates ``log_prior`` as a
constant on the device instead of copying every candidate to the host.
"""

import numpy as np

from .livepoint import live_points_to_array


class Model:
    names = None
    bounds = None
    uniform_prior = False

    @property
    def dims(self):
        return len(self.names)

    @property
    def lower_bounds(self):
        return np.array([self.bounds[n][0] for n in self.names], dtype=np.float64)

    @property
    def upper_bounds(self):
        return np.array([self.bounds[n][1] for n in self.names], dtype=np.float64)

    def in_bounds(self, x):
        """True where every parameter is inside its prior bounds (``model.py:497-518``)."""
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
        return ~np.any([(x[n] < 0.0) | (x[n] > 1.0) for n in self.names], axis=0)

    def from_unit_hypercube(self, x):
        """Unit hypercube -> physical space (uniform priors: an affine map)."""
        if not self.uniform_prior:
            raise NotImplementedError("from_unit_hypercube needs a user implementation")
        out = x.copy()
        for n in self.names:
            lo, hi = self.bounds[n]
            out[n] = lo + (hi - lo) * x[n]
        return out

    def to_unit_hypercube(self, x):
        if not self.uniform_prior:
            raise NotImplementedError("to_unit_hypercube needs a user implementation")
        out = x.copy()
        for n in self.names:
            lo, hi = self.bounds[n]
            out[n] = (x[n] - lo) / (hi - lo)
        return out

    def batch_evaluate_log_prior_unit_hypercube(self, x):
        """Log-prior density in the unit hypercube: 0 inside, -inf outside."""
        with np.errstate(divide="ignore"):
            return np.log(self.in_unit_hypercube(x), dtype="float")

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
        for bounds in self.bounds.values():
            log_p -= np.log(bounds[1] - bounds[0])
        return log_p


class RosenbrockModel(UniformBoxModel):
    """n-D Rosenbrock likelihood on [-5, 5]^n (``examples/rosenbrock.py:16-45``)."""

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

    def log_likelihood(self, x):
        a = self.unstructured_view(x)
        return -0.5 * np.sum(a**2, axis=-1) - 0.5 * self.dims * np.log(2 * np.pi)
