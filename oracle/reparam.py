"""CPU restatement of the reference's per-parameter reparameterisations.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): used by ``tests/`` as the checker of the
device epilogue's reparameterisation table, never by the product path.

Pinned against the reference's own classes: ``tests/golden/make_reparam_goldens.py`` imports
``nessai.reparameterisations`` (it has no glasflow dependency) in this container and stores
inputs and outputs in ``tests/golden/reparam_goldens.npz``; ``tests/test_reparam.py`` checks this
restatement against them bit for bit.

Follows ``src/nessai/reparameterisations``: ``ScaleAndShift`` (``rescale.py:119-312``),
``RescaleToBounds`` without boundary inversion (``rescale.py:321-741``), ``NullReparameterisation``
(``null.py``), ``CombinedReparameterisation`` ordering (``combined.py:59-72, 154-192``), and the
``logit`` / ``sigmoid`` pair (``utils/rescaling.py:290-337``).
"""

import numpy as np


def logit(x):  # utils/rescaling.py:290-315 (eps=None)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.log(x) - np.log1p(-x), -np.log(x) - np.log1p(-x)


def sigmoid(x):  # utils/rescaling.py:318-337
    with np.errstate(over="ignore", divide="ignore"):
        y = np.divide(1, 1 + np.exp(-x))
        return y, np.log(y) + np.log1p(-y)


class OracleScaleAndShift:
    def __init__(self, parameters, scale=None, shift=None, estimate_scale=False, estimate_shift=False):
        self.parameters = list(parameters)
        self.estimate_scale, self.estimate_shift = estimate_scale, estimate_shift
        self.scale = {p: 1.0 for p in parameters} if estimate_scale else dict(scale)
        if estimate_shift:
            self.shift = {p: 0.0 for p in parameters}
        else:
            self.shift = dict(shift) if shift else None

    def update(self, x):  # rescale.py:293-304
        for p in self.parameters:
            if self.estimate_scale:
                self.scale[p] = np.std(x[p])
            if self.estimate_shift:
                self.shift[p] = np.mean(x[p])

    def reparameterise(self, x, x_prime, log_j):  # rescale.py:233-261
        for p in self.parameters:
            v = x[p]
            v = (v - self.shift[p]) / self.scale[p] if self.shift else v / self.scale[p]
            x_prime[p + "_prime"] = v
            log_j -= np.log(np.abs(self.scale[p]))
        return log_j

    def inverse_reparameterise(self, x, x_prime, log_j):  # rescale.py:263-291
        for p in self.parameters:
            v = x_prime[p + "_prime"]
            v = (v * self.scale[p]) + self.shift[p] if self.shift else v * self.scale[p]
            log_j += np.log(np.abs(self.scale[p]))
            x[p] = v
        return log_j


class OracleRescaleToBounds:
    def __init__(self, parameters, prior_bounds, rescale_bounds=None, update_bounds=True, offset=False,
                 post_rescaling=None):
        self.parameters = list(parameters)
        self.prior_bounds = {p: np.asarray(prior_bounds[p], dtype=float) for p in parameters}
        rb = [-1, 1] if rescale_bounds is None else rescale_bounds
        self.rescale_bounds = {p: list(rb) for p in parameters}
        self.post = post_rescaling
        if post_rescaling in ("logit", "log"):  # rescale.py:479-497
            assert not update_bounds
            self.rescale_bounds = {p: [0, 1] for p in parameters}
        self._update = update_bounds
        if offset:  # rescale.py:467-475
            self.offsets = {p: b[0] + np.ptp(b) / 2 for p, b in self.prior_bounds.items()}
        else:
            self.offsets = {p: 0.0 for p in parameters}
        # rescale.py:702-718
        self.factor = {p: np.ptp(self.rescale_bounds[p]) for p in parameters}
        self.rshift = {p: self.rescale_bounds[p][0] for p in parameters}
        self.bounds = {p: b - self.offsets[p] for p, b in self.prior_bounds.items()}

    def update(self, x):  # rescale.py:720-741
        if self._update:
            self.bounds = {
                p: [np.min(x[p]) - self.offsets[p], np.max(x[p]) - self.offsets[p]] for p in self.parameters
            }

    def reparameterise(self, x, x_prime, log_j):  # rescale.py:639-671, 515-535
        for p in self.parameters:
            lo, hi = self.bounds[p]
            v = self.factor[p] * ((x[p] - self.offsets[p] - lo) / (hi - lo)) + self.rshift[p]
            log_j += -np.log(hi - lo) + np.log(self.factor[p])
            if self.post == "logit":
                v, lj = logit(v)
                log_j += lj
            elif self.post == "log":  # utils/rescaling.py:359-375
                with np.errstate(divide="ignore", invalid="ignore"):
                    v = np.log(v)
                log_j += -v
            x_prime[p + "_prime"] = v
        return log_j

    def inverse_reparameterise(self, x, x_prime, log_j):  # rescale.py:673-700, 537-547
        for p in reversed(self.parameters):
            v = x_prime[p + "_prime"]
            if self.post == "logit":
                v, lj = sigmoid(v)
                log_j += lj
            elif self.post == "log":  # utils/rescaling.py:378-395
                with np.errstate(over="ignore"):
                    log_j += v
                    v = np.exp(v)
            lo, hi = self.bounds[p]
            v = (hi - lo) * (v - self.rshift[p]) / self.factor[p] + lo
            log_j += np.log(hi - lo) - np.log(self.factor[p])
            x[p] = v + self.offsets[p]
        return log_j


class OracleNull:
    def __init__(self, parameters):
        self.parameters = list(parameters)

    def update(self, x):
        pass

    def reparameterise(self, x, x_prime, log_j):
        for p in self.parameters:
            x_prime[p + "_prime"] = x[p]
        return log_j

    def inverse_reparameterise(self, x, x_prime, log_j):
        for p in self.parameters:
            x[p] = x_prime[p + "_prime"]
        return log_j


class OracleCombined:
    """Ordered list of reparameterisations (``combined.py``), dict-of-arrays interface."""

    def __init__(self, reparameterisations):
        self.items = list(reparameterisations)

    def update(self, x):
        for r in self.items:
            r.update(x)

    def rescale(self, x, n):
        x_prime, log_j = {}, np.zeros(n)
        for r in self.items:  # combined.py:168-171
            log_j = r.reparameterise(x, x_prime, log_j)
        return x_prime, log_j

    def inverse_rescale(self, x_prime, n):
        x, log_j = {}, np.zeros(n)
        for r in reversed(self.items):  # combined.py:188-191
            log_j = r.inverse_reparameterise(x, x_prime, log_j)
        return x, log_j
