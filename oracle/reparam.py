"""CPU restatement of the reference's per-parameter reparameterisations.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): used by ``tests/`` as the checker of the
device epilogue's reparameterisation table, never by the product path.

Pinned against the reference's own classes: ``tests/golden/make_reparam_goldens.py`` imports
``nessai.reparameterisations`` (it has no glasflow dependency) in this container and stores
inputs and outputs in ``tests/golden/reparam_goldens.npz``; ``tests/test_reparam.py`` checks this
restatement against them bit for bit.

Follows ``src/nessai/reparameterisations``: ``ScaleAndShift`` (``rescale.py:119-312``),
``RescaleToBounds`` incl. boundary inversion (``rescale.py:321-741``; edge detection
``utils/rescaling.py:106-222``, ``utils/hist.py``), ``NullReparameterisation``
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


def auto_bins(x, max_bins=50):  # utils/hist.py:46-76
    x = np.asarray(x)
    iqr = np.subtract(*np.percentile(x, [75, 25]))
    fd_bw = 2.0 * iqr * x.size ** (-1.0 / 3.0)  # utils/hist.py:9-25
    sturges_bw = np.ptp(x) / (np.log2(x.size) + 1.0)  # utils/hist.py:28-43
    bw = min(fd_bw, sturges_bw) if fd_bw else sturges_bw
    n_bins = int(np.ceil(np.ptp(x)) / bw) if bw else 1
    return min(n_bins, max_bins)


def detect_edge(x, cutoff=0.5, allow_none=False, percent=0.1, test=None):  # utils/rescaling.py:106-193
    """Both bounds allowed, ``allow_both`` off, ``nbins="auto"``: the cases the registry entries reach."""
    if test is not None:
        return test
    nbins = auto_bins(x)
    hist, bins = np.histogram(x, bins=nbins, density=True)
    n = max(int(len(bins) * percent), 1)
    bounds_fraction = np.array([np.sum(hist[:n]), np.sum(hist[-n:])]) * (bins[1] - bins[0])
    max_idx = np.argmax(hist)
    max_density = hist[max_idx] * (bins[1] - bins[0])
    if max_idx < n:
        return "lower"
    if max_idx >= (len(bins) - n):
        return "upper"
    if not np.any(bounds_fraction > cutoff * max_density) and allow_none:
        return False
    return ["lower", "upper"][int(np.argmax(bounds_fraction))]


class OracleBoundaryInversion:
    """``RescaleToBounds(boundary_inversion=True, inversion_type=..., detect_edges=...)`` on dicts of arrays
    (``rescale.py:425-463, 541-612, 673-745``); ``rng`` draws the rows a split inversion flips."""

    def __init__(self, parameters, prior_bounds, inversion_type="split", detect_edges=True, rng=None):
        self.parameters = list(parameters)
        self.prior_bounds = {p: np.asarray(prior_bounds[p], dtype=float) for p in parameters}
        self.inversion_type, self.rng = inversion_type, rng
        # utils/rescaling.py:196-222
        self.edge_kwargs = dict(cutoff=0.5, allow_none=True) if detect_edges else dict(cutoff=0.0, allow_none=False)
        self._update = True if detect_edges else True  # update_bounds defaults to True (rescale.py:451)
        self.offsets = {p: 0.0 for p in parameters}
        self.bounds = {p: b - 0.0 for p, b in self.prior_bounds.items()}
        self.edges = {p: None for p in parameters}

    def update(self, x):  # rescale.py:720-745
        self.bounds = {p: [np.min(x[p]), np.max(x[p])] for p in self.parameters}
        self.edges = {p: None for p in self.parameters}

    def reparameterise(self, x, x_prime, log_j, compute_radius=False, test=None):  # rescale.py:614-645, 541-583
        """Returns (x, x_prime, log_j): duplication replaces every array of both dicts."""
        for p in self.parameters:
            pp = p + "_prime"
            v = x[p]
            if self.edges[p] is None:
                self.edges[p] = detect_edge(v, test=test, **self.edge_kwargs)
            lo, hi = self.bounds[p]
            if self.edges[p]:
                v = (v - self.offsets[p] - lo) / (hi - lo)
                log_j = log_j + (-np.log(hi - lo))
                if self.edges[p] == "upper":
                    v = 1 - v
                if self.inversion_type == "duplicate" or compute_radius:
                    x_prime = {k: np.concatenate([a, a]) for k, a in x_prime.items()}
                    x_prime[pp] = np.concatenate([v, -v])
                    x = {k: np.concatenate([a, a]) for k, a in x.items()}
                    log_j = np.concatenate([log_j, log_j])
                else:
                    n = v.size
                    inv = self.rng.choice(n, n // 2, replace=False)
                    v = v.copy()
                    v[inv] *= -1
                    x_prime[pp] = v
            else:
                x_prime[pp] = (2.0 * (v - self.offsets[p] - lo) / (hi - lo)) - 1
                log_j = log_j + (np.log(2) - np.log(hi - lo))
        return x, x_prime, log_j

    def inverse_reparameterise(self, x, x_prime, log_j):  # rescale.py:647-671, 585-612
        for p in reversed(self.parameters):
            v = x_prime[p + "_prime"]
            lo, hi = self.bounds[p]
            if self.edges[p]:
                v = v.copy()
                inv = v < 0.0
                v[inv] = -v[inv]
                if self.edges[p] == "upper":
                    v = 1 - v
                v = (hi - lo) * v + lo
                log_j += np.log(hi - lo)
            else:
                v = (hi - lo) * ((v + 1) / 2.0) + lo
                log_j += np.log(hi - lo) - np.log(2)
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

    def update(self, x):  # combined.py:219-230: update, then apply, group by group
        if not any(isinstance(r, OracleBoundaryInversion) for r in self.items):
            for r in self.items:
                r.update(x)
            return
        names = x.dtype.names if getattr(x, "dtype", None) is not None and x.dtype.names else list(x)
        cur, x_prime = {n: np.asarray(x[n]) for n in names}, {}
        log_j = np.zeros(len(next(iter(cur.values()))))
        for r in self.items:
            r.update(cur)
            if isinstance(r, OracleBoundaryInversion):
                cur, x_prime, log_j = r.reparameterise(cur, x_prime, log_j)
            else:
                log_j = r.reparameterise(cur, x_prime, log_j)

    def rescale(self, x, n, compute_radius=False, test=None):
        x_prime, log_j = {}, np.zeros(n)
        x = dict(x)
        for r in self.items:  # combined.py:168-171
            if isinstance(r, OracleBoundaryInversion):  # may double every array
                x, x_prime, log_j = r.reparameterise(x, x_prime, log_j, compute_radius=compute_radius, test=test)
            else:
                log_j = r.reparameterise(x, x_prime, log_j)
        return x_prime, log_j

    def inverse_rescale(self, x_prime, n):
        x, log_j = {}, np.zeros(n)
        for r in reversed(self.items):  # combined.py:188-191
            log_j = r.inverse_reparameterise(x, x_prime, log_j)
        return x, log_j
