"""Small numerical helpers used by the proposal."""

import numpy as np
from scipy import stats


def compute_radius(n, q=0.95):
    """Radius containing probability ``q`` of an n-D unit Gaussian.

    Inverse CDF of the chi distribution (``src/nessai/utils/sampling.py:15-33``).
    """
    return stats.chi.ppf(q, n)


def rescale_log_jacobian(scales):
    """``sum_i log|sigma_i|`` accumulated parameter by parameter in float64,

    the order in which ``ScaleAndShift.inverse_reparameterise`` adds the terms to
    every row (``src/nessai/reparameterisations/rescale.py:276-286``).
    """
    total = np.float64(0.0)
    for s in scales:
        total = total + np.log(np.abs(np.float64(s)))
    return float(total)
