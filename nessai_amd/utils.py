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


def eval_mode(module):
    """``module.eval()`` without the walk over the module tree when it is in eval mode already - a flow is ~90
    modules, ``Module.eval`` ~0.35 ms of host time per call, and the importance sampler asked for it on every
    ``draw`` / ``update_log_q``.  A ``ModuleList`` is checked flow by flow (a flow appended after the list was
    switched is still in training mode); within a flow the mode is only ever changed from the top."""
    import torch

    if isinstance(module, torch.nn.ModuleList):
        for m in module:
            eval_mode(m)
        module.training = False
    elif module.training:
        module.eval()
    return module
