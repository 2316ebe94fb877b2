"""CPU restatement of the importance nested sampler's meta-proposal (config 4).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): imported by ``tests/`` and never by
``nessai_amd``.  Restates, in float64 numpy exactly as the reference's glue is written,
``ImportanceFlowProposal.to_prime / from_prime / rescale / inverse_rescale / compute_log_Q / draw /
update_log_q`` (``src/nessai/proposal/importance.py:203-277, 382-543``) and
``ImportanceFlowModel.log_prob_all / log_prob_ith`` (``src/nessai/flowmodel/importance.py:100-129``)
on top of the oracle flows (``oracle/flows.py``).  The flow's own draw
(``sample_ith`` = ``flow.sample(N)``: standard-normal latents through the inverse transform,
``flows/base.py:262-275``) takes its latents from the caller so that the device path can be fed the
same ones.

Pinned against the reference itself: ``tests/golden/make_importance_goldens.py`` drives the
reference's ``draw`` / ``compute_log_Q`` / ``update_log_q`` on a stand-in flow with fixed
outputs; ``tests/test_golden.py::test_oracle_importance_draw_matches_reference`` replays them here.
"""

import numpy as np
import torch
from scipy.special import logsumexp

EPS = 1e-8  # src/nessai/config.py:159
NON_SAMPLING = [("logP", "f8"), ("logL", "f8"), ("it", "i4")]  # config.py:25-35
INS = [("logQ", "f8"), ("logW", "f8"), ("logU", "f8")]


def logit(x, eps=None):  # utils/rescaling.py:290-313
    if eps:
        x = np.clip(x, eps, 1 - eps)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.log(x) - np.log1p(-x), -np.log(x) - np.log1p(-x)


def sigmoid(x):  # utils/rescaling.py:316-336
    with np.errstate(over="ignore", divide="ignore", invalid="ignore"):
        x = np.divide(1, 1 + np.exp(-x))
        return x, np.log(x) + np.log1p(-x)


def live_points(x, names):
    """``numpy_array_to_live_points`` with the importance sampler's extra fields (livepoint.py:74-155)."""
    out = np.zeros(x.shape[0], dtype=[(n, "f8") for n in names] + NON_SAMPLING + INS)
    for i, n in enumerate(names):
        out[n] = x[:, i]
    out["logP"] = np.nan
    out["logL"] = np.nan
    for n, _ in INS:
        out[n] = np.nan
    return out


class OracleImportanceProposal:
    def __init__(self, model, flows, weights, clip=False):
        self.model = model
        self.flows = list(flows)  # oracle flows (torch modules on the CPU), eval mode
        self.weights_array = np.asarray(weights, dtype=float)  # prior first
        self.clip = clip
        self.names = list(model.names)

    n_proposals = property(lambda self: len(self.weights_array))

    def to_prime(self, x):  # importance.py:203-229
        x_prime, log_j = logit(np.atleast_2d(x), eps=EPS)
        return x_prime, log_j.sum(axis=1)

    def from_prime(self, x_prime):  # importance.py:231-257
        x, log_j = sigmoid(np.atleast_2d(x_prime))
        return x, log_j.sum(axis=1)

    def rescale(self, x):  # importance.py:259-266
        return self.to_prime(np.stack([x[n] for n in self.names], axis=1))

    def inverse_rescale(self, x_prime):  # importance.py:268-277
        x, log_j = self.from_prime(x_prime)
        if self.clip:
            x = np.clip(x, 0.0, 1.0)
        return live_points(x, self.names), log_j

    def log_prob_all(self, x_prime):  # flowmodel/importance.py:114-129
        x = torch.from_numpy(x_prime).type(torch.get_default_dtype())
        out = torch.empty(x.shape[0], len(self.flows))
        with torch.no_grad():
            for i, m in enumerate(self.flows):
                out[:, i] = m.log_prob(x)
        return out.numpy().astype(np.float64)

    def log_prob_ith(self, x_prime, i):  # flowmodel/importance.py:100-112
        with torch.no_grad():
            lp = self.flows[i].log_prob(torch.from_numpy(x_prime).type(torch.get_default_dtype()))
        return lp.numpy().astype(np.float64)

    def sample_ith(self, i, z):  # flowmodel/importance.py:131-142 with the latents given
        with torch.no_grad():
            x, _ = self.flows[i].inverse(torch.from_numpy(np.asarray(z)).type(torch.get_default_dtype()))
        return x.numpy().astype(np.float64)

    def compute_log_Q(self, x_prime, log_j):  # importance.py:382-441
        log_q_all = np.zeros([x_prime.shape[0], self.n_proposals])
        if len(self.flows) >= 1:
            log_q_all[:, 1 : (len(self.flows) + 1)] = self.log_prob_all(x_prime) + log_j[:, np.newaxis]
        log_Q = logsumexp(log_q_all, b=self.weights_array, axis=1)
        if np.isnan(log_Q).any():
            raise ValueError("There is a NaN in log g!")
        return log_Q, log_q_all

    def draw(self, n, flow_number, latents=None, x_primes=None):  # importance.py:443-526
        """``latents``: iterator of ``(n_draw, D)`` standard-normal arrays, one per batch of the
        loop (or ``x_primes``: the flow's samples themselves)."""
        n_draw = int(1.01 * n)
        dtype = live_points(np.zeros((0, len(self.names))), self.names).dtype
        samples = np.zeros(0, dtype=dtype)
        log_q_samples = np.empty([0, self.n_proposals])
        n_accepted = 0
        while n_accepted < n and n_draw > 0:
            x_prime = next(x_primes) if x_primes is not None else self.sample_ith(flow_number, next(latents))
            x, log_j_inv = self.inverse_rescale(x_prime)
            x_check, log_j = self.rescale(x)
            acc = (
                self.model.in_unit_hypercube(x)
                & np.isfinite(x_check).all(axis=1)
                & np.isfinite(x_prime).all(axis=1)
                & np.isfinite(log_j)
                & np.isfinite(log_j_inv)
            )
            if not np.any(acc):
                continue
            x, x_prime, log_j = x[acc], x_prime[acc], log_j[acc]
            x["logQ"], log_q_all = self.compute_log_Q(x_prime, log_j=log_j)
            x["logP"] = self.model.batch_evaluate_log_prior(x, unit_hypercube=True)
            x["logU"] = self.model.batch_evaluate_log_prior_unit_hypercube(x)
            x["logW"] = x["logU"] - x["logQ"]
            accept = (
                np.isfinite(x["logP"])
                & ~np.isposinf(x["logW"])
                & ~np.isnan(log_q_all).all(axis=1)
                & ~np.isposinf(log_q_all).all(axis=1)
            )
            if not np.any(accept):
                continue
            x, log_q_all = x[accept], log_q_all[accept]
            samples = np.concatenate([samples, x])
            log_q_samples = np.concatenate([log_q_samples, log_q_all], axis=0)
            n_accepted += x.size
        return samples[:n], log_q_samples[:n]

    def update_log_q(self, samples, log_q, level):  # importance.py:528-543
        if log_q.shape[1] == self.n_proposals:
            raise ValueError("log_q array already contains current proposal")
        x, log_j = self.rescale(samples)
        log_q_current = self.log_prob_ith(x, level)
        return np.concatenate([log_q, log_q_current[:, np.newaxis] + log_j[:, np.newaxis]], axis=1)
