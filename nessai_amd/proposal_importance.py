"""``ImportanceFlowProposal``: the meta-proposal of the importance nested sampler (config 4).

Counterpart of the reference's ``nessai.proposal.importance.ImportanceFlowProposal``
(``src/nessai/proposal/importance.py:35``).  Samples live in the unit hypercube; the flows are
trained in the logit space (``to_prime`` / ``from_prime``, ``importance.py:203-257``,
``utils/rescaling.py:290-336``); the meta-proposal is the weighted mixture
``log Q = logsumexp(log q_i + log w_i)`` over the prior (proposal ``-1``, ``log q = 0``) and
every flow added so far (``compute_log_Q``, ``importance.py:382-441``).  Flow evaluation and
training run on the GPU through ``ImportanceFlowModel``; the glue is float64 numpy as in the
reference.
"""

import logging
import os

import numpy as np
from scipy.special import logsumexp

from .flowmodel_importance import ImportanceFlowModel
from .livepoint import NON_SAMPLING_DEFAULTS, NON_SAMPLING_DTYPES, NON_SAMPLING_PARAMETERS, live_points_to_array

logger = logging.getLogger(__name__)

EPS = 1e-8  # src/nessai/config.py:159
INS_PARAMETERS = ["logQ", "logW", "logU"]  # tests/conftest.py:111-117 (``ins_parameters``)


def logit(x, eps=None):
    """(x', log|J|) with inputs clamped to [eps, 1 - eps] (``utils/rescaling.py:290-313``)."""
    if eps:
        x = np.clip(x, eps, 1 - eps)
    log_j = -np.log(x) - np.log1p(-x)
    return np.log(x) - np.log1p(-x), log_j


def sigmoid(x):
    """(x, log|J|) (``utils/rescaling.py:316-336``)."""
    with np.errstate(over="ignore", divide="ignore"):
        x = np.divide(1, 1 + np.exp(-x))
        log_j = np.log(x) + np.log1p(-x)
    return x, log_j


def ins_dtype(names):
    """Live-point dtype with the importance sampler's extra fields logQ, logW, logU."""
    dtype = [(n, "f8") for n in names]
    dtype += list(zip(NON_SAMPLING_PARAMETERS, NON_SAMPLING_DTYPES))
    dtype += [(p, "f8") for p in INS_PARAMETERS]
    return np.dtype(dtype)


def empty_ins_array(n, names):
    out = np.empty(n, dtype=ins_dtype(names))
    for nm in names:
        out[nm] = np.nan
    for nm, v in zip(NON_SAMPLING_PARAMETERS, NON_SAMPLING_DEFAULTS):
        out[nm] = v
    for nm in INS_PARAMETERS:
        out[nm] = np.nan
    return out


class ImportanceFlowProposal:
    def __init__(self, model, output, rng=None, flow_config=None, training_config=None,
                 reparameterisation="logit", weighted_kl=True, reset_flow=True, clip=False,
                 plot_training=False):
        self.level_count = -1
        self.initialised = False
        self.model = model
        self.rng = rng if rng is not None else np.random.default_rng()
        self.output = output
        config = dict(flow_config or {})
        config["n_inputs"] = len(model.names)
        self.flow_config = config
        self.training_config = training_config
        self.reset_flow = int(reset_flow)
        self.reparameterisation = reparameterisation
        self.weighted_kl = weighted_kl
        self.clip = clip
        self._weights = {-1: 1.0}
        self.dtype = ins_dtype(model.names)
        self.flow = None

    # -- meta-proposal weights -------------------------------------------------

    @property
    def weights(self):
        return self._weights

    @property
    def weights_array(self):
        return np.fromiter(self._weights.values(), dtype=float)

    @property
    def n_proposals(self):
        return len(self._weights)

    @property
    def _reset_flow(self):
        return bool(self.reset_flow) and not (self.level_count % self.reset_flow)

    def update_proposal_weights(self, weights):
        """(``importance.py:279-290``)."""
        self._weights.update(weights)
        w_sum = np.sum(np.fromiter(self._weights.values(), float))
        if not np.isclose(w_sum, 1.0):
            raise RuntimeError(f"Weights must sum to 1! Actual value: {w_sum}")

    def initialise(self):
        if self.initialised:
            return
        self.flow = ImportanceFlowModel(
            flow_config=self.flow_config, training_config=self.training_config, output=self.output,
            rng=self.rng,
        )
        self.flow.initialise()
        self.initialised = True

    # -- unit hypercube <-> x' ------------------------------------------------------

    def to_prime(self, x):
        x = np.atleast_2d(x)
        if self.reparameterisation == "logit":
            x_prime, log_j = logit(x, eps=EPS)
            log_j = log_j.sum(axis=1)
        elif self.reparameterisation is None:
            x_prime, log_j = x.copy(), np.zeros(x.shape[0])
        else:
            raise ValueError(f"Unknown reparameterisation: '{self.reparameterisation}'")
        return x_prime, log_j

    def from_prime(self, x_prime):
        x_prime = np.atleast_2d(x_prime)
        if self.reparameterisation == "logit":
            x, log_j = sigmoid(x_prime)
            log_j = log_j.sum(axis=1)
        elif self.reparameterisation is None:
            x, log_j = x_prime.copy(), np.zeros(x_prime.shape[0])
        else:
            raise ValueError(f"Unknown reparameterisation: '{self.reparameterisation}'")
        return x, log_j

    def rescale(self, x):
        return self.to_prime(live_points_to_array(x, self.model.names))

    def inverse_rescale(self, x_prime):
        x, log_j = self.from_prime(x_prime)
        if self.clip:
            x = np.clip(x, 0.0, 1.0)
        out = empty_ins_array(x.shape[0], self.model.names)
        for i, n in enumerate(self.model.names):
            out[n] = x[:, i]
        return out, log_j

    # -- training ----------------------------------------------------------------------

    def train(self, samples, plot=False, output=None, weights=None, **kwargs):
        """Add a level: train a new flow on the samples (``importance.py:292-380``)."""
        self.level_count += 1
        self._weights[self.level_count] = np.nan
        output = self.output if output is None else output
        level_output = os.path.join(output, f"level_{self.level_count}", "")
        os.makedirs(level_output, exist_ok=True)
        training_data = samples.copy()
        x_prime, _ = self.rescale(training_data)
        if self.weighted_kl or weights is not None:
            if weights is not None:
                weights = weights / np.sum(weights)
            else:
                log_weights = training_data["logW"].copy()
                log_weights -= logsumexp(log_weights)
                weights = np.exp(log_weights)
            if np.isnan(weights).any():
                raise ValueError("Weights contain NaN(s)")
            if not np.isfinite(weights).all():
                raise ValueError("Weights contain Inf(s)")
        else:
            weights = None
        self.flow.add_new_flow(reset=self._reset_flow)
        self.flow.train(x_prime, weights=weights, output=level_output, plot=False, **kwargs)

    # -- meta-proposal -------------------------------------------------------------------

    def compute_log_Q(self, x_prime, log_j=None):
        """(log Q, log q of every proposal) (``importance.py:382-441``)."""
        if any(np.isnan(w) for w in self.weights.values()):
            raise RuntimeError("Some weights are not set!")
        log_q_all = np.zeros([x_prime.shape[0], self.n_proposals])
        n_flows = self.flow.n_models
        if self.n_proposals > 1 and log_j is None:
            raise RuntimeError("Must specify log_j! Meta-proposal includes flows")
        if any(flow.training for flow in self.flow.models):
            raise RuntimeError("One or more flows are in training mode!")
        if n_flows >= 1:
            log_q_all[:, 1 : (n_flows + 1)] = self.flow.log_prob_all(x_prime) + log_j[:, np.newaxis]
        log_Q = logsumexp(log_q_all, b=self.weights_array, axis=1)
        if np.isnan(log_Q).any():
            raise ValueError("There is a NaN in log g!")
        return log_Q, log_q_all

    def compute_meta_proposal_from_log_q(self, log_q):
        return logsumexp(log_q, b=self.weights_array, axis=1)

    def compute_meta_proposal_samples(self, samples):
        if self.level_count not in self.weights or np.isnan(self.weights[self.level_count]):
            raise RuntimeError(f"Weight(s) missing or not set. Current weights: {self.weights}.")
        x, log_j = self.rescale(samples)
        return self.compute_log_Q(x, log_j=log_j)

    def get_proposal_log_prob(self, it):
        if it == -1:
            return lambda x: np.zeros(x.shape[0])
        elif it < len(self.flow.models):
            return lambda x: self.flow.log_prob_ith(x, it)
        raise ValueError

    def update_log_q(self, samples, log_q):
        """Append the newest proposal's column (``importance.py:528-543``)."""
        if log_q.shape[1] == self.n_proposals:
            raise ValueError("log_q array already contains current proposal")
        x, log_j = self.rescale(samples)
        log_q_current = self.get_proposal_log_prob(self.level_count)(x)
        return np.concatenate([log_q, log_q_current[:, np.newaxis] + log_j[:, np.newaxis]], axis=1)

    def draw(self, n, flow_number=None):
        """Draw n points from one flow with their meta-proposal weights (``importance.py:443-526``)."""
        if flow_number is None:
            flow_number = self.level_count
        n_draw = int(1.01 * n)
        samples = np.zeros(0, dtype=self.dtype)
        log_q_samples = np.empty([0, self.n_proposals])
        n_accepted = 0
        while n_accepted < n and n_draw > 0:
            x_prime = self.flow.sample_ith(i=flow_number, N=n_draw)
            x, log_j_inv = self.inverse_rescale(x_prime)
            with np.errstate(divide="ignore", invalid="ignore"):
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
            x["logP"] = self.model.batch_evaluate_log_prior(self.model.from_unit_hypercube(x))
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

    def draw_from_prior(self, n):
        """(``importance.py:632-641``)."""
        samples = empty_ins_array(n, self.model.names)
        u = self.rng.random((n, len(self.model.names)))
        for i, nm in enumerate(self.model.names):
            samples[nm] = u[:, i]
        samples["logU"] = self.model.batch_evaluate_log_prior_unit_hypercube(samples)
        prime_samples, log_j = self.rescale(samples)
        log_Q, log_q = self.compute_log_Q(prime_samples, log_j=log_j)
        samples["logQ"] = log_Q
        samples["logW"] = samples["logU"] - log_Q
        return samples, log_q
