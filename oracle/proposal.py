"""CPU restatement of the reference's FlowModel / FlowProposal glue (numpy f64).

Test infrastructure only (see ``oracle/__init__.py``).  Restates, for the
default configuration of the benchmark configs (z-score reparameterisation,
constant-volume latent-radius truncation, bounded box priors):

* ``FlowModel`` numpy<->torch boundary and training loop —
  ``src/nessai/flowmodel/base.py:195-236, 238-352, 365-523, 530-696, 774-948``
* ``FlowProposal.backward_pass`` / ``populate`` —
  ``src/nessai/proposal/flowproposal/flowproposal.py:345-389, 391-534``
* ``BaseFlowProposal.latent_log_prob`` / ``compute_weights`` /
  ``check_prior_bounds`` — ``proposal/flowproposal/base.py:401-414, 1069-1098,
  939-959``
* ``LatentRadiusTruncation`` — ``proposal/flowproposal/truncation.py:292-365``
  and ``compute_radius`` — ``utils/sampling.py:15-33``
* z-score ``ScaleAndShift`` — ``reparameterisations/rescale.py:233-312``
* ``Model.in_bounds`` — ``model.py:497-518``

``tests/test_golden.py`` checks these against outputs captured from the
reference's own classes (fixtures G1-G5 of SURVEY.md §8c).
"""

import copy
import datetime

import numpy as np
import torch
from scipy import stats
from torch.nn.utils import clip_grad_norm_

from .flows import configure_model

NON_SAMPLING = ["logP", "logL", "it"]  # config.py:25-35
NON_SAMPLING_DTYPES = ["f8", "f8", "i4"]
NON_SAMPLING_DEFAULTS = [np.nan, np.nan, 0]


def compute_radius(n, q=0.95):
    """Radius enclosing probability ``q`` of an n-D unit Gaussian."""
    return stats.chi.ppf(q, n)  # utils/sampling.py:33


def live_point_dtype(names):
    return [(n, "f8") for n in names] + list(zip(NON_SAMPLING, NON_SAMPLING_DTYPES))


def empty_structured_array(n, names):
    """Structured array with default non-sampling fields (livepoint.py:105-155)."""
    out = np.empty(n, dtype=live_point_dtype(names))
    for name in names:
        out[name] = np.nan
    for name, default in zip(NON_SAMPLING, NON_SAMPLING_DEFAULTS):
        out[name] = default
    return out


class BoxModel:
    """Minimal stand-in for ``nessai.model.Model`` with a uniform box prior.

    ``log_prior`` follows ``examples/rosenbrock.py:26-31`` (uniform, one
    ``-log(width)`` term per dimension subtracted in turn); ``in_bounds``
    follows ``src/nessai/model.py:512-518``.
    """

    def __init__(self, dims, lower=-5.0, upper=5.0):
        self.names = [f"x_{d}" for d in range(dims)]
        self.bounds = {n: [float(lower), float(upper)] for n in self.names}
        self.dims = dims

    def in_bounds(self, x):
        return ~np.any(
            [
                (x[n] < self.bounds[n][0]) | (x[n] > self.bounds[n][1])
                for n in self.names
            ],
            axis=0,
        )

    def log_prior(self, x):
        with np.errstate(divide="ignore"):
            log_p = np.log(self.in_bounds(x), dtype="float")
        for bounds in self.bounds.values():
            log_p -= np.log(bounds[1] - bounds[0])
        return log_p


class RosenbrockModel(BoxModel):
    """``examples/rosenbrock.py:16-45`` (bounds [-5, 5]^n)."""

    def log_likelihood(self, x):
        a = np.stack([x[n] for n in self.names], axis=-1)
        return -np.sum(
            100.0 * (a[..., 1:] - a[..., :-1] ** 2.0) ** 2.0
            + (1.0 - a[..., :-1]) ** 2.0,
            axis=-1,
        )


class GaussianModel(BoxModel):
    """Unit Gaussian likelihood in a box (style of ``examples/2d_gaussian.py``)."""

    def __init__(self, dims, lower=-10.0, upper=10.0):
        super().__init__(dims, lower, upper)

    def log_likelihood(self, x):
        a = np.stack([x[n] for n in self.names], axis=-1)
        return -0.5 * np.sum(a**2, axis=-1) - 0.5 * self.dims * np.log(2 * np.pi)


# --------------------------------------------------------------------------
# FlowModel
# --------------------------------------------------------------------------

DEFAULT_TRAINING_CONFIG = dict(  # flowmodel/config.py:31-46
    lr=1e-3,
    annealing=False,
    clip_grad_norm=5.0,
    batch_size=1000,
    val_size=0.1,
    max_epochs=500,
    patience=20,
    optimiser="adamw",
)


def check_batch_size(x, batch_size, min_fraction=0.1):
    """flowmodel/base.py:195-236."""
    if batch_size == 1:
        raise ValueError("Cannot use a batch size of 1!")
    min_batch_size = int(min_fraction * batch_size)
    final_batch_size = len(x) % batch_size
    if final_batch_size and (final_batch_size < min_batch_size):
        while True:
            batch_size -= 1
            final_batch_size = len(x) % batch_size
            if batch_size < 2:
                raise RuntimeError("Could not find a valid batch size")
            elif (final_batch_size == 0) or (final_batch_size >= min_batch_size):
                break
            elif (batch_size <= min_batch_size) and final_batch_size > 1:
                break
    return batch_size


class OracleFlowModel:
    """numpy-f64-in / numpy-f64-out wrapper around an oracle flow."""

    def __init__(self, flow_config, training_config=None, rng=None):
        self.flow_config = dict(flow_config)
        self.training_config = dict(DEFAULT_TRAINING_CONFIG)
        self.training_config.update(training_config or {})
        self.rng = rng if rng is not None else np.random.default_rng()
        self.model = configure_model(self.flow_config)
        self.model.device = torch.device("cpu")
        self._optimiser = torch.optim.AdamW(
            self.model.parameters(), lr=self.training_config["lr"]
        )
        self._batch_size = None

    def numpy_array_to_tensor(self, array):  # base.py:774-780
        return torch.from_numpy(array).type(torch.get_default_dtype())

    def forward_and_log_prob(self, x):  # base.py:782-811
        x = self.numpy_array_to_tensor(x)
        self.model.eval()
        with torch.inference_mode():
            z, log_prob = self.model.forward_and_log_prob(x)
        return (
            z.numpy().astype(np.float64),
            log_prob.numpy().astype(np.float64),
        )

    #: rows per call of ``model.inverse`` (None = the whole array at once, as the reference does);
    #: chunking only bounds the size of the eager temporaries, the rows are independent
    chunk_rows = None

    def inverse(self, z):  # base.py:813-840
        z = self.numpy_array_to_tensor(z)
        self.model.eval()
        with torch.inference_mode():
            if self.chunk_rows and len(z) > self.chunk_rows:
                parts = [self.model.inverse(c) for c in torch.split(z, int(self.chunk_rows))]
                x = torch.cat([p[0] for p in parts])
                log_j = torch.cat([p[1] for p in parts])
            else:
                x, log_j = self.model.inverse(z)
        return x.numpy().astype(np.float64), log_j.numpy().astype(np.float64)

    def log_prob(self, x):  # base.py:842-864
        x = self.numpy_array_to_tensor(x)
        self.model.eval()
        with torch.inference_mode():
            log_prob = self.model.log_prob(x)
        return log_prob.numpy().astype(np.float64)

    def sample_latent_distribution(self, n=1):  # base.py:889-904
        with torch.inference_mode():
            z = self.model.sample_latent_distribution(n)
        return z.numpy().astype(np.float64)

    def base_distribution_log_prob(self, z):  # proposal base.py:411-414
        with torch.inference_mode():
            log_p = self.model.base_distribution_log_prob(self.numpy_array_to_tensor(z))
        return log_p.numpy().astype(np.float64)

    # -- training ----------------------------------------------------------

    def prep_data(self, samples, val_size, batch_size):  # base.py:238-352
        if not np.isfinite(samples).all():
            raise ValueError("Cannot train with non-finite samples!")
        idx = self.rng.permutation(samples.shape[0])
        samples = samples[idx]
        n = int((1 - (val_size or 0)) * samples.shape[0])
        x_train, x_val = samples[:n], samples[n:]
        batch_size = check_batch_size(x_train, batch_size)
        self._batch_size = batch_size
        dtype = torch.get_default_dtype()
        return (
            torch.from_numpy(x_train).type(dtype),
            torch.from_numpy(x_val).type(dtype),
            batch_size,
        )

    def _train(self, train_data):  # base.py:365-452
        model = self.model
        model.train()
        train_loss = 0
        p = torch.randperm(train_data.shape[0])
        batches = train_data[p, :].split(self._batch_size)
        n = 0
        for x in batches:
            for param in model.parameters():
                param.grad = None
            loss = -model.log_prob(x).mean()
            train_loss += loss.item()
            loss.backward()
            if self.training_config["clip_grad_norm"]:
                clip_grad_norm_(
                    model.parameters(), self.training_config["clip_grad_norm"]
                )
            self._optimiser.step()
            n += 1
        return train_loss / n

    def _validate(self, val_data):  # base.py:454-523
        if not len(val_data):
            return np.nan
        self.model.eval()
        with torch.inference_mode():
            return (-self.model.log_prob(val_data).mean()).item()

    def train(self, samples, max_epochs=None, patience=None, val_size=None):
        """Maximum-likelihood fit; follows ``base.py:530-696``."""
        if not np.isfinite(samples).all():
            raise ValueError("Training data is not finite")
        if val_size is None:
            val_size = self.training_config["val_size"]
        validate = val_size != 0.0
        train_data, val_data, _ = self.prep_data(
            samples, val_size=val_size, batch_size=self.training_config["batch_size"]
        )
        if max_epochs is None:
            max_epochs = self.training_config["max_epochs"]
        if patience is None:
            patience = self.training_config["patience"]
        best_epoch = 0
        best_val_loss = np.inf
        best_model = copy.deepcopy(self.model.state_dict())
        history = dict(loss=[], val_loss=[])
        for epoch in range(1, max_epochs + 1):
            loss = self._train(train_data)
            val_loss = self._validate(val_data)
            history["loss"].append(loss)
            history["val_loss"].append(val_loss)
            if validate and (val_loss < best_val_loss):
                best_epoch = epoch
                best_val_loss = val_loss
                best_model = copy.deepcopy(self.model.state_dict())
            if validate and (epoch - best_epoch > patience):
                break
        self.model.train()
        self.model.eval()
        if validate:
            self.model.load_state_dict(best_model)
        self.model.eval()
        return history


# --------------------------------------------------------------------------
# FlowProposal
# --------------------------------------------------------------------------


class OracleFlowProposal:
    """Draw → truncate → inverse flow → rescale → bounds → reject, on the CPU.

    z-score rescaling only (``x' = (x - shift) / scale`` per parameter with
    the live-point mean/std, ``rescale.py:293-304``); constant-volume latent
    radius by default (``truncation.py:15-21``).
    """

    def __init__(self, model, flow, rng=None, poolsize=2000, drawsize=None,
                 volume_fraction=0.95, accumulate_weights=False, radius_mode="constant_volume",
                 fixed_radius=False, expansion_fraction=4.0, min_radius=False, max_radius=50.0,
                 truncate_log_q=False, enforce_likelihood_threshold=False):
        self.model = model
        self.flow = flow
        self.rng = rng if rng is not None else np.random.default_rng()
        self.poolsize = poolsize
        self.drawsize = drawsize if drawsize is not None else poolsize
        self.accumulate_weights = accumulate_weights
        self.names = list(model.names)
        self.prime_names = [n + "_prime" for n in self.names]
        self.scale = {n: 1.0 for n in self.names}
        self.shift = {n: 0.0 for n in self.names}
        # LatentRadiusTruncation.configure (truncation.py:292-308): outside constant-volume mode the
        # ball is inflated by fuzz = (1 + expansion_fraction) ** (1 / dims)
        self.radius_mode = radius_mode
        self.min_radius, self.max_radius = min_radius, max_radius
        self.fuzz = (1 + expansion_fraction) ** (1 / len(self.names)) if expansion_fraction else 1.0
        self.fixed_radius = fixed_radius
        if radius_mode == "constant_volume":
            self.fixed_radius = compute_radius(len(self.names), volume_fraction)
            self.fuzz = 1.0  # truncation.py:305-308
        self.truncate_log_q = truncate_log_q                      # MinLogQTruncation, truncation.py:368-392
        self.enforce_likelihood_threshold = enforce_likelihood_threshold  # :398-430
        self.training_data = None
        self.population_time = datetime.timedelta()
        self.population_acceptance = None
        self.x = None
        self.samples = None
        self.indices = []
        self.n_proposed = 0
        self.n_accepted = 0
        self.injected_latents = None  # iterator of (N, D) arrays, for parity tests
        self.reparam = None  # optional oracle.reparam.OracleCombined replacing the z-score
        self.latent_temperature = None  # base.py:393-414

    # -- reparameterisation (z-score) ---------------------------------------

    def update_rescaling(self, live_points):  # rescale.py:293-304
        if self.reparam is not None:
            self.reparam.update(live_points)
            return
        for n in self.names:
            self.scale[n] = np.std(live_points[n])
            self.shift[n] = np.mean(live_points[n])

    def rescale(self, x, compute_radius=False):  # rescale.py:233-261, base.py:716-753
        if self.reparam is not None:
            cols, log_J = self.reparam.rescale({n: x[n] for n in self.names}, x.size, compute_radius=compute_radius)
            x_prime = empty_structured_array(len(log_J), self.prime_names)  # boundary inversion may double the rows
            for pp in self.prime_names:
                x_prime[pp] = cols[pp]
            for p in NON_SAMPLING:
                x_prime[p] = np.tile(x[p], len(log_J) // max(x.size, 1))
            return x_prime, log_J
        x_prime = empty_structured_array(x.size, self.prime_names)
        log_J = np.zeros(x_prime.size)
        for p, pp in zip(self.names, self.prime_names):
            x_prime[pp] = (x[p] - self.shift[p]) / self.scale[p]
            log_J -= np.log(np.abs(self.scale[p]))
        for p in NON_SAMPLING:
            x_prime[p] = x[p]
        return x_prime, log_J

    def inverse_rescale(self, x_prime):  # rescale.py:263-291, base.py:755-784
        x = empty_structured_array(x_prime.size, self.names)
        if self.reparam is not None:
            cols, log_J = self.reparam.inverse_rescale(x_prime, x_prime.size)
            for p in self.names:
                x[p] = cols[p]
            for p in NON_SAMPLING:
                x[p] = x_prime[p]
            return x, log_J
        log_J = np.zeros(x.size)
        for p, pp in zip(self.names, self.prime_names):
            value = x_prime[pp]
            value = (value * self.scale[p]) + self.shift[p]
            log_J += np.log(np.abs(self.scale[p]))
            x[p] = value
        for p in NON_SAMPLING:
            x[p] = x_prime[p]
        return x, log_J

    # -- flow passes ----------------------------------------------------------

    def compute_population_radius(self, worst_point, r=None):  # truncation.py:323-356
        if r is not None:
            return float(r)
        if self.radius_mode != "adaptive":
            return float(self.fixed_radius)
        worst_z = self.forward_pass(worst_point)[0]
        radius = float(np.sqrt(np.sum(worst_z**2.0, axis=-1)).max())
        if self.max_radius and radius > self.max_radius:
            radius = self.max_radius
        if self.min_radius and radius < self.min_radius:
            radius = self.min_radius
        return float(radius)

    def train(self, live_points, **kwargs):  # base.py:870-925
        self.training_data = live_points.copy()
        self.update_rescaling(live_points)
        x_prime, _ = self.rescale(live_points)
        arr = np.stack([x_prime[pp] for pp in self.prime_names], axis=-1)
        return self.flow.train(arr, **kwargs)

    def forward_pass(self, x, compute_radius=False):  # base.py:961-994
        x_prime, log_J = self.rescale(x, compute_radius=compute_radius)
        arr = np.stack([x_prime[pp] for pp in self.prime_names], axis=-1)
        z, log_prob = self.flow.forward_and_log_prob(arr)
        return z, log_prob + log_J

    def sample_latent_distribution(self, n):  # base.py:393-399
        if self.injected_latents is not None:
            return np.asarray(next(self.injected_latents), dtype=np.float64)
        z = self.flow.sample_latent_distribution(n)
        if self.latent_temperature in (None, 1.0):
            return z
        return np.sqrt(self.latent_temperature) * z

    def latent_log_prob(self, z):  # base.py:401-414
        z = np.asarray(z)
        if self.latent_temperature in (None, 1.0):
            return self.flow.base_distribution_log_prob(z)
        scale = np.sqrt(float(self.latent_temperature))
        return self.flow.base_distribution_log_prob(z / scale) - z.shape[-1] * np.log(scale)

    def apply_latent(self, z, threshold):  # truncation.py:358-365
        radius = np.sqrt(np.sum(z**2.0, axis=-1))
        keep = radius <= threshold
        return z[keep]

    def backward_pass(self, z):  # flowproposal.py:345-389
        try:
            x, log_j = self.flow.inverse(z)
            log_prob = self.latent_log_prob(z) - log_j
        except AssertionError:
            return np.array([]), np.array([]), np.array([])
        valid = np.isfinite(log_prob)
        x, log_prob, z = x[valid], log_prob[valid], z[valid]
        x_array = np.asarray(x, dtype="f8")
        x = empty_structured_array(x_array.shape[0], self.prime_names)
        for i, parameter in enumerate(self.prime_names):
            x[parameter] = x_array[:, i]
        x, log_J = self.inverse_rescale(x)
        log_prob -= log_J
        flags = self.model.in_bounds(x)  # base.py:939-959
        return x[flags], log_prob[flags], z[flags]

    def compute_weights(self, x, log_q):  # base.py:1069-1098
        return self.model.log_prior(x) - log_q

    def populate(self, n_samples=None, r=None, max_samples=1_000_000, worst_point=None):
        """flowproposal.py:391-534 (both ``accumulate_weights`` modes, the three truncation rules)."""
        st = datetime.datetime.now()
        if n_samples is None:
            n_samples = self.poolsize
        radius = self.compute_population_radius(worst_point, r)
        self.r = radius
        threshold = self.fuzz * radius
        min_log_q = None
        if self.truncate_log_q:  # MinLogQTruncation.prepare
            min_log_q = float(self.forward_pass(self.training_data)[1].min())
        log_l_threshold = None
        if self.enforce_likelihood_threshold:  # LikelihoodThresholdTruncation.prepare
            log_l_threshold = float(np.asarray(worst_point["logL"], dtype=float).reshape(-1)[0])
            if not np.isfinite(log_l_threshold):
                log_l_threshold = -np.inf
        self.indices = []
        if self.accumulate_weights:
            samples = empty_structured_array(0, self.names)
        else:
            samples = empty_structured_array(n_samples, self.names)
        log_n = np.log(n_samples)
        log_n_expected = -np.inf
        n_proposed = 0
        log_weights = np.empty(0)
        log_constant = -np.inf
        n_accepted = 0
        accept = None

        from scipy.special import logsumexp

        while n_accepted < n_samples:
            z = self.sample_latent_distribution(self.drawsize)
            n_proposed += z.shape[0]
            z = self.apply_latent(z, threshold)
            if not len(z):
                if n_proposed > max_samples:
                    break
                continue
            x, log_q, z = self.backward_pass(z)
            if min_log_q is not None and len(x):  # apply_after_backward
                keep = log_q > min_log_q
                x, log_q, z = x[keep], log_q[keep], z[keep]
            if not len(x):
                if n_proposed > max_samples:
                    break
                continue
            if log_l_threshold is not None:  # apply_after_likelihood (flowproposal.py:455-466)
                x["logL"] = self.model.log_likelihood(x)
                keep = x["logL"] > log_l_threshold
                x, log_q, z = x[keep], log_q[keep], z[keep]
                if not len(x):
                    if n_proposed > max_samples:
                        break
                    continue
            log_w = self.compute_weights(x, log_q)
            if self.accumulate_weights:
                samples = np.concatenate([samples, x])
                log_weights = np.concatenate([log_weights, log_w])
                log_constant = max(np.nanmax(log_w), log_constant)
                log_n_expected = logsumexp(log_weights - log_constant)
                if log_n_expected >= log_n:
                    log_u = np.log(self.rng.random(len(log_weights)))
                    accept = (log_weights - log_constant) > log_u
                    n_accepted = np.sum(accept)
                if n_proposed > max_samples:
                    break
            else:
                log_w -= log_w.max()
                log_u = np.log(self.rng.random(len(log_w)))
                accept = log_w > log_u
                n_accept_batch = accept.sum()
                m = min(n_samples - n_accepted, n_accept_batch)
                samples[n_accepted : n_accepted + m] = x[accept][:m]
                n_accepted += n_accept_batch
                if n_proposed > max_samples:
                    break

        if self.accumulate_weights:
            if accept is None or len(accept) != len(samples):
                log_u = np.log(self.rng.random(len(log_weights)))
                accept = (log_weights - log_constant) > log_u
            n_accepted = np.sum(accept)
            self.x = samples[accept][:n_samples]
        else:
            self.x = samples[: min(n_accepted, n_samples)]

        self.samples = self.x.copy()  # convert_to_samples, base.py:1100-1128
        self.samples["logP"] = self.model.log_prior(self.samples)
        self.population_time += datetime.datetime.now() - st
        self.indices = self.rng.permutation(self.samples.size).tolist()
        self.population_acceptance = n_accepted / n_proposed
        self.n_proposed = n_proposed
        self.n_accepted = int(n_accepted)
        return self.samples
