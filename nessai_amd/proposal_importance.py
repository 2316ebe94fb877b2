"""``ImportanceFlowProposal``: the meta-proposal of the importance nested sampler (config 4).

Counterpart of the reference's ``nessai.proposal.importance.ImportanceFlowProposal``
(``src/nessai/proposal/importance.py:35``).  Samples live in the unit hypercube; the flows are
trained in the logit space (``to_prime`` / ``from_prime``, ``importance.py:203-257``,
``utils/rescaling.py:290-336``); the meta-proposal is the weighted mixture
``log Q = logsumexp(log q_i + log w_i)`` over the prior (proposal ``-1``, ``log q = 0``) and every
flow added so far (``compute_log_Q``, ``importance.py:382-441``).

Where the reference loops over host arrays, one batch of ``draw`` here is three launches on
device tensors and one copy of the rows that survive:

1. ``nsa_importance_draw`` - the fused flow kernel with the importance epilogue: Philox latents
   indexed by the global row -> inverse flow ``i`` -> ``x = sigmoid(x')`` in float64, the
   flow-space point ``x'`` for step 2, ``log|J|`` of the logit map on the clipped ``x`` and the
   reference's finiteness / unit-hypercube checks (``importance.py:470-483``) as a NaN mark;
2. ``nsa_flow_run_many`` - ``log_prob`` of EVERY stored flow on those points in one grouped launch
   (``flowmodel/importance.py:114-129`` is a Python loop over flows);
3. ``nsa_importance_mix`` - ``log_q_all`` and ``log Q = logsumexp(log_q_all, b = weights)`` in
   float64.

The prior terms (``logP``, ``logU``) stay with the user's model on the host, on the surviving rows
only.  With ``shard=True`` under ``torch.distributed`` the rows of every batch - and of
``update_log_q`` / ``compute_log_Q``, whose cost grows with every level - are split over the ranks
and all-gathered; the Philox streams are indexed by the global row, so the
sharded run draws exactly the points of the single-rank run.
"""

import logging
import os

import numpy as np
import torch
from scipy.special import logsumexp

from . import lib
from .flowmodel_importance import ImportanceFlowModel
from .flows.base import log_prob_many
from .utils import eval_mode
from .livepoint import (NON_SAMPLING_DEFAULTS, NON_SAMPLING_DTYPES, NON_SAMPLING_PARAMETERS, concatenate_points,
                        live_points_to_array, parameter_block, take_points)

logger = logging.getLogger(__name__)

EPS = 1e-8  # src/nessai/config.py:159
INS_PARAMETERS = ["logQ", "logW", "logU"]  # tests/conftest.py:111-117 (``ins_parameters``)


def logit(x, eps=None):
    """(x', log|J|) with inputs clamped to [eps, 1 - eps] (``utils/rescaling.py:290-313``)."""
    if eps:
        x = np.clip(x, eps, 1 - eps)
    a, b = np.log(x), np.log1p(-x)  # each evaluated once; the same two roundings as the reference's lines
    return a - b, -a - b


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
    block = parameter_block(out, list(names))
    if block is not None:
        block.fill(np.nan)
    else:
        for nm in names:
            out[nm] = np.nan
    for nm, v in zip(NON_SAMPLING_PARAMETERS, NON_SAMPLING_DEFAULTS):
        out[nm] = v
    for nm in INS_PARAMETERS:
        out[nm] = np.nan
    return out


def uniform_box_of(model, dtype):
    """``(lower, upper, total)`` (float64 arrays and ``-sum log(width)``) when every prior method ``draw`` calls on the
    model is the framework's own uniform-box implementation - ``UniformBoxModel.log_prior`` through
    ``Model.batch_evaluate_log_prior``, ``Model.from_unit_hypercube`` / ``in_bounds`` / ``in_unit_hypercube`` /
    ``batch_evaluate_log_prior_unit_hypercube`` - and the parameters are one float64 block of the point dtype; ``None``
    otherwise (a subclass that overrides any of them is called on the host, as the reference calls it).
    ``NESSAI_AMD_DRAW_POINTS=0`` always answers ``None``."""
    from .model import Model, UniformBoxModel

    t = type(model)
    own = (getattr(model, "uniform_prior", False) and isinstance(model, UniformBoxModel)
           and t.log_prior is UniformBoxModel.log_prior and t.batch_evaluate_log_prior is Model.batch_evaluate_log_prior
           and t.from_unit_hypercube is Model.from_unit_hypercube and t.in_unit_hypercube is Model.in_unit_hypercube
           and t.in_bounds is Model.in_bounds
           and t.batch_evaluate_log_prior_unit_hypercube is Model.batch_evaluate_log_prior_unit_hypercube
           and parameter_block(np.empty(1, dtype=dtype), list(model.names)) is not None
           and os.environ.get("NESSAI_AMD_DRAW_POINTS", "1") != "0")
    if not own:
        return None
    total = np.float64(0.0)  # (UniformBoxModel.log_prior's own sequence of float64 subtractions)
    for bounds in model.bounds.values():
        total -= np.log(bounds[1] - bounds[0])
    return (np.asarray(model.lower_bounds, dtype=np.float64).copy(), np.asarray(model.upper_bounds, dtype=np.float64).copy(),
            float(total))


class ImportanceFlowProposal:
    def __init__(self, model, output, rng=None, flow_config=None, training_config=None,
                 reparameterisation="logit", weighted_kl=True, reset_flow=True, clip=False,
                 plot_training=False, device_seed=None, shard=False):
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
        self.shard = bool(shard)
        self.device_seed = int(device_seed) if device_seed is not None else int(self.rng.integers(0, 2**63 - 1))
        self._draw_counter = 0  # rows drawn so far: Philox row offset

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
    def dims(self):
        return len(self.model.names)

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
        block = parameter_block(out, list(self.model.names))
        if block is not None:
            block[:] = x
        else:
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
        rank, world = self._rank_world()
        if world > 1:
            self._replicate_newest_flow()

    # -- device plumbing --------------------------------------------------------------------

    def _rank_world(self):
        if not self.shard:
            return 0, 1
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return 0, 1
        return dist.get_rank(), dist.get_world_size()

    def _replicate_newest_flow(self):
        """Rank 0's fit becomes the newest flow of every rank (training does not shard and the
        fused trainer is not bit-reproducible - see ``FlowProposal._replicate_flow``)."""
        import torch.distributed as dist

        model = self.flow.model
        with torch.no_grad():
            trainer = model.trainer()
            dist.broadcast(trainer.flat, src=0)
            for _, buf in model.named_buffers():
                if buf.numel() and buf.data_ptr() not in trainer.flat_range():
                    dist.broadcast(buf, src=0)
        state = [self.rng.bit_generator.state]
        dist.broadcast_object_list(state, src=0)
        self.rng.bit_generator.state = state[0]
        model._image_key = None

    def _check_ready(self):
        if any(np.isnan(w) for w in self.weights.values()):
            raise RuntimeError("Some weights are not set!")
        if any(flow.training for flow in self.flow.models):
            raise RuntimeError("One or more flows are in training mode!")

    def _stream(self):
        return lib.c_void_p(torch.cuda.current_stream(self.flow.device).cuda_stream)

    def _mix(self, log_prob_all, log_j):
        """(log_q_all ``(n, P)``, log_Q ``(n,)``) float64 device tensors (``nsa_importance_mix``)."""
        dev = self.flow.device
        n = log_j.shape[0]
        n_flows = self.flow.n_models
        weights = torch.tensor(self.weights_array, dtype=torch.float64, device=dev)
        if weights.numel() != n_flows + 1:
            raise RuntimeError(f"{weights.numel()} proposal weights for {n_flows} flows and the prior")
        log_q_all = torch.empty((n, n_flows + 1), dtype=torch.float64, device=dev)
        log_Q = torch.empty(n, dtype=torch.float64, device=dev)
        rc = lib.load().nsa_importance_mix(
            n, n_flows, lib.ptr(log_prob_all) if n_flows else None, lib.ptr(log_j), lib.ptr(weights),
            lib.ptr(log_q_all), lib.ptr(log_Q), self._stream(),
        )
        lib.check(rc, "nsa_importance_mix")
        return log_q_all, log_Q

    def _log_prob_rows(self, flows, xp):
        """``(len(flows), n)`` float32 log-densities of the rows of ``xp`` (device, float32); with
        ``shard`` every rank evaluates a slice of the rows and the slices are all-gathered."""
        rank, world = self._rank_world()
        n = xp.shape[0]
        if world == 1 or n < world:
            return log_prob_many(flows, xp)
        import torch.distributed as dist

        per = (n + world - 1) // world
        lo, hi = min(rank * per, n), min((rank + 1) * per, n)
        mine = torch.zeros((len(flows), per), dtype=torch.float32, device=xp.device)
        if hi > lo:
            mine[:, : hi - lo] = log_prob_many(flows, xp[lo:hi].contiguous())
        gathered = torch.empty((world * len(flows), per), dtype=torch.float32, device=xp.device)  # rank-major
        dist.all_gather_into_tensor(gathered, mine)
        gathered = gathered.reshape(world, len(flows), per)
        return gathered.permute(1, 0, 2).reshape(len(flows), world * per)[:, :n].contiguous()

    def _draw_batch(self, flow_number, n_rows, z_in=None, want_points=False):
        """One batch of ``draw`` on the device.  Returns host arrays ``(x (m, D), log_q_all (m, P),
        log_Q (m,))`` of the rows that pass the reference's first filter, in draw order - or, with
        ``want_points`` on one rank for a model with the framework's own uniform box prior, ``(points, log_q_all,
        None)``: the structured array of the batch with every field already filled (``_finish_batch_points``)."""
        if self.reparameterisation != "logit":
            raise NotImplementedError("the gfx950 draw epilogue implements the logit reparameterisation only")
        L = lib.load()
        flow = self.flow.models[flow_number]
        eval_mode(flow)
        dev, D = self.flow.device, self.dims
        rank, world = self._rank_world()
        per = (n_rows + world - 1) // world if world > 1 else n_rows
        lo = min(rank * per, n_rows)
        n_loc = min((rank + 1) * per, n_rows) - lo if world > 1 else n_rows
        # same capacity on every rank: the all-gather pads to the largest count
        packed = torch.empty((max(per, 1), D + self.n_proposals + 1), dtype=torch.float64, device=dev)
        n_keep = torch.zeros(1, dtype=torch.int64, device=dev)
        if n_loc > 0:
            header, image = flow.device_image()
            x64 = torch.empty((n_loc, D), dtype=torch.float64, device=dev)
            xp = torch.empty((n_loc, D), dtype=torch.float32, device=dev)
            log_j = torch.empty(n_loc, dtype=torch.float64, device=dev)
            z_dev = None
            if z_in is not None:
                z_dev = torch.from_numpy(np.ascontiguousarray(z_in[lo : lo + n_loc])).to(torch.float32).to(dev)
            for _ in range(2):
                flow._status.zero_()
                rc = L.nsa_importance_draw(
                    header, lib.ptr(image), n_loc, lib.ptr(z_dev), self.device_seed, self._draw_counter + lo, EPS,
                    lib.ptr(x64), lib.ptr(xp), lib.ptr(log_j), lib.ptr(flow._status), flow.waves_per_group,
                    flow.max_groups, self._stream(),
                )
                lib.check(rc, "nsa_importance_draw")
                # split-f16 draw kernel outside the f16 range: repeat this rank's rows in float32
                # (one rank: the flag travels with the batch's single read-back - _finish_batch_one_rank - and a
                # batch that left the range is drawn again as a whole, below)
                if world == 1:
                    box = self._uniform_box()
                    done = (self._finish_batch_points(x64, xp, log_j, flow, box) if box is not None and want_points
                            else self._finish_batch_one_rank(x64, xp, log_j, flow))
                    if done is not None:
                        self._draw_counter += n_rows
                        return done
                elif flow.effective_precision != "f16x3" or int(flow._status[1].item()) == 0:
                    break
                flow.note_f16_overflow()
                header, image = flow.device_image()
            if world == 1:
                raise RuntimeError("the draw kernel left the f16 range in float32 mode")  # (cannot happen)
            if self.clip:
                x64.clamp_(0.0, 1.0)
            log_prob_all = log_prob_many(self.flow.models, xp)  # this rank's rows under every flow
            log_q_all, log_Q = self._mix(log_prob_all, log_j)
            keep = torch.nonzero(~torch.isnan(log_j)).reshape(-1)  # draw order
            m = keep.numel()
            if m and bool(torch.isnan(log_Q[keep]).any()):  # compute_log_Q's check (importance.py:436-441)
                raise ValueError("There is a NaN in log g!")
            packed[:m, :D] = x64[keep]
            packed[:m, D:-1] = log_q_all[keep]
            packed[:m, -1] = log_Q[keep]
            n_keep[0] = m
        self._draw_counter += n_rows
        if world > 1:
            from .proposal import _gather_accepted

            _, rows = _gather_accepted(packed, n_keep, packed.shape[0], packed.shape[1])
        else:
            rows = packed[: int(n_keep.item())]
        rows = rows.cpu().numpy()
        return rows[:, :D], rows[:, D:-1], rows[:, -1]

    def _weights_device(self):
        """The proposal weights on the device, uploaded when they change (not per draw)."""
        key = tuple(float(w) for w in self.weights_array)
        cached = getattr(self, "_weights_dev", None)
        # (by device TYPE: `self.flow.device` may read "cuda" where the tensor says "cuda:0" - see _finish_batch_one_rank)
        if cached is None or cached[0] != key or cached[1].device.type != torch.device(self.flow.device).type:
            cached = (key, torch.tensor(key, dtype=torch.float64, device=self.flow.device))
            self._weights_dev = cached
        return cached[1]

    def _uniform_box(self):
        """``(lower, upper, total)`` on the device when the model's prior methods are the framework's own uniform box
        (``Model.from_unit_hypercube`` / ``in_unit_hypercube`` / ``batch_evaluate_log_prior_unit_hypercube`` and
        ``UniformBoxModel.log_prior`` not overridden): the draw's prior fields are then the device's business; a user's
        own methods are called on the host as the reference calls them.  ``total`` is ``UniformBoxModel.log_prior``'s
        own ``-sum log(width)`` (same float64 sequence)."""
        cached = getattr(self, "_box_dev", None)
        if cached is not None:
            return cached[0]
        host = uniform_box_of(self.model, self.dtype)
        box = None
        if host is not None:
            dev = self.flow.device
            box = (torch.tensor(host[0], device=dev), torch.tensor(host[1], device=dev), host[2])
        self._box_dev = (box,)
        return box

    def _finish_batch_points(self, x64, xp, log_j, flow, box):
        """``_finish_batch_one_rank`` with the batch's records in the live-point wire format itself
        (``nsa_importance_mix_points``): the prior fields of a uniform box model, ``logQ`` / ``logW`` / ``logU`` and the
        defaults of the other fields are written per row on the device and the host VIEWS the pinned read-back as the
        structured array ``draw`` returns - no numpy pass over the parameters (1.5 of the draw's 2.7 ms before)."""
        D, P = self.dims, self.n_proposals
        n = x64.shape[0]
        log_prob_all = log_prob_many(self.flow.models, xp)
        if self.flow.n_models + 1 != P:
            raise RuntimeError(f"{P} proposal weights for {self.flow.n_models} flows and the prior")
        words = 2 * D + 11
        assert 4 * words == self.dtype.itemsize, (words, self.dtype)
        bufs = getattr(self, "_draw_point_bufs", None)
        if bufs is None or bufs[0].shape != (n, words) or bufs[1].shape != (n, P + 1) or bufs[0].device != x64.device:
            bufs = (torch.empty((n, words), dtype=torch.int32, device=x64.device),
                    torch.empty((n, P + 1), dtype=torch.float64, device=x64.device),
                    torch.empty((n, words), dtype=torch.int32).pin_memory(),
                    torch.empty((n, P + 1), dtype=torch.float64).pin_memory(),
                    torch.zeros(4, dtype=torch.int32).pin_memory())
            self._draw_point_bufs = bufs
        points_dev, aux_dev, points_host, aux_host, status_host = bufs
        lower, upper, total = box
        rc = lib.load().nsa_importance_mix_points(
            n, self.flow.n_models, D, lib.ptr(log_prob_all), lib.ptr(log_j), lib.ptr(self._weights_device()),
            lib.ptr(x64), 1 if self.clip else 0, lib.ptr(lower), lib.ptr(upper), total, lib.ptr(points_dev),
            lib.ptr(aux_dev), self._stream(),
        )
        lib.check(rc, "nsa_importance_mix_points")
        points_host.copy_(points_dev, non_blocking=True)
        aux_host.copy_(aux_dev, non_blocking=True)
        status_host.copy_(flow._status, non_blocking=True)
        torch.cuda.current_stream(x64.device).synchronize()
        if flow.effective_precision == "f16x3" and int(status_host[1]) != 0:
            return None  # a value beyond the f16 range: the caller draws the batch again in float32
        raw, aux = points_host.numpy(), aux_host.numpy()
        keep = ~np.isnan(aux[:, -1])  # the reference's first filter, draw order
        # (out of the pinned buffers, which are the next batch's - as 32-bit words: numpy copies a packed structured
        # array element by element, 1.3 ms for these 2.8 MB against 0.2 for the words)
        if keep.all():
            raw, log_q_all = raw.copy(), aux[:, :P].copy()
        else:
            raw, log_q_all = raw[keep], aux[keep, :P]
        points = raw.view(self.dtype).reshape(raw.shape[0])
        if np.isnan(points["logQ"]).any():  # compute_log_Q's check (importance.py:436-441)
            raise ValueError("There is a NaN in log g!")
        return points, log_q_all, None

    def _finish_batch_one_rank(self, x64, xp, log_j, flow):
        """The device epilogue of a draw batch on one rank: every flow's density of the rows, the meta-proposal,
        ONE record per row (``nsa_importance_mix_packed``) and ONE read-back through pinned memory (``None``: the
        split-f16 kernel reported a value beyond its range - the flag travels with the same copy); the reference's
        first filter (rows the draw kernel NaN-marked) and ``compute_log_Q``'s NaN check run on the host copy.
        (Five synchronisations and a dozen small torch launches before: 3.0 ms around a 0.45 ms kernel.)"""
        dev, D, P = self.flow.device, self.dims, self.n_proposals
        n = x64.shape[0]
        log_prob_all = log_prob_many(self.flow.models, xp)
        if self.flow.n_models + 1 != P:
            raise RuntimeError(f"{P} proposal weights for {self.flow.n_models} flows and the prior")
        width = D + P + 2
        bufs = getattr(self, "_draw_bufs", None)
        # (keyed on the shape and the device INDEX: `self.flow.device` may read "cuda" where the tensors say "cuda:0" -
        # compared as devices the two differ, and the pinned buffers were allocated anew on every draw, 1.6 of 2.9 ms)
        if bufs is None or bufs[0].shape != (n, width) or bufs[0].device != x64.device:
            bufs = (torch.empty((n, width), dtype=torch.float64, device=x64.device),
                    torch.empty((n, width), dtype=torch.float64).pin_memory(),
                    torch.zeros(4, dtype=torch.int32).pin_memory())
            self._draw_bufs = bufs
        packed, host, status_host = bufs
        rc = lib.load().nsa_importance_mix_packed(
            n, self.flow.n_models, D, lib.ptr(log_prob_all), lib.ptr(log_j), lib.ptr(self._weights_device()),
            lib.ptr(x64), 1 if self.clip else 0, lib.ptr(packed), self._stream(),
        )
        lib.check(rc, "nsa_importance_mix_packed")
        host.copy_(packed, non_blocking=True)
        status_host.copy_(flow._status, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        if flow.effective_precision == "f16x3" and int(status_host[1]) != 0:
            return None  # a value beyond the f16 range: the caller draws the batch again in float32
        rows = host.numpy()
        keep = ~np.isnan(rows[:, -1])  # draw order
        if not keep.all():
            rows = rows[keep]
        if np.isnan(rows[:, -2]).any():  # compute_log_Q's check (importance.py:436-441)
            raise ValueError("There is a NaN in log g!")
        return rows[:, :D].copy(), rows[:, D : D + P].copy(), rows[:, -2].copy()

    # -- meta-proposal -------------------------------------------------------------------

    def compute_log_Q(self, x_prime, log_j=None):
        """(log Q, log q of every proposal) (``importance.py:382-441``)."""
        self._check_ready()
        n_flows = self.flow.n_models
        if self.n_proposals > 1 and log_j is None:
            raise RuntimeError("Must specify log_j! Meta-proposal includes flows")
        dev = self.flow.device
        n = x_prime.shape[0]
        if n == 0:
            return np.empty(0), np.zeros((0, self.n_proposals))
        lj = torch.from_numpy(np.ascontiguousarray(log_j if log_j is not None else np.zeros(n), dtype=np.float64)).to(dev)
        log_prob_all = None
        if n_flows >= 1:
            xp = torch.from_numpy(np.ascontiguousarray(x_prime)).to(torch.float32).to(dev)
            log_prob_all = self._log_prob_rows(self.flow.models, xp)
        log_q_all, log_Q = self._mix(log_prob_all, lj)
        log_Q = log_Q.cpu().numpy()
        if np.isnan(log_Q).any():
            raise ValueError("There is a NaN in log g!")
        return log_Q, log_q_all.cpu().numpy()

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
            return lambda x: self._log_prob_ith(x, it)
        raise ValueError

    def _log_prob_ith(self, x_prime, it):
        flow = self.flow.models[it]
        eval_mode(flow)
        xp = torch.from_numpy(np.ascontiguousarray(x_prime)).to(torch.float32).to(self.flow.device)
        return self._log_prob_rows([flow], xp)[0].cpu().numpy().astype(np.float64)

    def update_log_q(self, samples, log_q):
        """Append the newest proposal's column (``importance.py:528-543``); this is evaluated on
        every stored sample at every level, so its rows are sharded as well."""
        if log_q.shape[1] == self.n_proposals:
            raise ValueError("log_q array already contains current proposal")
        if self.reparameterisation == "logit" and self.level_count >= 0:
            column = self._newest_column_device(samples)
        else:
            x, log_j = self.rescale(samples)
            column = self.get_proposal_log_prob(self.level_count)(x) + log_j
        return np.concatenate([log_q, column[:, np.newaxis]], axis=1)

    def _newest_column_device(self, samples):
        """``log q_newest(x)`` of stored samples with the logit map on the device
        (``nsa_importance_to_prime``): the host only flattens the records and uploads them.  On the
        host the map was 60 % of ``update_log_q`` (two float64 logarithms per element in numpy)."""
        x = live_points_to_array(samples, self.model.names)
        dev = self.flow.device
        n, d = x.shape
        with torch.inference_mode(False), torch.no_grad():
            xd = torch.from_numpy(x).to(dev)
            xp = torch.empty((n, d), dtype=torch.float32, device=dev)
            log_j = torch.empty(n, dtype=torch.float64, device=dev)
            rc = lib.load().nsa_importance_to_prime(n, d, lib.ptr(xd), EPS, lib.ptr(xp), lib.ptr(log_j), self._stream())
            lib.check(rc, "nsa_importance_to_prime")
            flow = self.flow.models[self.level_count]
            eval_mode(flow)
            log_prob = self._log_prob_rows([flow], xp)[0]
            return (log_prob.to(torch.float64) + log_j).cpu().numpy()

    def draw(self, n, flow_number=None, injected=None):
        """``n`` points from one flow with their meta-proposal weights (``importance.py:443-526``).

        Batches of ``int(1.01 n)`` rows until ``n`` survive both filters; the first ``n`` are
        returned.  ``injected`` (parity tests): iterator of latent arrays used instead of the
        device generator."""
        self._check_ready()
        source = self.level_count if flow_number is None else flow_number
        batch = int(1.01 * n)
        names = self.model.names
        kept_points, kept_log_q, have = [], [], 0
        while have < n and batch > 0:
            z_in = next(injected) if injected is not None else None
            x, log_q_all, log_Q = self._draw_batch(source, batch, z_in=z_in, want_points=True)
            if not len(x):
                continue
            if log_Q is None:  # the batch came back as the structured array itself (_finish_batch_points)
                points = x
            else:
                points = empty_ins_array(len(x), names)
                block = parameter_block(points, list(names))
                if block is not None:
                    block[:] = x
                else:
                    for col, name in enumerate(names):
                        points[name] = x[:, col]
                points["logQ"] = log_Q
                points["logP"] = self.model.batch_evaluate_log_prior(self.model.from_unit_hypercube(points))
                points["logU"] = self.model.batch_evaluate_log_prior_unit_hypercube(points)
                points["logW"] = points["logU"] - points["logQ"]
            # second filter (importance.py:497-503): a usable prior, a weight that is not +inf, and
            # not every proposal density NaN / +inf
            usable = np.isfinite(points["logP"]) & ~np.isposinf(points["logW"])
            usable &= ~np.isnan(log_q_all).all(axis=1) & ~np.isposinf(log_q_all).all(axis=1)
            if not usable.any():
                continue
            all_usable = bool(usable.all())
            kept_points.append(points if all_usable else take_points(points, usable))
            kept_log_q.append(log_q_all if all_usable else log_q_all[usable])
            have += int(usable.sum())
        if not kept_points:
            return np.zeros(0, dtype=self.dtype), np.empty([0, self.n_proposals])
        log_q_kept = kept_log_q[0] if len(kept_log_q) == 1 else np.concatenate(kept_log_q, axis=0)
        return concatenate_points(kept_points)[:n], log_q_kept[:n]

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
