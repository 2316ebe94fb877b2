"""Training loop of ``FlowModel.train`` on the gfx950 training kernels.

Follows the reference's schedule (``src/nessai/flowmodel/base.py:530-696`` and
``_train`` / ``_validate``, ``base.py:365-523``): per epoch the training set is
re-shuffled with ``torch.randperm`` and split into minibatches; every minibatch
is one ``nsa_train_step`` call (forward, ``-mean(log_prob)``, backward,
``clip_grad_norm_``, AdamW — all enqueued on the stream by the C side, no host
round trip); the validation loss is one ``nsa_train_eval`` call; the best
weights (lowest validation loss) are kept on the device and restored at the
end; training stops ``patience`` epochs after the best one.  The host reads two
scalars per epoch.

Trainable scope: permutations, LU-linear layers, coupling layers (affine,
additive, rational-quadratic spline) with a residual or plain-MLP conditioner,
batch norm between layers and act norm (with its data-dependent
initialisation) - i.e. every RealNVP / NeuralSplineFlow configuration of the
reference, conditional inputs included (layer-wise kernels).
"""

import contextlib
import ctypes
import math
import os

import numpy as np
import torch

from . import lib, pack
from .flows import modules as M


class Trainer:
    """Flat parameter / gradient / Adam-moment buffers + the layer plan of ONE flow.

    A flow has at most one live ``Trainer`` (``HipFlow.trainer()``): it rebinds the storage of
    every ``Parameter`` and batch-norm statistic to views of its flat buffer, which both the
    fused optimiser step (``nsa_train_step``) and the autograd path (``nessai_amd.ops``:
    ``nsa_train_forward`` / ``nsa_train_backward``) work on.
    """

    OPTIMISERS = {"adamw": 0, "adam": 1, "sgd": 2}  # nsa_train_step's `optimiser` argument

    def __init__(self, flow, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, optimiser="adamw"):
        self.flow = flow
        self.configure_optimiser(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, optimiser=optimiser,
                                 reset=False)
        self.step_count = 0
        device = next(flow.parameters()).device
        if device.type != "cuda":
            raise RuntimeError("nessai_amd training needs the flow on a ROCm device (no CPU fallback)")
        self.device = device
        desc = M.describe(flow)
        is_mlp = desc["net"] == "mlp"
        leaves = M.flatten_transforms(flow._transform)
        D = desc["D"]
        cur = list(range(D))
        tables, records = [], []
        params = list(flow.parameters())
        if not all(p.dtype == torch.float32 for p in params):
            raise TypeError("float32 parameters expected")
        # one flat buffer [trainable parameters | running statistics]; every Parameter and
        # batch-norm buffer becomes a view of it
        stats = [b for name, b in flow.named_buffers() if name.endswith(("running_mean", "running_var"))]
        total = sum(p.numel() for p in params)
        flat = torch.empty(total + sum(b.numel() for b in stats), dtype=torch.float32, device=device)
        offsets, pos = {}, 0
        with torch.no_grad():
            for p in params + stats:
                flat[pos : pos + p.numel()].copy_(p.detach().reshape(-1))
                p.data = flat[pos : pos + p.numel()].view(p.shape)
                offsets[id(p)] = pos
                pos += p.numel()
        self.flat = flat
        self.params = flat[:total]
        self.grads = torch.zeros(total, dtype=torch.float32, device=device)
        self.adam_m = torch.zeros_like(self.grads)
        self.adam_v = torch.zeros_like(self.grads)
        self.best = torch.empty_like(flat)
        n_res = None

        def f32_bits(value):
            return int(np.array([value], dtype=np.float32).view(np.uint32)[0])

        tmap = np.arange(total, dtype=np.int64)  # transposition map of the weight matrices
        dropout = None  # dropout_probability of the residual blocks (nflows ResidualBlock; one value per flow)
        inner_bn = False  # batch_norm_within_layers: two BatchNorm1d per residual block
        inner_bn_eps = 1e-3
        self._inner_bns = []  # their modules: num_batches_tracked ticks with every training-mode pass
        self._actnorms = []  # (record index, module) of every act-norm layer

        for t in leaves:
            if hasattr(t, "_permutation"):
                perm = t._permutation.detach().cpu().numpy()
                cur = [cur[int(perm[i])] for i in range(D)]
            elif hasattr(t, "transform_net"):
                net = t.transform_net
                if is_mlp:  # input, hidden x (n - 1), output (src/nessai/flows/nets.py:55-126)
                    layers = net.layers()
                    depth = len(layers) - 2
                else:
                    p_drop = float(getattr(net, "dropout_probability", 0.0) or 0.0)
                    if dropout is not None and p_drop != dropout:
                        raise ValueError("all coupling layers must share one dropout_probability")
                    dropout = p_drop
                    if bool(getattr(net, "use_batch_norm", False)) != bool(inner_bn) and records and any(
                            r[0] == 1 for r in records):
                        raise ValueError("all coupling layers must share batch_norm_within_layers")
                    inner_bn = bool(getattr(net, "use_batch_norm", False))
                    layers = [net.initial_layer]
                    for block in net.blocks:
                        layers.extend(block.linear_layers)
                    layers.append(net.final_layer)
                    depth = len(net.blocks)
                n_res = depth if n_res is None else n_res
                if depth != n_res:
                    raise ValueError("all coupling layers must share one conditioner depth")
                id_cols = [cur[int(f)] for f in t.identity_features.tolist()]
                tr_cols = [cur[int(f)] for f in t.transform_features.tolist()]
                rec = [1, len(id_cols), len(tr_cols), len(tables), len(tables) + len(id_cols)]
                tables.extend(id_cols)
                tables.extend(tr_cols)
                for layer in layers:
                    rec.append(offsets[id(layer.weight)])
                    rec.append(offsets[id(layer.bias)])
                    o_, k_ = layer.weight.shape  # W[o][k] -> W^T[k][o] at the same offset
                    base = offsets[id(layer.weight)]
                    idx = np.arange(o_ * k_)
                    tmap[base : base + o_ * k_] = base + (idx % k_) * o_ + idx // k_
                if desc.get("context_features"):  # conditional: the context layer of every block, behind the rest
                    for block in net.blocks:
                        rec.append(offsets[id(block.context_layer.weight)])
                        rec.append(offsets[id(block.context_layer.bias)])
                if not is_mlp and inner_bn:  # {gamma, beta, running mean, running variance} of the block's two batch norms
                    for block in net.blocks:
                        for bn in block.batch_norm_layers:
                            if bn.momentum is None or abs(bn.momentum - 0.1) > 1e-12 or not bn.affine:
                                raise NotImplementedError("BatchNorm1d with non-default momentum / affine=False")
                            inner_bn_eps = float(bn.eps)
                            rec.extend([offsets[id(bn.weight)], offsets[id(bn.bias)], offsets[id(bn.running_mean)],
                                        offsets[id(bn.running_var)]])
                            self._inner_bns.append(bn)
                records.append(rec)
            elif isinstance(t, M.BatchNorm):
                rec = [2, len(tables), offsets[id(t.unconstrained_weight)], offsets[id(t.bias)],
                       offsets[id(t.running_mean)], offsets[id(t.running_var)], f32_bits(t.eps),
                       f32_bits(t.momentum)]
                tables.extend(cur)
                records.append(rec)
            elif isinstance(t, M.ActNorm):
                # the data-dependent initialisation (first training batch) is done by the kernels;
                # word [4] says whether it is still due
                rec = [4, len(tables), offsets[id(t.log_scale)], offsets[id(t.shift)],
                       0 if bool(t.initialized) else 1]
                tables.extend(cur)
                records.append(rec)
                self._actnorms.append((len(records) - 1, t))
            elif isinstance(t, M.LULinear):
                if D > 128:  # (beyond 64 features: the layer-wise LU kernels, any width; the limit is the inference kernel's)
                    raise ValueError("LU-linear transforms support at most 128 features")
                rec = [3, len(tables), offsets[id(t.lower_entries)], offsets[id(t.upper_entries)],
                       offsets[id(t.unconstrained_upper_diag)], offsets[id(t.bias)], f32_bits(t.eps)]
                tables.extend(cur)
                records.append(rec)
                cur = list(range(D))
            elif isinstance(t, M.SVDLinear):
                # y = H1(diag(s) H2(x)) + b with Householder sequences H1, H2 (flows/utils.py:316-323): one wave per
                # row, the reflections walked back in the backward pass (csrc/nsa_train.hip, tr_svd_*)
                if D > 128:
                    raise ValueError("SVD-linear transforms support at most 128 features")
                nh = int(t.orthogonal_1.q_vectors.shape[0])
                if int(t.orthogonal_2.q_vectors.shape[0]) != nh:
                    raise ValueError("SVD-linear: the two Householder sequences must have the same length")
                rec = [5, len(tables), offsets[id(t.orthogonal_1.q_vectors)], offsets[id(t.unconstrained_diagonal)],
                       offsets[id(t.orthogonal_2.q_vectors)], offsets[id(t.bias)], f32_bits(t.eps), nh]
                tables.extend(cur)
                records.append(rec)
                cur = list(range(D))
            else:
                raise NotImplementedError(
                    f"{type(t).__name__} has no training kernel (supported: permutation, LU-linear, SVD-linear, "
                    "coupling with a residual / MLP conditioner, batch norm, act norm)"
                )
        if n_res is None:
            raise ValueError("the flow has no coupling layer")
        rec_len = max(len(r) for r in records)
        records = [r + [0] * (rec_len - len(r)) for r in records]
        # parameters per transformed feature as the conditioner emits them (the inference kernels' generic spline
        # instance pads them to 16 bins - pack.transform_policy - the training kernels take the flow's own count)
        n_tr = (3 * int(desc.get("bins", 0)) - 1 if desc["transform"] == "rq"
                else pack.transform_policy(desc["transform"], 0)[2])
        rec_words = len(records[0])
        header = [D, desc["hidden"], n_res, len(records), pack.TRANSFORMS[desc["transform"]],
                  desc.get("bins", 0) if desc["transform"] == "rq" else 0,
                  pack.ACTIVATIONS[desc["activation"]], n_tr, total, rec_words, len(tables),
                  pack.NETS[desc["net"]]]
        header.append(len(tables) + total)  # [12] final column table: logical output i -> physical column
        #: context features of a conditional flow: [13] of the plan; the context rows of a batch travel inside the
        #: workspace (``_put_context``)
        self.context_features = int(desc.get("context_features", 0) or 0)
        header.append(self.context_features)
        #: dropout of the residual blocks, training-mode passes only: [14] the probability (float bits), [15] the
        #: device address of {seed, step} - the masks come from a Philox stream keyed by them (csrc/nsa_train.hip,
        #: tr_dropout_mask), the step ticks on the device after every training-mode forward pass.  Such a flow
        #: trains on the layer-wise kernels.
        self.dropout_probability = float(dropout or 0.0)
        self.drop_state = None
        if self.dropout_probability > 0.0:
            seed = int(torch.randint(0, 2**62, (1,)).item())  # (from torch's global generator: seeded runs repeat)
            self.drop_state = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        header.append(f32_bits(self.dropout_probability) if self.dropout_probability > 0.0 else 0)
        header.append(self.drop_state.data_ptr() if self.drop_state is not None else 0)
        #: batch norm within the residual blocks: [16] flag, [17] eps (float bits); batch statistics in training-mode
        #: passes, running statistics otherwise (csrc/nsa_train.hip, tr_bn1d_forward); the layer-wise kernels
        self.inner_batch_norm = bool(self._inner_bns)
        header.append(1 if self.inner_batch_norm else 0)
        header.append(f32_bits(inner_bn_eps))
        tables = list(tables) + tmap.tolist() + [int(c) for c in cur]
        plan = header + [w for rec in records for w in rec]
        self.plan = (ctypes.c_longlong * len(plan))(*plan)
        self._rec_base, self._rec_words = len(header), rec_words
        self.tables = torch.tensor(tables, dtype=torch.int32, device=device)
        min_d = np.float32(desc.get("min_derivative", 1e-3))
        edge_raw = np.float32(math.log(math.exp(1 - float(min_d)) - 1))
        edge_d = np.float32(min_d + np.float32(max(edge_raw, 0)) + np.float32(np.log1p(np.exp(-abs(edge_raw)))))
        var = float(desc.get("base_var", 1.0))
        inv_div = 1.0 / math.sqrt(desc["hidden"]) if desc.get("scale_logits", True) else 1.0
        consts = [desc.get("tail_bound", 5.0), desc.get("min_bin_width", 1e-3),
                  desc.get("min_bin_height", 1e-3), float(min_d), float(edge_d), inv_div,
                  1.0 / var, 0.5 * D * math.log(2 * math.pi * var),
                  float(desc.get("affine_shift", 2.0)), float(desc.get("affine_eps", 1e-3))]
        self.consts = (ctypes.c_float * len(consts))(*consts)
        self.features = D
        self.losses = torch.zeros(2, dtype=torch.float32, device=device)  # one device-to-host copy per epoch
        self.loss_acc = self.losses[0:1]
        self.val_acc = self.losses[1:2]
        #: {lr, optimiser steps done, 1 - beta1^t, sqrt(1 - beta2^t)} for nsa_train_step_dev
        self.opt = torch.zeros(4, dtype=torch.float32, device=device)
        self._ws = None
        self._ws_rows = 0
        self.n_params = total
        self._params = params
        self._stats = stats
        self._bound = [(t, offsets[id(t)]) for t in params + stats]
        self._perm_versions = self._permutation_versions()
        self.has_batch_norm = any(r[0] == 2 for r in records)
        #: (offset, numel, shape) of every Parameter, in ``flow.parameters()`` order
        self.param_slices = [(offsets[id(p)], p.numel(), tuple(p.shape)) for p in params]

    def configure_optimiser(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, optimiser="adamw",
                            reset=True):
        """Hyper-parameters of the fused optimiser step; ``reset`` clears the moments and the
        step count (a fresh optimiser, as ``FlowModel.reset_model`` makes one, flowmodel/base.py:748-772)."""
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.base_lr = float(lr)
        self.optimiser = self.OPTIMISERS[optimiser]
        if reset:
            self.step_count = 0
            self.adam_m.zero_()
            self.adam_v.zero_()

    def flat_range(self):
        """``range`` of byte addresses covered by the flat buffer (is a tensor a view of it?)."""
        base = self.flat.data_ptr()
        return range(base, base + self.flat.numel() * self.flat.element_size())

    def _permutation_versions(self):
        return tuple(b._version for name, b in self.flow.named_buffers() if name.endswith("_permutation"))

    def is_current(self):
        """False once the flow's tensors were re-allocated (``.to()`` across devices) or its
        permutations re-drawn: the plan's tables and the flat views are then stale."""
        if self._perm_versions != self._permutation_versions():
            return False
        base, item = self.flat.data_ptr(), self.flat.element_size()
        return all(t.data_ptr() == base + off * item for t, off in self._bound)

    # -- differentiable evaluation (nessai_amd.ops) ------------------------------

    def _put_context(self, ws, n, context):
        """Hand the ``(n, C)`` context rows of a conditional flow's batch to the kernels: they read them from the
        workspace, ``nsa_train_context_offset`` floats from its start (the C entry points keep their signatures)."""
        if not self.context_features:
            if context is not None:
                raise ValueError("this flow takes no context")
            return
        if context is None or tuple(context.shape) != (n, self.context_features):
            raise ValueError(f"a conditional flow needs a context of shape ({n}, {self.context_features})")
        off = int(lib.load().nsa_train_context_offset(self.plan, n))
        ws[off : off + n * self.context_features].view(n, self.context_features).copy_(context)

    def forward_saved(self, x, training, want_z=True, want_logdet=True, want_log_prob=True, context=None):
        """Forward pass that keeps every layer's activations in a fresh workspace tensor.

        Returns ``(z, logdet, log_prob, workspace)``; unwanted outputs are ``None``."""
        n = x.shape[0]
        words = int(lib.load().nsa_train_workspace_floats(self.plan, n))
        ws = torch.empty(words, dtype=torch.float32, device=self.device)
        self._put_context(ws, n, context)
        z = torch.empty((n, self.features), dtype=torch.float32, device=self.device) if want_z else None
        ld = torch.empty(n, dtype=torch.float32, device=self.device) if want_logdet else None
        lp = torch.empty(n, dtype=torch.float32, device=self.device) if want_log_prob else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = lib.load().nsa_train_forward(
            self.plan, lib.ptr(self.tables), self.consts, lib.ptr(self.flat), lib.ptr(x), n, lib.ptr(ws),
            1 if training else 0, lib.ptr(z), lib.ptr(ld), lib.ptr(lp), lib.c_void_p(stream),
        )
        lib.check(rc, "nsa_train_forward")
        if training:
            self._count_training_pass()
            self._actnorm_initialised()
        return z, ld, lp, ws

    def backward_saved(self, ws, n, training, g_z=None, g_logdet=None, g_log_prob=None, want_gx=False):
        """Gradients of every trainable float (a fresh flat tensor) and optionally of the input."""
        grads = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        gx = torch.empty((n, self.features), dtype=torch.float32, device=self.device) if want_gx else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = lib.load().nsa_train_backward(
            self.plan, lib.ptr(self.tables), self.consts, lib.ptr(self.flat), lib.ptr(grads), n, lib.ptr(ws),
            1 if training else 0, lib.ptr(g_z), lib.ptr(g_logdet), lib.ptr(g_log_prob), lib.ptr(gx),
            lib.c_void_p(stream),
        )
        lib.check(rc, "nsa_train_backward")
        return grads, gx

    def _count_training_pass(self):
        for bn in self._inner_bns:  # torch's BatchNorm1d counts its training-mode passes (a state_dict entry)
            bn.num_batches_tracked.add_(1)

    def _actnorm_initialised(self):
        for index, module in self._actnorms:  # initialised by the training-mode pass that just ran
            flag = self._rec_base + index * self._rec_words + 4
            if self.plan[flag]:
                self.plan[flag] = 0
                module.initialized.fill_(True)

    def pinned_buffers(self, n_train):
        """Host-pinned staging of the epoch loop (permutation up, the two losses down), kept across fits: pinning
        memory is a driver call of the order of a short fit's epoch."""
        cached = getattr(self, "_pinned", None)
        if cached is None or cached[0].shape[1] < n_train:
            size = max(n_train, 4096)
            # TWO pinned permutation buffers, used in turn: the upload of epoch e's permutation may still be
            # in flight when the look-ahead draws the one of epoch e + 1 (the host only waits for the losses,
            # which are recorded before that upload in stream order)
            cached = (torch.empty((2, size), dtype=torch.int64, pin_memory=True),
                      torch.empty(size, dtype=torch.int64, device=self.device),
                      torch.empty(2, dtype=torch.float32, pin_memory=True))
            self._pinned = cached
        return cached[0][:, :n_train], cached[1][:n_train], cached[2]

    def _workspace(self, n):
        if self._ws is None or n > self._ws_rows:
            words = int(lib.load().nsa_train_workspace_floats(self.plan, n))
            self._ws = torch.empty(words, dtype=torch.float32, device=self.device)
            self._ws_rows = n
        return self._ws

    def step(self, x, clip_norm=5.0, lr=None, weights=None, context=None):
        """One optimiser step on a ``(n, D)`` float32 device batch; the loss is added to ``loss_acc``.

        ``weights`` (``(n,)`` float32 device tensor) selects the weighted loss
        ``-sum(w log_prob) / sum(w)`` (``flowmodel/base.py:404-407``); ``context``: the ``(n, C)`` context rows of
        a conditional flow."""
        self.step_count += 1
        ws = self._workspace(x.shape[0])
        self._put_context(ws, x.shape[0], context)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = lib.load().nsa_train_step(
            self.plan, lib.ptr(self.tables), self.consts, lib.ptr(self.params), lib.ptr(self.grads),
            lib.ptr(self.adam_m), lib.ptr(self.adam_v), lib.ptr(x), lib.ptr(weights), x.shape[0], lib.ptr(ws),
            self.lr if lr is None else lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
            float(clip_norm or 0.0), self.step_count, self.optimiser, lib.ptr(self.loss_acc), lib.c_void_p(stream),
        )
        lib.check(rc, "nsa_train_step")
        self._count_training_pass()
        self._actnorm_initialised()

    def step_dev(self, x, clip_norm=5.0, weights=None, context=None):
        """``step`` with the learning rate and the step counter in device memory (``self.opt``):
        the form whose launches can be captured into a HIP graph (``nsa_train_step_dev``)."""
        self.step_count += 1
        ws = self._workspace(x.shape[0])
        self._put_context(ws, x.shape[0], context)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = lib.load().nsa_train_step_dev(
            self.plan, lib.ptr(self.tables), self.consts, lib.ptr(self.params), lib.ptr(self.grads),
            lib.ptr(self.adam_m), lib.ptr(self.adam_v), lib.ptr(x), lib.ptr(weights), x.shape[0], lib.ptr(ws),
            lib.ptr(self.opt), self.betas[0], self.betas[1], self.eps, self.weight_decay, float(clip_norm or 0.0),
            self.optimiser, lib.ptr(self.loss_acc), lib.c_void_p(stream),
        )
        lib.check(rc, "nsa_train_step_dev")
        self._count_training_pass()
        # the data-dependent act-norm initialisation belongs to the FIRST training batch only (nflows ActNorm):
        # clear the plan's flag now, not at the end of the epoch - every further minibatch would re-initialise
        # log_scale / shift from its own statistics on top of the optimiser's updates.  (No-op, hence safe
        # under graph capture, once the flag is clear: the first epoch always runs launch by launch.)
        self._actnorm_initialised()

    def evaluate(self, x, log_prob_out=None, weights=None, context=None):
        """``loss_acc += -mean(log_prob(x))`` without touching the weights."""
        ws = self._workspace(x.shape[0])
        self._put_context(ws, x.shape[0], context)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = lib.load().nsa_train_eval(
            self.plan, lib.ptr(self.tables), self.consts, lib.ptr(self.params), lib.ptr(x), lib.ptr(weights),
            x.shape[0], lib.ptr(ws), lib.ptr(self.loss_acc), lib.ptr(log_prob_out), lib.c_void_p(stream),
        )
        lib.check(rc, "nsa_train_eval")


def get_trainer(flow_model):
    """The flow's trainer with the optimiser ``FlowModel.get_optimiser`` would build; a new
    optimiser state whenever the FlowModel dropped its handle (``reset_model``)."""
    tc = flow_model.training_config
    trainer = flow_model.model.trainer()
    if flow_model._trainer is not trainer:
        # FlowModel.get_optimiser (src/nessai/flowmodel/base.py:105-135): adam adds weight_decay=1e-6
        name = (tc.get("optimiser") or "adamw").lower()
        kwargs = dict(tc.get("optimiser_kwargs") or {})
        defaults = {"adamw": 0.01, "adam": 1e-6, "sgd": 0.0}
        if name not in defaults:
            raise NotImplementedError(f"optimiser {name!r} has no gfx950 kernel (adamw, adam, sgd)")
        weight_decay = kwargs.pop("weight_decay", defaults[name])
        lr = kwargs.pop("lr", tc["lr"])
        betas, eps = kwargs.pop("betas", (0.9, 0.999)), kwargs.pop("eps", 1e-8)
        if name == "sgd":
            for k in ("momentum", "dampening", "nesterov"):
                if kwargs.pop(k, 0):
                    raise NotImplementedError(f"SGD with {k} has no gfx950 kernel")
        if kwargs.pop("amsgrad", False):
            raise NotImplementedError("amsgrad has no gfx950 kernel")
        if kwargs:
            raise TypeError(f"Unknown optimiser arguments: {sorted(kwargs)}")
        trainer.configure_optimiser(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, optimiser=name)
        flow_model._trainer = trainer
    return flow_model._trainer


@contextlib.contextmanager
def _training_kernels(deterministic):
    """``training_config["deterministic"]``: run this fit on the layer-wise kernels, whose reductions
    have a fixed order (the fused coupling kernels add weight gradients with float atomics).  The
    library reads the switch per call."""
    before = os.environ.get("NESSAI_AMD_TRAIN_FUSED")
    if deterministic:
        os.environ["NESSAI_AMD_TRAIN_FUSED"] = "0"
    try:
        yield
    finally:
        if deterministic:
            if before is None:
                os.environ.pop("NESSAI_AMD_TRAIN_FUSED", None)
            else:
                os.environ["NESSAI_AMD_TRAIN_FUSED"] = before


def train_flow(flow_model, samples, max_epochs=None, patience=None, val_size=None, weights=None, conditional=None):
    tc = flow_model.training_config
    trainer = get_trainer(flow_model)
    trainer.lr = trainer.base_lr
    # Gaussian noise added to every training minibatch (flowmodel/base.py:424-425, 596-605)
    noise_scale = None
    if tc.get("noise_type") == "adaptive":
        from scipy.spatial import distance

        d = distance.cdist(samples, samples, "euclidean")  # utils/distance.py:10-30
        d[d == 0] = np.nan
        noise_scale = float(tc["noise_scale"] * np.mean(np.nanmin(d, axis=1)))
    elif tc.get("noise_type") == "constant":
        noise_scale = float(tc["noise_scale"])
    elif tc.get("noise_type"):
        raise ValueError(f"Unknown noise type: {tc['noise_type']}")
    model = flow_model.model
    validate = val_size != 0.0
    train_data, val_data, batch_size = flow_model.prep_data(
        samples, val_size=val_size, batch_size=tc["batch_size"], weights=weights, conditional=conditional
    )
    w_train = w_val = c_train = c_val = None
    if weights is not None:
        (train_data, w_train), (val_data, w_val) = train_data, val_data
    elif conditional is not None:  # (flowmodel/base.py:286-302: the conditional inputs are shuffled and split with the samples)
        (train_data, c_train), (val_data, c_val) = train_data, val_data
    if max_epochs is None:
        max_epochs = tc["max_epochs"]
    if patience is None:
        patience = tc["patience"]
    clip = tc["clip_grad_norm"]
    best_epoch, best_val_loss = 0, np.inf
    trainer.best.copy_(trainer.flat)
    history = dict(loss=[], val_loss=[])
    model.train()
    # One epoch = the minibatch steps + the validation pass.  Live-point sets are small (2000 rows:
    # two minibatches), so an epoch is ~35 short launches and launch overhead dominates: after the
    # first epoch (which also runs the data-dependent act-norm initialisation) the launches of an
    # epoch are captured ONCE into a HIP graph and every later epoch is one graph launch.  Nothing
    # in the captured sequence depends on a per-epoch host value: the shuffled rows are gathered
    # into a static buffer, the learning rate and the optimiser's step counter live in device
    # memory (Trainer.opt).  NESSAI_AMD_EPOCH_GRAPH=0 keeps the launch-by-launch loop.
    # (Round 3 also ran the validation pass of epoch e beside the training steps of epoch e + 1 - second
    # stream, snapshot of the weights, lagged early-stopping bookkeeping; same losses, weights and stopping
    # epoch - and measured 0.805 against 0.827 ms per step with the graph, 0.829 against 0.805 without: nothing.
    # The epoch is a chain of latency-bound launches either way; not kept.  DESIGN.md section 4.3.)
    n_train = train_data.shape[0]
    epoch_rows = torch.empty_like(train_data)
    epoch_w = torch.empty_like(w_train) if w_train is not None else None
    epoch_c = torch.empty_like(c_train) if c_train is not None else None
    bounds = [(a, min(a + batch_size, n_train)) for a in range(0, n_train, batch_size)]
    has_val = bool(validate and len(val_data))
    trainer.opt[1] = float(trainer.step_count)

    def run_epoch():
        trainer.losses.zero_()
        for a, b in bounds:
            trainer.step_dev(epoch_rows[a:b], clip_norm=clip, weights=None if epoch_w is None else epoch_w[a:b],
                             context=None if epoch_c is None else epoch_c[a:b])
        if has_val:
            keep, trainer.loss_acc = trainer.loss_acc, trainer.val_acc
            trainer.evaluate(val_data, weights=w_val, context=c_val)
            trainer.loss_acc = keep

    # The host's share of an epoch - the permutation, its upload, the gather of the shuffled rows - is queued
    # for epoch e + 1 BEHIND the launches of epoch e and before the host waits for epoch e's losses, so the device
    # goes from one epoch to the next without waiting for it (the gather writes the buffer the running epoch
    # reads: stream order keeps it behind).  If epoch e turns out to be the last one, the generators are put
    # back: the look-ahead leaves no trace in torch's random streams.
    # (the permutation goes up from pinned memory: an upload from pageable memory holds the host until the stream
    # has drained - the whole running epoch)
    perm_host, perm_dev, losses_host = trainer.pinned_buffers(n_train)

    turn = [0]

    def prepare_rows():
        # alternate the pinned buffers: overwriting the one whose asynchronous upload may not have finished
        # would tear the previous epoch's permutation (rows duplicated or missing, silently)
        host = perm_host[turn[0]]
        turn[0] ^= 1
        torch.randperm(n_train, out=host)
        perm_dev.copy_(host, non_blocking=True)
        p = perm_dev
        torch.index_select(train_data, 0, p, out=epoch_rows)
        if epoch_w is not None:
            torch.index_select(w_train, 0, p, out=epoch_w)
        if epoch_c is not None:
            torch.index_select(c_train, 0, p, out=epoch_c)
        if noise_scale:
            epoch_rows.add_(noise_scale * torch.randn_like(epoch_rows))

    def generator_states():
        return torch.get_rng_state(), (torch.cuda.get_rng_state(train_data.device) if noise_scale else None)

    def restore_generators(states):
        torch.set_rng_state(states[0])
        if states[1] is not None:
            torch.cuda.set_rng_state(states[1], train_data.device)

    use_graph = os.environ.get("NESSAI_AMD_EPOCH_GRAPH", "1") != "0"
    graph = None
    losses_ready = torch.cuda.Event()
    with _training_kernels(tc.get("deterministic")):
        prepare_rows()
        trainer.opt[0] = trainer.lr
        for epoch in range(1, max_epochs + 1):
            if tc.get("annealing"):
                # CosineAnnealingLR(T_max=max_epochs, eta_min=0), stepped once per epoch (base.py:449-450, 628-631)
                trainer.lr = trainer.base_lr * 0.5 * (1.0 + math.cos(math.pi * (epoch - 1) / max_epochs))
                trainer.opt[0] = trainer.lr
            if graph is not None:
                graph.replay()
                trainer.step_count += len(bounds)
            elif use_graph and epoch == 2:
                graph = torch.cuda.CUDAGraph()
                steps_before = trainer.step_count
                with torch.cuda.graph(graph):
                    run_epoch()
                trainer.step_count = steps_before  # capture records, it does not run
                graph.replay()
                trainer.step_count += len(bounds)
            else:
                run_epoch()
                trainer._actnorm_initialised()
            # the copy of the two losses is queued first, so the wait below is for this epoch's launches only
            losses_host.copy_(trainer.losses, non_blocking=True)
            losses_ready.record()
            if epoch < max_epochs:
                states = generator_states()
                prepare_rows()
            losses_ready.synchronize()  # the epoch's one sync
            both = losses_host.numpy()
            loss, val_loss = float(both[0]) / len(bounds), (float(both[1]) if has_val else np.nan)
            history["loss"].append(loss)
            history["val_loss"].append(val_loss)
            if validate and (val_loss < best_val_loss):
                best_epoch, best_val_loss = epoch, val_loss
                trainer.best.copy_(trainer.flat)
            if validate and (epoch - best_epoch > patience):
                if epoch < max_epochs:
                    restore_generators(states)
                break
    if validate:
        trainer.flat.copy_(trainer.best)
    model._image_key = None  # weights changed behind torch's version counters
    model.eval()
    return history
