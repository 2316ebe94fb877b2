"""PyTorch-ROCm custom operators over the C ABI (``torch.ops.nessai_amd.*``).

The reference drives a flow through plain torch calls: inference under
``torch.inference_mode()`` and training through autograd
(``src/nessai/flowmodel/base.py:400-444``: ``loss = -model.log_prob(x).mean();
loss.backward(); clip_grad_norm_(model.parameters(), ...); optimiser.step()``).
This module registers the gfx950 kernels as dispatcher-visible operators (schema + ROCm
kernel + shape function, so they can be traced and introspected like any ATen op) and wires
the two differentiable ones into autograd:

===============================  =========================================================
``nessai_amd::flow_run``         whole flow, inference kernel (``nsa_flow_run``)
``nessai_amd::flow_run_ctx``     the same for a conditional flow (``nsa_flow_run_ctx``)
``nessai_amd::flow_forward``     whole flow in exact float32, activations kept in a
                                 workspace tensor (``nsa_train_forward``)
``nessai_amd::flow_backward``    reverse mode of the above for arbitrary upstream
                                 gradients (``nsa_train_backward``)
===============================  =========================================================

There is no CPU implementation: the operators exist for the ``CUDA`` (ROCm) dispatch key
only, plus ``Meta`` shape functions.
"""

import weakref

import os

import numpy as np
import torch

from . import lib

_LIB = torch.library.Library("nessai_amd", "DEF")

_LIB.define(
    "flow_run(Tensor x, Tensor image, Tensor header, int direction, int mode, bool want_x, "
    "Tensor(a!) status, int waves_per_group, int max_groups) -> (Tensor, Tensor)"
)
_LIB.define(
    "flow_run_ctx(Tensor x, Tensor context, Tensor image, Tensor header, int direction, int mode, bool want_x, "
    "Tensor(a!) status, int waves_per_group, int max_groups) -> (Tensor, Tensor)"
)
_LIB.define(
    "flow_forward(Tensor x, Tensor(a!) flat, int plan, bool training, bool want_z, bool want_logdet, "
    "bool want_log_prob, Tensor? context=None) -> (Tensor, Tensor, Tensor, Tensor)"
)
_LIB.define(
    "flow_backward(Tensor(a!) workspace, Tensor flat, int plan, int n, bool training, Tensor? g_z, "
    "Tensor? g_logdet, Tensor? g_log_prob, bool want_gx) -> (Tensor, Tensor)"
)

#: plan handle -> Trainer (the plan, tables and constants of one flow live on the host side)
_PLANS = weakref.WeakValueDictionary()
_NEXT_HANDLE = [1]


def register_plan(trainer):
    handle = _NEXT_HANDLE[0]
    _NEXT_HANDLE[0] += 1
    _PLANS[handle] = trainer
    return handle


def _plan(handle):
    trainer = _PLANS.get(int(handle))
    if trainer is None:
        raise RuntimeError(f"nessai_amd: unknown flow plan handle {handle} (the flow was deleted)")
    return trainer


def _empty(like, *shape):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# -- ROCm kernels --------------------------------------------------------------------------


def _flow_run(x, image, header, direction, mode, want_x, status, waves_per_group, max_groups):
    n = x.shape[0]
    out = torch.empty_like(x) if want_x else _empty(x, 0)
    ld = _empty(x, n)
    status.zero_()
    if n == 0:
        return out, ld
    if header.dtype != torch.int32 or header.device.type != "cpu" or not header.is_contiguous():
        raise RuntimeError("nessai_amd::flow_run: `header` must be the host int32 tensor of ops.header_tensor")
    stream = torch.cuda.current_stream(x.device).cuda_stream
    rc = lib.load().nsa_flow_run(
        lib.c_void_p(header.data_ptr()), lib.ptr(image), int(direction), int(mode), n, lib.ptr(x), lib.ptr(out) if want_x else None,
        lib.ptr(ld), lib.ptr(status), int(waves_per_group), int(max_groups), lib.c_void_p(stream),
    )
    lib.check(rc, "nsa_flow_run")
    return out, ld


def _flow_run_ctx(x, context, image, header, direction, mode, want_x, status, waves_per_group, max_groups):
    """``flow_run`` of a conditional flow (``nsa_flow_run_ctx``): ``context`` is ``(n, C)`` float32."""
    n = x.shape[0]
    out = torch.empty_like(x) if want_x else _empty(x, 0)
    ld = _empty(x, n)
    status.zero_()
    if n == 0:
        return out, ld
    if header.dtype != torch.int32 or header.device.type != "cpu" or not header.is_contiguous():
        raise RuntimeError("nessai_amd::flow_run_ctx: `header` must be the host int32 tensor of ops.header_tensor")
    if context.shape[0] != n or context.dtype != torch.float32 or not context.is_contiguous() or context.device != x.device:
        raise RuntimeError("nessai_amd::flow_run_ctx: `context` must be a contiguous float32 (n, C) tensor on x's device")
    stream = torch.cuda.current_stream(x.device).cuda_stream
    rc = lib.load().nsa_flow_run_ctx(
        lib.c_void_p(header.data_ptr()), lib.ptr(image), int(direction), int(mode), n, lib.ptr(x), lib.ptr(context),
        lib.ptr(out) if want_x else None, lib.ptr(ld), lib.ptr(status), int(waves_per_group), int(max_groups),
        lib.c_void_p(stream),
    )
    lib.check(rc, "nsa_flow_run_ctx")
    return out, ld


def _flow_forward(x, flat, plan, training, want_z, want_logdet, want_log_prob, context=None):
    trainer = _plan(plan)
    if flat.data_ptr() != trainer.flat.data_ptr():
        raise RuntimeError("nessai_amd::flow_forward: `flat` is not the plan's parameter buffer")
    z, ld, lp, ws = trainer.forward_saved(x, training, want_z, want_logdet, want_log_prob, context=context)
    if training and (trainer._stats or trainer._actnorms):
        # batch norm / act norm: the kernels just rewrote running statistics (or initialised act norm)
        trainer.stats_version = getattr(trainer, "stats_version", 0) + 1
    none = _empty(x, 0)
    return (z if z is not None else none, ld if ld is not None else none, lp if lp is not None else none, ws)


def _flow_backward(workspace, flat, plan, n, training, g_z, g_logdet, g_log_prob, want_gx):
    trainer = _plan(plan)
    grads, gx = trainer.backward_saved(workspace, int(n), training, g_z, g_logdet, g_log_prob, want_gx)
    return grads, gx if gx is not None else _empty(flat, 0)


_LIB.impl("flow_run", _flow_run, "CUDA")
_LIB.impl("flow_run_ctx", _flow_run_ctx, "CUDA")
_LIB.impl("flow_forward", _flow_forward, "CUDA")
_LIB.impl("flow_backward", _flow_backward, "CUDA")


# -- shape functions (Meta) -----------------------------------------------------------------


def _flow_run_meta(x, image, header, direction, mode, want_x, status, waves_per_group, max_groups):
    return (torch.empty_like(x) if want_x else x.new_empty((0,)), x.new_empty((x.shape[0],)))


def _flow_forward_meta(x, flat, plan, training, want_z, want_logdet, want_log_prob, context=None):
    n = x.shape[0]
    words = int(lib.load().nsa_train_workspace_floats(_plan(plan).plan, n))
    return (
        torch.empty_like(x) if want_z else x.new_empty((0,)),
        x.new_empty((n if want_logdet else 0,)),
        x.new_empty((n if want_log_prob else 0,)),
        x.new_empty((words,)),
    )


def _flow_backward_meta(workspace, flat, plan, n, training, g_z, g_logdet, g_log_prob, want_gx):
    trainer = _plan(plan)
    return (flat.new_empty((trainer.n_params,)),
            flat.new_empty((n, trainer.features)) if want_gx else flat.new_empty((0,)))


_LIB.impl("flow_run", _flow_run_meta, "Meta")
_LIB.impl("flow_run_ctx", lambda x, context, *rest: _flow_run_meta(x, *rest), "Meta")
_LIB.impl("flow_forward", _flow_forward_meta, "Meta")
_LIB.impl("flow_backward", _flow_backward_meta, "Meta")


# -- autograd -------------------------------------------------------------------------------


class FlowFunction(torch.autograd.Function):
    """``(z, logdet, log_prob) = flow(x)`` differentiable in ``x`` and every ``Parameter``.

    The parameters are passed as inputs only to tie the node into the graph: the kernels read
    the flat buffer the parameters are views of (``Trainer``), and the backward pass hands
    every parameter its slice of the flat gradient.
    """

    @staticmethod
    def forward(ctx, x, trainer, training, want_z, want_logdet, want_log_prob, context, *params):
        z, ld, lp, ws = torch.ops.nessai_amd.flow_forward(
            x, trainer.flat, trainer.handle, bool(training), bool(want_z), bool(want_logdet), bool(want_log_prob),
            context,
        )
        ctx.trainer = trainer
        ctx.workspace = ws
        # what the backward kernels will read besides the workspace: the live flat parameter buffer (the
        # Parameters are views of it) and the kernel family.  Nothing goes through save_for_backward, so
        # torch's own version check does not see an optimiser step or a second training-mode forward between
        # this call and its backward: the node checks by itself.
        ctx.flat_version = _state_version(trainer)  # AFTER the forward (it may rewrite statistics itself)
        ctx.fused = os.environ.get("NESSAI_AMD_TRAIN_FUSED")
        ctx.n = x.shape[0]
        ctx.training = bool(training)
        ctx.want = (want_z, want_logdet, want_log_prob)
        ctx.mark_non_differentiable(*[t for t, w in zip((z, ld, lp), ctx.want) if not w])
        return z, ld, lp

    @staticmethod
    def backward(ctx, g_z, g_ld, g_lp):
        trainer = ctx.trainer
        if ctx.workspace is None:
            raise RuntimeError(
                "nessai_amd: backward through a flow evaluation whose activations were already consumed "
                "(retain_graph=True / a second backward is not supported: run the forward pass again)"
            )
        if _state_version(trainer) != ctx.flat_version:
            raise RuntimeError(
                "nessai_amd: a parameter or running statistic of the flow was modified in place between the "
                "forward pass and its backward (an optimiser step or another training-mode forward); the "
                "gradients would be computed against the new values"
            )
        if os.environ.get("NESSAI_AMD_TRAIN_FUSED") != ctx.fused:
            raise RuntimeError("nessai_amd: NESSAI_AMD_TRAIN_FUSED changed between the forward pass and its backward")
        want_z, want_ld, want_lp = ctx.want

        def grad(g, wanted):
            return g.to(torch.float32).contiguous() if (wanted and g is not None) else None

        want_gx = ctx.needs_input_grad[0]
        grads, gx = torch.ops.nessai_amd.flow_backward(
            ctx.workspace, trainer.flat, trainer.handle, ctx.n, ctx.training, grad(g_z, want_z),
            grad(g_ld, want_ld), grad(g_lp, want_lp), bool(want_gx),
        )
        ctx.workspace = None  # a second backward would read activations that were consumed
        per_param = [
            grads[off : off + numel].view(shape) if need else None
            for (off, numel, shape), need in zip(trainer.param_slices, ctx.needs_input_grad[7:])
        ]
        return (gx if want_gx else None, None, None, None, None, None, None, *per_param)


def _state_version(trainer):
    """Changes when a parameter / buffer of the flow is written in place through torch (the Parameters are
    views of ``trainer.flat`` with version counters of their own) or when a training-mode forward of a flow
    with batch norm / act norm rewrote its statistics inside the kernels (``stats_version``)."""
    flow = trainer.flow
    tensors = flow.__dict__.get("_fp_tensors")
    if tensors is None:
        tensors = list(flow.parameters()) + list(flow.buffers())
    return (getattr(trainer, "stats_version", 0), tuple(t._version for t in tensors))


def differentiable_flow(flow, x, want_z, want_logdet, want_log_prob, context=None):
    """Run ``flow`` forward under autograd; returns ``(z, logdet, log_prob)`` (``None`` if unwanted)."""
    trainer = flow.trainer()
    if flow.training and x.shape[0] < 2:
        raise ValueError("a training-mode pass needs at least two rows")
    params = list(flow.parameters())
    z, ld, lp = FlowFunction.apply(x, trainer, flow.training, want_z, want_logdet, want_log_prob, context, *params)
    flow._image_key = None  # training-mode passes update running statistics behind the version counters
    return (z if want_z else None, ld if want_logdet else None, lp if want_log_prob else None)


def header_tensor(header):
    """Host int32 tensor holding the 32-bit header words bit for bit (operator argument of ``flow_run``, which
    hands its address to the library: no per-call conversion)."""
    return torch.from_numpy(np.frombuffer(bytes(header), dtype=np.uint32).view(np.int32).copy())
