"""Flow plug-in surface and its HIP implementation.

``BaseFlow`` restates the abstract interface the reference's ``FlowModel``
programs against (``src/nessai/flows/base.py:11-167``): a flow class selected
with ``flow_config["flow"]`` must provide these methods
(``src/nessai/flows/utils.py:231-246``).  ``HipFlow`` is the counterpart of
the reference's ``NFlow`` (``flows/base.py:170-316``) whose transform and base
distribution run in the gfx950 kernels of ``nessai_amd/csrc``.
"""

import math
import os
from abc import ABC, abstractmethod

import numpy as np
import torch
from torch.nn import Module

from .. import lib, ops, pack
from . import modules as M


class BaseFlow(Module, ABC):
    """Abstract flow API (same names, arguments and return values as the reference)."""

    device = None

    def to(self, device):
        """Store the device, then move the parameters (returns None, as the reference)."""
        self.device = torch.device(device)
        super().to(device)

    @abstractmethod
    def forward(self, x, context=None):
        raise NotImplementedError()

    @abstractmethod
    def inverse(self, z, context=None):
        raise NotImplementedError()

    @abstractmethod
    def sample(self, n, context=None):
        raise NotImplementedError()

    @abstractmethod
    def log_prob(self, x, context=None):
        raise NotImplementedError()

    @abstractmethod
    def sample_latent_distribution(self, n, context=None):
        raise NotImplementedError

    @abstractmethod
    def base_distribution_log_prob(self, z, context=None):
        raise NotImplementedError()

    @abstractmethod
    def forward_and_log_prob(self, x, context=None):
        raise NotImplementedError()

    @abstractmethod
    def sample_and_log_prob(self, n, context=None):
        raise NotImplementedError()

    def finalise(self):
        pass

    def end_iteration(self):
        pass

    @abstractmethod
    def freeze_transform(self):
        raise NotImplementedError()

    @abstractmethod
    def unfreeze_transform(self):
        raise NotImplementedError()


MODE_APPLY, MODE_LOGPROB = 0, 1
DIR_FORWARD, DIR_INVERSE = 0, 1

#: Arithmetic of the conditioner GEMMs (see ``nsa_format.h``).
#: ``"f32"`` (default): exact float32 products on ``v_mfma_f32_32x32x2_f32`` - meets the
#: 1e-5 relative ``log_prob`` parity bar for every configuration.
#: ``"bf16x3"``: operands split ``x = hi + lo`` into two bf16, products ``hi*hi + hi*lo + lo*hi``
#: on the bf16 matrix cores with float32 accumulation - 2x faster; measured ``log_prob`` error
#: 1e-6..7e-5 relative depending on the flow (within the bar for the spline flows of the
#: benchmark configs, outside it for small or affine flows; table in DESIGN.md).
#: ``"f16x3"``: the same three products on f16 pairs (11 + 11 significant bits: operands carried to
#: 2^-24 relative, i.e. float32-class products at the bf16 matrix rate) for operands below 65520 in
#: magnitude; a call that meets a larger value is detected (``status[1]``) and raises
#: ``FloatingPointError`` - under ``"auto"`` it is repeated in exact float32 instead.
#: ``"auto"``: the fastest mode MEASURED to agree with float32 on the current weights
#: (``HipFlow._pack_auto``).
#: Override per flow (``flow.precision = ...``) or with ``NESSAI_AMD_PRECISION``.
DEFAULT_PRECISION = os.environ.get("NESSAI_AMD_PRECISION", "f32")


def populate_precision(flow):
    """First candidate of ``precision = "auto"`` (the probe of ``HipFlow._pack_auto`` decides):
    split-f16 - float32-class products at the 16-bit matrix rate - for every flow."""
    return "f16x3"


def bf16x3_envelope(flow):
    """Static envelope in which split-bf16 is worth probing: rational-quadratic flows with >= 15
    features and >= 30 hidden units (configs 3, 4 and 5: <= 3.3e-6 relative; a grid of randomised
    flows with 16-32 features: <= 6.6e-6).  Narrower flows have a small |log_prob| and reach
    1.4e-5 - 2.8e-5 relative in split-bf16, affine flows 3.3e-5: outside the 1e-5 bar."""
    return flow.transform_kind == "rq" and flow.hidden_features >= 30 and flow.features >= 15


def auto_candidates(flow):
    """Fast modes ``precision = "auto"`` probes, in the order it prefers them; exact float32 is the
    fallback behind the last one.  Split-f16 first (as fast as split-bf16 - 1.68 against 1.74 ms per
    1e6 rows of config 3 - and ~100x closer to float32); split-bf16, whose operands cannot leave
    their range, behind it inside its envelope: it is what is left when a flow's activations
    exceed the f16 range."""
    order = os.environ.get("NESSAI_AMD_AUTO_ORDER")
    if order:
        return tuple(p for p in order.split(",") if p)
    return ("f16x3", "bf16x3") if bf16x3_envelope(flow) else ("f16x3",)


class F16RangeError(FloatingPointError):
    """A value beyond the f16 range (65520) met a split-f16 matrix product."""


class HipFlow(BaseFlow):
    """Transform + base distribution evaluated by the fused gfx950 flow kernel.

    Inputs must be contiguous float32 tensors on a ROCm device; outputs are
    allocated on the same device.  There is no CPU path: calling a method with
    CPU tensors raises ``RuntimeError``.

    Two kernel families sit behind the methods (``nessai_amd.ops``):

    * inference (no gradient wanted, eval mode): the fused flow kernel ``nessai_amd::flow_run``
      in the flow's ``precision``;
    * ``forward`` / ``log_prob`` / ``forward_and_log_prob`` when autograd is recording and the
      input or a parameter requires a gradient, or in training mode with batch norm / act norm
      (batch statistics): ``nessai_amd::flow_forward`` + ``flow_backward`` in exact float32, so
      the reference's own training loop (``flowmodel/base.py:400-444``) can drive the class.
      ``inverse`` / ``sample`` are not differentiable (the reference never trains through them).
    """

    def __init__(self, transform, distribution):
        super().__init__()
        if not isinstance(transform, M.CompositeTransform):
            raise TypeError("transform must be a nessai_amd CompositeTransform")
        if not isinstance(distribution, M.StandardNormal):
            raise TypeError(
                "distribution must be a nessai_amd StandardNormal; other base "
                "distributions have no gfx950 kernel"
            )
        self._transform = transform
        self._distribution = distribution
        self._refuse_float64()
        self._image_key = None
        self._image_dev = None
        self._header = None
        self._header_t = None
        self._status = None
        self.__dict__["_trainer"] = None
        self.waves_per_group = 0
        self.max_groups = 0
        self.precision = DEFAULT_PRECISION
        self.effective_precision = self.precision if self.precision != "auto" else "f32"
        self.probe_error = None
        self.probe_errors = {}
        self._f16_blocked = None  # fingerprint of the weights for which a split-f16 call left the f16 range
        if self.precision not in pack.PRECISIONS and self.precision != "auto":
            raise ValueError(f"unknown precision {self.precision!r}; choose from {sorted(pack.PRECISIONS)} or 'auto'")
        desc = M.describe(self)
        #: context features of a conditional flow (``context_features`` of the reference's constructors), 0 = none
        self.context_features = int(desc.get("context_features", 0) or 0)
        self.features = desc["D"]
        self.transform_kind = desc["transform"]
        self.hidden_features = desc["hidden"]
        ht = pack.hidden_tiles(desc["hidden"])
        if ht > 8 or not lib_supports(ht, desc["transform"], desc.get("bins", 0)):
            raise ValueError(
                "No gfx950 kernel instance for this flow configuration "
                f"(hidden={desc['hidden']}, transform={desc['transform']}, "
                f"bins={desc.get('bins', 0)})"
            )
        # every other structural limit of the kernels (LU-linear beyond 64 features, mixed conditioner
        # depths, ...) is known to the packer: pack once now so that an unsupported configuration raises
        # at construction like the checks above, not at the first call
        pack.pack_flow(dict(desc, precision=self.precision if self.precision != "auto" else "f32"))

    _FLOAT64_REFUSAL = (
        "nessai_amd flows compute in float32 (split-f16 / f32 matrix products, float32 spline; float64 only in the "
        "populate epilogue): there is no float64 kernel, and nothing is cast down silently. {what} - the "
        "reference's torch_dtype='float64' (src/nessai/utils/torchutils.py:12-25, flowsampler.py:101,121) is not "
        "available on this device path; keep torch's default dtype float32 and pass float32 tensors."
    )

    def _refuse_float64(self, x=None):
        """Float64 is refused, never downgraded: a flow built under ``torch.set_default_dtype(torch.float64)`` (or
        converted with ``.double()``) and float64 inputs raise - nflows raises a dtype mismatch for float64 inputs
        to a float32 flow as well."""
        for p in self.parameters():
            if p.dtype == torch.float64:
                raise TypeError(self._FLOAT64_REFUSAL.format(
                    what="This flow holds float64 parameters (torch default dtype "
                         f"{torch.get_default_dtype()}, or .double() was called)"))
            break
        if x is not None and x.dtype == torch.float64:
            raise TypeError(self._FLOAT64_REFUSAL.format(what="The input tensor is float64"))

    def double(self):
        raise TypeError(self._FLOAT64_REFUSAL.format(what=".double() was called on the flow"))

    def apply(self, fn):
        """``torch.nn.Module.apply``, with the reference's two reset visitors
        (``nessai.flows.utils.reset_weights`` / ``reset_permutations``, which ``FlowModel.reset_model``
        applies: ``flowmodel/base.py:748-772``) replaced by their counterparts here: the reference's
        test ``isinstance(module, glasflow...BatchNorm / LULinear / RandomPermutation)``, which the module
        tree of this class - same names, same parameters, other classes - cannot satisfy."""
        module, name = getattr(fn, "__module__", None) or "", getattr(fn, "__name__", "")
        if module.startswith("nessai.") and name in ("reset_weights", "reset_permutations"):
            from . import utils as own

            fn = getattr(own, name)
        return super().apply(fn)

    # -- device image ----------------------------------------------------------

    def _fingerprint(self):
        """Changes whenever a weight or buffer may have changed.  The tensor list is collected
        once (the module tree is fixed after construction; ``.to()`` / ``load_state_dict`` /
        the trainer's flat buffer rebind ``.data`` of the same objects): walking the tree costs
        ~0.3 ms per call, the cached list ~20 us."""
        tensors = self.__dict__.get("_fp_tensors")
        if tensors is None:
            tensors = list(self.parameters()) + list(self.buffers())
            self.__dict__["_fp_tensors"] = tensors
        key = [self.training]
        for t in tensors:
            key.append((t.data_ptr(), t._version))
        return tuple(key)

    def _device(self):
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    def device_image(self):
        """(header ctypes array, image tensor on the device), repacked when weights change."""
        device = self._device()
        if device.type != "cuda":
            raise RuntimeError(
                "nessai_amd flows run on a ROCm device only; move the flow with "
                ".to('cuda') (there is no CPU fallback)"
            )
        key = (device, self.precision, self._fingerprint())
        if key != self._image_key:
            # `_image_key = None` is how the trainer, load paths and note_f16_overflow say "the weights changed
            # behind torch's version counters": every cached image is then stale, whatever the fingerprint says
            old = self._image_key
            if old is None or old[0] != key[0] or old[2] != key[2]:
                self.__dict__["_image_epoch"] = self.__dict__.get("_image_epoch", 0) + 1
            with torch.inference_mode(False), torch.no_grad():  # cached across calls: never inference tensors
                images = _ImageSource(self, device, self.__dict__["_image_epoch"])
                if self.precision == "auto":
                    chosen = self._pack_auto(images, device)
                else:
                    chosen = self.effective_precision = self.precision
                self._header, self._header_t, self._image_dev = images.get(chosen)
                if self._status is None or self._status.device != device:
                    self._status = torch.zeros(4, dtype=torch.int32, device=device)
            self._image_key = key
        return self._header, self._image_dev

    #: ``precision = "auto"``: rows of the probe and the bar its two precision modes must agree to -
    #: half the 1e-5 parity bar, because the probe sees 4096 rows and a populate pass a million
    AUTO_PROBE_ROWS = 4096
    AUTO_PROBE_BAR = 5e-6

    #: split-f16 is accepted up to this multiple of float32's OWN response to a 1-ulp change of the
    #: probe inputs (see ``_pack_auto``)
    AUTO_NOISE_FACTOR = 4.0

    def _pack_auto(self, images, device):
        """Precision for ``precision = "auto"``: the first fast mode (``auto_candidates``:
        split-f16, then split-bf16 inside its static envelope) that is MEASURED to agree with the
        exact-float32 kernels on this flow's current weights, float32 otherwise.

        Every time the weights change, a probe of ``AUTO_PROBE_ROWS`` latent rows goes through the
        kernels in both directions (inverse: ``log q`` of the populate path; forward:
        ``log_prob``) and a fast mode is kept only if it agrees with float32 to ``AUTO_PROBE_BAR``
        relative.  Split-bf16 needs that test because its ~2^-16 per product is amplified by the
        flow's conditioning: random flows inside its envelope reach 6e-5 relative at initialisation
        scale and 1e-2 with weights three times larger (``scripts/fuzz_precision.py``).  Split-f16
        products are float32-class by construction (2^-24 relative per operand), so on an
        ill-conditioned flow it differs from the float32 kernel as any two float32 evaluations
        differ (the same sweep: 2e-3 against 7e-3 for the float32 kernel itself, both measured
        against the CPU oracle); it is therefore also accepted within ``AUTO_NOISE_FACTOR`` times
        the float32 kernel's own response to moving every probe input by one ulp - and never when
        the probe left the f16 range.  Costs the images of the probed modes (``_ImageSource``: one
        gather launch each once the flow's structure is known) and four small launches per
        candidate and change of the weights."""
        if self.training:
            self.effective_precision, self.probe_error, self.probe_errors = "f32", None, {}
            return "f32"
        fingerprint = self._fingerprint()
        n, d = self.AUTO_PROBE_ROWS, self.features
        self.probe_errors = {}
        _count(self, "probe")
        chosen = "f32"
        # The probe's own arithmetic is ONE kernel of the library (nsa_probe_compare: float64, max over the
        # rows of the two relative differences); written with torch operators it loaded a dozen of torch's
        # code objects the first time it ran - ~450 ms of the first populate() of a process.
        L = lib.load()
        stream = lib.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        cached = self.__dict__.get("_probe_latents")
        if cached is None or cached.device != device or cached.shape != (n, d):
            # (the same latents for every probe of this flow: drawn on the host and uploaded ONCE - 0.4 ms of
            # every retrain otherwise)
            cached = torch.randn(n, d, generator=torch.Generator().manual_seed(20260930)).to(device)
            self.__dict__["_probe_latents"] = cached
        z = cached
        ctx = None
        if self.context_features:  # a conditional flow is probed at a fixed standard-normal context
            ctx = self.__dict__.get("_probe_context")
            if ctx is None or ctx.device != device or ctx.shape != (n, self.context_features):
                ctx = torch.randn(n, self.context_features, generator=torch.Generator().manual_seed(20261001)).to(device)
                self.__dict__["_probe_context"] = ctx
        status = torch.zeros(4, dtype=torch.int32, device=device)
        errs_dev = torch.zeros(4, dtype=torch.float64, device=device)  # [candidate, one-ulp twin]

        def flow_run(x_in, *args):
            if ctx is not None:
                return torch.ops.nessai_amd.flow_run_ctx(x_in, ctx, *args)
            return torch.ops.nessai_amd.flow_run(x_in, *args)

        def run(precision, z_in, x_fwd=None):
            _, header_t, image_dev = images.get(precision)
            x, ld = flow_run(z_in, image_dev, header_t, DIR_INVERSE, MODE_APPLY, True, status,
                             self.waves_per_group, self.max_groups)
            _, lp = flow_run(x if x_fwd is None else x_fwd, image_dev, header_t, DIR_FORWARD,
                             MODE_LOGPROB, False, status, self.waves_per_group, self.max_groups)
            return x, ld, lp

        def compare(ld, lp, ref, slot):
            """max relative difference of (log|det J| of the inverse pass, measured on log q; log_prob) -> errs_dev[slot]"""
            errs_dev[slot : slot + 1].zero_()
            lib.check(L.nsa_probe_compare(n, d, lib.ptr(z), lib.ptr(ld), lib.ptr(lp), lib.ptr(ref[1]), lib.ptr(ref[2]),
                                          lib.ptr(errs_dev[slot:]), stream), "nsa_probe_compare")

        def nudged(t):
            out = torch.empty_like(t)
            lib.check(L.nsa_probe_nudge(t.numel(), lib.ptr(t), lib.ptr(out), stream), "nsa_probe_nudge")
            return out

        ref = noise = None
        for p in auto_candidates(self):
            if p == "f16x3" and self._f16_blocked == fingerprint:
                continue
            try:
                images.get(p)
            except ValueError:  # a weight beyond the f16 range
                continue
            if ref is None:
                ref = run("f32", z)
            status.zero_()
            _, ld, lp = run(p, z, x_fwd=ref[0])
            compare(ld, lp, ref, 0)
            err, over = float(errs_dev[0].item()), int(status[1].item())
            self.probe_errors[p] = err
            bar = self.AUTO_PROBE_BAR
            if p == "f16x3" and over == 0 and math.isfinite(err) and err > bar:
                if noise is None:  # the float32 kernel against itself, inputs one ulp further out
                    _, ld_n, lp_n = run("f32", nudged(z), x_fwd=nudged(ref[0]))
                    compare(ld_n, lp_n, ref, 1)
                    noise = float(errs_dev[1].item())
                    self.probe_errors["f32_one_ulp"] = noise
                if math.isfinite(noise):
                    bar = max(bar, self.AUTO_NOISE_FACTOR * noise)
            if over == 0 and math.isfinite(err) and err <= bar:
                chosen = p
                break
        self.effective_precision = chosen
        errs = [v for k, v in self.probe_errors.items() if k in pack.PRECISIONS]
        self.probe_error = self.probe_errors.get(chosen, max(errs, default=None))
        return chosen

    def note_f16_overflow(self):
        """A split-f16 call on the current weights left the f16 range: ``"auto"`` stops choosing the mode
        until the weights change; a flow pinned to ``"f16x3"`` raises."""
        if self.precision != "auto":
            raise F16RangeError(
                "a value beyond the f16 range (65520) met a matrix product of the 'f16x3' kernels; "
                "use precision 'auto' (falls back by itself), 'bf16x3' or 'f32' for this flow"
            )
        self._f16_blocked = self._fingerprint()
        self._image_key = None

    # -- kernel calls ----------------------------------------------------------

    def _check_input(self, x):
        if not isinstance(x, torch.Tensor):
            raise TypeError("inputs must be torch tensors")
        if x.device.type != "cuda":
            raise RuntimeError(
                "nessai_amd flows take tensors on a ROCm device (got "
                f"{x.device}); there is no CPU fallback"
            )
        if x.dim() != 2 or x.shape[1] != self.features:
            raise ValueError(
                "Expected inputs of shape (N, {}), got {}".format(self.features, tuple(x.shape))
            )
        self._refuse_float64(x)
        return x.detach().to(torch.float32).contiguous()

    def trainer(self):
        """The flow's ``nessai_amd.train.Trainer`` (flat parameter buffer + layer plan), built on
        first use and rebuilt when the tensors were re-allocated or the permutations re-drawn."""
        from ..train import Trainer

        tr = self.__dict__.get("_trainer")
        if tr is None or not tr.is_current():
            with torch.inference_mode(False), torch.no_grad():
                tr = Trainer(self)
            tr.handle = ops.register_plan(tr)
            self.__dict__["_trainer"] = tr
            self._image_key = None
        return tr

    def _wants_grad(self, x):
        if not torch.is_grad_enabled():
            return False
        return x.requires_grad or any(p.requires_grad for p in self.parameters())

    def _check_context(self, context, n):
        """The ``context`` argument of the B1 methods: ``(n, context_features)`` for a conditional flow (float32 on
        the flow's device; converted like the inputs), ``None`` for any other - anything else raises."""
        if not self.context_features:
            if context is not None:
                raise ValueError("this flow was built without context_features: it takes no context")
            return None
        if context is None:
            raise ValueError(f"a conditional flow (context_features={self.context_features}) needs a context")
        if not isinstance(context, torch.Tensor) or context.device.type != "cuda":
            raise RuntimeError("nessai_amd flows take tensors on a ROCm device; there is no CPU fallback")
        if context.dim() != 2 or context.shape != (n, self.context_features):
            raise ValueError("Expected a context of shape ({}, {}), got {}".format(n, self.context_features,
                                                                                  tuple(context.shape)))
        if context.dtype == torch.float64:
            raise TypeError(self._FLOAT64_REFUSAL.format(what="The context tensor is float64"))
        if context.requires_grad and torch.is_grad_enabled():
            # (the reverse-mode kernels compute the gradient of the parameters and of x only: a context produced by
            # a trainable embedding net would train with a zero gradient and no error otherwise)
            raise RuntimeError(
                "the context of a conditional nessai_amd flow carries no gradient: it was given with "
                "requires_grad=True (an embedding net upstream of the flow would not train); detach it, or "
                "evaluate under torch.no_grad()"
            )
        return context.detach().to(torch.float32).contiguous()

    def _run(self, x, direction, mode, want_x=True, context=None):
        stochastic = self.training and _has_dropout(self)  # training mode draws dropout masks, as nflows does
        inner_bn = self.training and _has_inner_batch_norm(self)  # ... and normalises with batch statistics
        if direction == DIR_FORWARD and (self._wants_grad(x) or stochastic or inner_bn or
                                         (self.training and _has_norm_layers(self))):
            return self._run_differentiable(x, mode, want_x, context)
        if inner_bn:
            raise RuntimeError(
                "batch_norm_within_layers: the inverse / sampling kernels run on the running statistics (eval mode); "
                "call .eval() before inverting or sampling the flow"
            )
        if self.training and _has_norm_layers(self):
            raise RuntimeError(
                "The inverse of batch-norm / act-norm transforms uses running statistics; "
                "call .eval() before inverting or sampling the flow"
            )
        if stochastic:
            raise RuntimeError(
                "dropout_probability > 0: the inverse / sampling kernels have no dropout (it is the identity in eval "
                "mode); call .eval() before inverting or sampling the flow"
            )
        x = self._check_input(x)
        context = self._check_context(context, x.shape[0])
        for _ in range(2):
            header, image = self.device_image()
            if context is not None:
                out, ld = torch.ops.nessai_amd.flow_run_ctx(
                    x, context, image, self._header_t, direction, mode, want_x, self._status, self.waves_per_group,
                    self.max_groups,
                )
            else:
                out, ld = torch.ops.nessai_amd.flow_run(
                    x, image, self._header_t, direction, mode, want_x, self._status, self.waves_per_group,
                    self.max_groups,
                )
            # split-f16 products hold for operands inside the f16 range only: the kernel says when a call
            # left it (one 16-byte read-back; the callers of this API read the results back anyway)
            if self.effective_precision != "f16x3" or x.shape[0] == 0 or int(self._status[1].item()) == 0:
                break
            self.note_f16_overflow()  # raises unless "auto": then the repeat runs in float32
        return (out if want_x else None), ld, self._status

    def _run_differentiable(self, x, mode, want_x, context=None):
        """Forward direction through the float32 training kernels, recorded by autograd."""
        context = self._check_context(context, x.shape[0] if isinstance(x, torch.Tensor) and x.dim() == 2 else 0)
        if not isinstance(x, torch.Tensor) or x.device.type != "cuda":
            raise RuntimeError("nessai_amd flows take tensors on a ROCm device; there is no CPU fallback")
        if x.dim() != 2 or x.shape[1] != self.features:
            raise ValueError("Expected inputs of shape (N, {}), got {}".format(self.features, tuple(x.shape)))
        if x.shape[0] == 0:  # nothing to differentiate: empty outputs, as the inference path returns
            empty = torch.empty(0, dtype=torch.float32, device=x.device)
            z = torch.empty((0, self.features), dtype=torch.float32, device=x.device) if want_x else None
            return z, empty, self._status
        self._refuse_float64(x)
        xin = x.to(torch.float32).contiguous()
        log_prob = mode == MODE_LOGPROB
        z, ld, lp = ops.differentiable_flow(self, xin, want_x, not log_prob, log_prob, context=context)
        if self._status is None:
            with torch.inference_mode(False):
                self._status = torch.zeros(4, dtype=torch.int32, device=x.device)
        else:
            self._status.zero_()
        return z, (lp if log_prob else ld), self._status

    def forward(self, x, context=None):
        """x -> (z, log|det J|)  (``NFlow.forward``, flows/base.py:209-214)."""
        z, ld, _ = self._run(x, DIR_FORWARD, MODE_APPLY, context=context)
        return z, ld

    def inverse(self, z, context=None):
        """z -> (x, log|det J| of the inverse)  (flows/base.py:216-221).

        Raises ``AssertionError`` if the spline inverse met a negative
        discriminant, as the reference's arithmetic does (the caller drops the
        batch, ``flowproposal.py:357-364``).
        """
        x, ld, status = self._run(z, DIR_INVERSE, MODE_APPLY, context=context)
        if int(status[0].item()) != 0:
            raise AssertionError("negative discriminant in the spline inverse")
        return x, ld

    def log_prob(self, x, context=None):
        """log p(x)  (flows/base.py:236-245)."""
        _, lp, _ = self._run(x, DIR_FORWARD, MODE_LOGPROB, want_x=False, context=context)
        return lp

    def forward_and_log_prob(self, x, context=None):
        """(z, log p(x))  (flows/base.py:259-273)."""
        z, lp, _ = self._run(x, DIR_FORWARD, MODE_LOGPROB, context=context)
        return z, lp

    def sample_latent_distribution(self, n, context=None):
        """Draw from the base distribution on the flow's device (flows/base.py:247-250)."""
        if context is not None:
            raise NotImplementedError
        std = float(np.sqrt(self._distribution.var))
        z = torch.randn(int(n), self.features, device=self._device(), dtype=torch.float32)
        return z if std == 1.0 else z * std

    def base_distribution_log_prob(self, z, context=None):
        """log N(z; 0, var I)  (flows/base.py:252-257).

        A D-term reduction per row: evaluated with torch device ops on the
        same stream (the fused populate kernel computes it in-kernel).
        """
        z = self._check_input(z)
        var = float(self._distribution.var)
        log_z = np.float32(0.5 * self.features * np.log(2 * np.pi * var))
        return -(0.5 / var) * torch.sum(z * z, dim=1) - log_z

    def sample(self, num_samples, context=None):
        """(flows/base.py:223-234)."""
        noise = self.sample_latent_distribution(num_samples)
        x, _ = self.inverse(noise, context=context)
        return x

    def sample_and_log_prob(self, N, context=None):
        """(x, log p(x)) for fresh draws  (flows/base.py:275-287)."""
        z = self.sample_latent_distribution(N)
        log_prob = self.base_distribution_log_prob(z)
        x, logabsdet = self.inverse(z, context=context)
        return x, log_prob - logabsdet

    def freeze_transform(self):
        self._transform.requires_grad_(False)

    def unfreeze_transform(self):
        self._transform.requires_grad_(True)


class _ImageSource:
    """Device images of one flow at its CURRENT weights, per precision: ``get(precision)`` returns
    ``(header, header tensor, image tensor)``.

    The first image of a (structure, precision) is packed on the host (``pack.pack_flow``); for a flow of
    couplings and permutations - whose image words are constants or plain (split) copies of parameters -
    every later one is rebuilt ON THE DEVICE from the flat parameter vector by one gather launch
    (``nsa_repack_image`` with the map of ``pack.repack_map``, built the second time the structure is
    seen).  The images and maps live on the flow (``flow._repack``) across changes of the weights; the
    structure key holds what the constants depend on - the device and the permutations - and the entries are
    kept per train / eval mode."""

    def __init__(self, flow, device, epoch):
        self.flow, self.device, self.epoch = flow, device, epoch
        self._desc = None
        self._params = None
        store = flow.__dict__.setdefault("_repack", {})
        # The permutations are identified by (storage address, version counter) of their buffers; the buffers
        # themselves are held with the entries (`_perm_refs`), so that a module that REBINDS its permutation to a
        # new tensor cannot get the old address - and a stale image - back from the allocator.  One entry per
        # train / eval mode: a flow that alternates between the two keeps both (and their device maps).
        perm_buffers = [b for name, b in flow.named_buffers() if name.endswith("_permutation")]
        perms = tuple((b.data_ptr(), b._version) for b in perm_buffers)
        self.structure = (str(device), perms)
        self.mode = bool(flow.training)
        for key in [k for k in store if k != "_perm_refs" and k[0] != self.structure]:
            del store[key]  # other permutations / another device: its images are of no use any more
        store["_perm_refs"] = perm_buffers
        self.store = store

    def desc(self):
        if self._desc is None:
            self._desc = M.describe(self.flow)
        return self._desc

    def params(self):
        """The flat float32 parameter vector on the device (the trainer's own buffer when the Parameters are
        views of it in order - no copy)."""
        if self._params is None:
            ps = list(self.flow.parameters())
            base, at, flat = ps[0].data_ptr(), 0, True
            for p in ps:
                flat = flat and p.dtype == torch.float32 and p.is_contiguous() and p.data_ptr() == base + 4 * at
                at += p.numel()
            trainer = self.flow.__dict__.get("_trainer")
            if flat and trainer is not None and trainer.params.data_ptr() == base and trainer.params.numel() == at:
                self._params = trainer.params
            else:
                self._params = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in ps])
        return self._params

    def get(self, precision):
        key = (self.structure, self.mode, precision)
        entry = self.store.get(key)
        flow = self.flow
        if entry is not None and entry["epoch"] == self.epoch:  # built for these weights (e.g. by the probe)
            return entry["header"], entry["header_t"], entry["image"]
        pure = flow.__dict__.get("_repack_pure")
        if entry is not None and pure is not False and os.environ.get("NESSAI_AMD_DEVICE_REPACK", "1") != "0":
            if "map" not in entry:
                pure = self._build_map(entry, precision)
            if pure is not False and precision == "f16x3":
                # the host pack raises for a weight beyond the f16 range (pack.split_f16: `_pack_auto` relies on
                # it); the device gather would write +-inf halves instead - same rule, checked on the device
                flag = torch.zeros(1, dtype=torch.int32, device=self.device)
                stream = torch.cuda.current_stream(self.device).cuda_stream
                lib.check(lib.load().nsa_probe_range(self.params().numel(), lib.ptr(self.params()), pack.F16_LIMIT,
                                                     lib.ptr(flag), lib.c_void_p(stream)), "nsa_probe_range")
                if int(flag.item()):
                    raise ValueError("precision 'f16x3' needs every weight below 65520 in magnitude")
            if pure is not False:
                pos, e0, e1, kind = entry["map"]
                _count(flow, "device_repack")
                stream = torch.cuda.current_stream(self.device).cuda_stream
                rc = lib.load().nsa_repack_image(
                    pos.numel(), lib.ptr(pos), lib.ptr(e0), lib.ptr(e1), lib.ptr(kind), lib.ptr(self.params()),
                    pack.PRECISIONS[precision], lib.ptr(entry["image"]), lib.c_void_p(stream),
                )
                lib.check(rc, "nsa_repack_image")
                entry["epoch"] = self.epoch
                return entry["header"], entry["header_t"], entry["image"]
        _count(flow, "host_pack")
        image = pack.pack_flow(dict(self.desc(), precision=precision))
        header = (lib.ctypes.c_uint32 * pack.HEADER_WORDS)(*image[: pack.HEADER_WORDS].tolist())
        entry = dict(header=header, header_t=ops.header_tensor(header),
                     image=torch.from_numpy(image.view(np.int32)).to(self.device), epoch=self.epoch)
        self.store[key] = entry
        # `flow._repack_eager` (set by FlowProposal.initialise): the map of the device-side rebuild is computed
        # WITH the first host pack instead of at the second sight of the structure - a sampler retrains the flow
        # and would otherwise meet those two host packs (~100 ms for config 3) inside a populate() of the run
        if (flow.__dict__.get("_repack_eager") and pure is not False
                and os.environ.get("NESSAI_AMD_DEVICE_REPACK", "1") != "0"):
            self._build_map(entry, precision)
        return entry["header"], entry["header_t"], entry["image"]

    def _build_map(self, entry, precision):
        """``entry["map"]`` (``pack.repack_map`` on the device) or ``False`` for a flow whose image words are derived."""
        flow = self.flow
        try:
            pos, e0, e1, kind = pack.repack_map(lambda off: M.describe(flow, index_offset=off), precision)
            entry["map"] = tuple(torch.from_numpy(a).to(self.device) for a in (pos, e0, e1, kind))
            flow.__dict__["_repack_pure"] = True
            return True
        except ValueError:  # LU-linear / batch norm / act norm: derived image words
            flow.__dict__["_repack_pure"] = False
            return False


def _count(flow, what):
    """``flow.image_stats``: how the kernel images of this flow were built so far (``host_pack`` /
    ``device_repack``, ``probe`` = precision probes of ``"auto"``) - read by the profiling scripts."""
    stats = flow.__dict__.setdefault("image_stats", {})
    stats[what] = stats.get(what, 0) + 1


def _has_batch_norm(flow):
    return any(isinstance(m, M.BatchNorm) for m in flow.modules())


def _has_dropout(flow):
    """``dropout_probability`` > 0 in the residual blocks of the conditioners."""
    cached = flow.__dict__.get("_dropout")
    if cached is None:
        cached = any(getattr(m, "dropout_probability", 0.0) > 0.0 for m in flow.modules() if isinstance(m, M.ResidualNet))
        flow.__dict__["_dropout"] = cached
    return cached


def _has_inner_batch_norm(flow):
    """Batch norm inside the residual blocks of the conditioners (``batch_norm_within_layers``)."""
    cached = flow.__dict__.get("_inner_batch_norm")
    if cached is None:
        cached = any(getattr(m, "use_batch_norm", False) for m in flow.modules() if isinstance(m, M.ResidualNet))
        flow.__dict__["_inner_batch_norm"] = cached
    return cached


def _has_norm_layers(flow):
    """Batch norm or a not-yet-initialised act norm: layers whose training-mode pass differs.  The
    act-norm flags live on the device; they are read once per (buffer, version) - not per call, which
    would be a device synchronisation on the inference path of every act-norm flow."""
    cached = flow.__dict__.get("_norm_layers")
    if cached is None:
        cached = [m for m in flow.modules() if isinstance(m, (M.BatchNorm, M.ActNorm))]
        flow.__dict__["_norm_layers"] = cached
    for m in cached:
        if isinstance(m, M.BatchNorm):
            return True
        key = (m.initialized.data_ptr(), m.initialized._version)
        known = m.__dict__.get("_initialized_host")
        if known is None or known[0] != key:
            known = (key, bool(m.initialized))
            m.__dict__["_initialized_host"] = known
        if not known[1]:
            return True
    return False


def lib_supports(ht, transform, bins):
    """Ask the library whether a kernel instance exists (no GPU needed)."""
    kind = pack.TRANSFORMS[transform]
    return bool(lib.load().nsa_flow_supported(int(ht), int(kind), int(bins if transform == "rq" else 0)))


_STRUCT_WORDS = (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 25, 26, 27, 29, 34, 35)  # nsa_format.h ints


@torch.no_grad()
def log_prob_many(flows, x):
    """``(n_flows, N)`` log-densities of one batch under several flows (inference only).

    The loop of ``ImportanceFlowModel.log_prob_all`` (``flowmodel/importance.py:114-129``)
    as ONE grouped launch (``nsa_flow_run_many``) when the flows share their structure -
    they do in the importance sampler, every level is built from one ``flow_config`` -
    and one launch per flow otherwise.
    """
    flows = list(flows)
    if not flows:
        raise ValueError("no flows")
    first = flows[0]
    for f in flows:
        if f.training and _has_batch_norm(f):
            raise RuntimeError("call .eval() before using batch-norm flows for inference")
    x = first._check_input(x)
    n = x.shape[0]
    out = torch.empty((len(flows), n), dtype=torch.float32, device=x.device)
    if n == 0:
        return out
    images = [f.device_image() for f in flows]
    h0 = images[0][0]
    same = all(all(h[w] == h0[w] for w in _STRUCT_WORDS) for h, _ in images)
    if not same or len(flows) == 1:
        for i, f in enumerate(flows):
            out[i] = f.log_prob(x)
        return out
    ptrs = torch.tensor([im.data_ptr() for _, im in images], dtype=torch.int64, device=x.device)
    status = first._status
    status.zero_()
    stream = torch.cuda.current_stream(x.device).cuda_stream
    rc = lib.load().nsa_flow_run_many(
        h0, lib.ptr(ptrs), len(flows), DIR_FORWARD, MODE_LOGPROB, n, lib.ptr(x), None, lib.ptr(out),
        lib.ptr(status), first.waves_per_group, first.max_groups, lib.c_void_p(stream),
    )
    lib.check(rc, "nsa_flow_run_many")
    if first.effective_precision == "f16x3" and int(status[1].item()) != 0:
        # the batch left the f16 range under (at least) one of the flows: all of them again, each on its own
        # (HipFlow._run repeats a flow whose split-f16 launch reports the overflow in float32)
        for i, f in enumerate(flows):
            out[i] = f.log_prob(x)
    return out
