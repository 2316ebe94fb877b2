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
#: Override per flow (``flow.precision = ...``) or with ``NESSAI_AMD_PRECISION``.
DEFAULT_PRECISION = os.environ.get("NESSAI_AMD_PRECISION", "f32")


def populate_precision(flow):
    """Precision the fused populate kernel uses when the proposal asks for ``"auto"``:
    split-bf16 inside its validated envelope - rational-quadratic flows with >= 15 features and
    >= 30 hidden units (configs 3, 4 and 5: <= 3.3e-6 relative; a grid of randomised flows
    with 16-32 features: <= 6.6e-6) - and exact float32 otherwise (narrower flows have a small
    |log_prob| and reach 1.4e-5 - 2.8e-5 relative in split-bf16, outside the 1e-5 bar)."""
    if flow.transform_kind == "rq" and flow.hidden_features >= 30 and flow.features >= 15:
        return "bf16x3"
    return "f32"


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
        self._image_key = None
        self._image_host = None
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
        if self.precision not in pack.PRECISIONS and self.precision != "auto":
            raise ValueError(f"unknown precision {self.precision!r}; choose from {sorted(pack.PRECISIONS)} or 'auto'")
        desc = M.describe(self)
        self.features = desc["D"]
        self.transform_kind = desc["transform"]
        self.hidden_features = desc["hidden"]
        ht = pack.hidden_tiles(desc["hidden"])
        if ht > 4 or not lib_supports(ht, desc["transform"], desc.get("bins", 0)):
            raise ValueError(
                "No gfx950 kernel instance for this flow configuration "
                f"(hidden={desc['hidden']}, transform={desc['transform']}, "
                f"bins={desc.get('bins', 0)})"
            )
        # every other structural limit of the kernels (LU-linear beyond 64 features, mixed conditioner
        # depths, ...) is known to the packer: pack once now so that an unsupported configuration raises
        # at construction like the checks above, not at the first call
        pack.pack_flow(dict(desc, precision=self.precision if self.precision != "auto" else "f32"))

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
            desc = M.describe(self)
            if self.precision == "auto":
                image = self._pack_auto(desc, device)
            else:
                image = pack.pack_flow(dict(desc, precision=self.precision))
                self.effective_precision = self.precision
            self._image_host = image
            self._header = (lib.ctypes.c_uint32 * pack.HEADER_WORDS)(*image[: pack.HEADER_WORDS].tolist())
            with torch.inference_mode(False):  # cached across calls: must not be inference tensors
                self._header_t = ops.header_tensor(self._header)
                self._image_dev = torch.from_numpy(image.view(np.int32)).to(device)
                self._status = torch.zeros(4, dtype=torch.int32, device=device)
            self._image_key = key
        return self._header, self._image_dev

    #: ``precision = "auto"``: rows of the probe and the bar its two precision modes must agree to -
    #: half the 1e-5 parity bar, because the probe sees 4096 rows and a populate pass a million
    AUTO_PROBE_ROWS = 4096
    AUTO_PROBE_BAR = 5e-6

    def _pack_auto(self, desc, device):
        """Packed image for ``precision = "auto"``: split-bf16 where it is MEASURED to agree with the
        exact-float32 kernels on this flow's current weights, float32 otherwise.

        The static envelope (``populate_precision``) only says where split-bf16 is worth trying: its
        error grows with the conditioning of the flow - random flows inside the envelope reach 6e-5
        relative at initialisation scale and 1e-3 with weights three times larger
        (``scripts/fuzz_bf16x3.py``) - so every time the weights change, a probe of
        ``AUTO_PROBE_ROWS`` latent rows goes through both kernels in both directions (inverse:
        ``log q`` of the populate path; forward: ``log_prob``) and split-bf16 is kept only if they
        agree to ``AUTO_PROBE_BAR`` relative.  Costs one extra pack and four small launches per
        change of the weights."""
        if populate_precision(self) != "bf16x3" or self.training:
            self.effective_precision, self.probe_error = "f32", None
            return pack.pack_flow(dict(desc, precision="f32"))
        images = {p: pack.pack_flow(dict(desc, precision=p)) for p in ("f32", "bf16x3")}
        n, d = self.AUTO_PROBE_ROWS, self.features
        with torch.inference_mode(False), torch.no_grad():
            z = torch.randn(n, d, generator=torch.Generator().manual_seed(20260930)).to(device)
            status = torch.zeros(4, dtype=torch.int32, device=device)
            out = {}
            for p, image in images.items():
                header = (lib.ctypes.c_uint32 * pack.HEADER_WORDS)(*image[: pack.HEADER_WORDS].tolist())
                header_t = ops.header_tensor(header)
                image_dev = torch.from_numpy(image.view(np.int32)).to(device)
                x, ld = torch.ops.nessai_amd.flow_run(z, image_dev, header_t, DIR_INVERSE, MODE_APPLY, True, status,
                                                      self.waves_per_group, self.max_groups)
                out[p] = (x, ld, image_dev, header_t)
            x32, ld32 = out["f32"][0], out["f32"][1].double()
            log_pz = -0.5 * (z.double() ** 2).sum(dim=1) - 0.5 * d * math.log(2.0 * math.pi)
            log_q32 = log_pz - ld32
            err = ((out["bf16x3"][1].double() - ld32).abs() / log_q32.abs().clamp_min(1e-3)).max()
            lp = {}
            for p in images:
                _, lp[p] = torch.ops.nessai_amd.flow_run(x32, out[p][2], out[p][3], DIR_FORWARD, MODE_LOGPROB, False,
                                                         status, self.waves_per_group, self.max_groups)
            lp32 = lp["f32"].double()
            err = torch.maximum(err, ((lp["bf16x3"].double() - lp32).abs() / lp32.abs().clamp_min(1e-3)).max())
            err = float(err)
        ok = math.isfinite(err) and err <= self.AUTO_PROBE_BAR
        self.effective_precision, self.probe_error = ("bf16x3" if ok else "f32"), err
        return images[self.effective_precision]

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

    def _run(self, x, direction, mode, want_x=True):
        if direction == DIR_FORWARD and (self._wants_grad(x) or (self.training and _has_norm_layers(self))):
            return self._run_differentiable(x, mode, want_x)
        if self.training and _has_norm_layers(self):
            raise RuntimeError(
                "The inverse of batch-norm / act-norm transforms uses running statistics; "
                "call .eval() before inverting or sampling the flow"
            )
        x = self._check_input(x)
        header, image = self.device_image()
        out, ld = torch.ops.nessai_amd.flow_run(
            x, image, self._header_t, direction, mode, want_x, self._status, self.waves_per_group,
            self.max_groups,
        )
        return (out if want_x else None), ld, self._status

    def _run_differentiable(self, x, mode, want_x):
        """Forward direction through the float32 training kernels, recorded by autograd."""
        if not isinstance(x, torch.Tensor) or x.device.type != "cuda":
            raise RuntimeError("nessai_amd flows take tensors on a ROCm device; there is no CPU fallback")
        if x.dim() != 2 or x.shape[1] != self.features:
            raise ValueError("Expected inputs of shape (N, {}), got {}".format(self.features, tuple(x.shape)))
        if x.shape[0] == 0:
            raise ValueError("cannot differentiate through an empty batch")
        xin = x.to(torch.float32).contiguous()
        log_prob = mode == MODE_LOGPROB
        z, ld, lp = ops.differentiable_flow(self, xin, want_x, not log_prob, log_prob)
        if self._status is None:
            with torch.inference_mode(False):
                self._status = torch.zeros(4, dtype=torch.int32, device=x.device)
        else:
            self._status.zero_()
        return z, (lp if log_prob else ld), self._status

    def forward(self, x, context=None):
        """x -> (z, log|det J|)  (``NFlow.forward``, flows/base.py:209-214)."""
        _no_context(context)
        z, ld, _ = self._run(x, DIR_FORWARD, MODE_APPLY)
        return z, ld

    def inverse(self, z, context=None):
        """z -> (x, log|det J| of the inverse)  (flows/base.py:216-221).

        Raises ``AssertionError`` if the spline inverse met a negative
        discriminant, as the reference's arithmetic does (the caller drops the
        batch, ``flowproposal.py:357-364``).
        """
        _no_context(context)
        x, ld, status = self._run(z, DIR_INVERSE, MODE_APPLY)
        if int(status[0].item()) != 0:
            raise AssertionError("negative discriminant in the spline inverse")
        return x, ld

    def log_prob(self, x, context=None):
        """log p(x)  (flows/base.py:236-245)."""
        _no_context(context)
        _, lp, _ = self._run(x, DIR_FORWARD, MODE_LOGPROB, want_x=False)
        return lp

    def forward_and_log_prob(self, x, context=None):
        """(z, log p(x))  (flows/base.py:259-273)."""
        _no_context(context)
        z, lp, _ = self._run(x, DIR_FORWARD, MODE_LOGPROB)
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
        _no_context(context)
        noise = self.sample_latent_distribution(num_samples)
        x, _ = self.inverse(noise)
        return x

    def sample_and_log_prob(self, N, context=None):
        """(x, log p(x)) for fresh draws  (flows/base.py:275-287)."""
        _no_context(context)
        z = self.sample_latent_distribution(N)
        log_prob = self.base_distribution_log_prob(z)
        x, logabsdet = self.inverse(z)
        return x, log_prob - logabsdet

    def freeze_transform(self):
        self._transform.requires_grad_(False)

    def unfreeze_transform(self):
        self._transform.requires_grad_(True)


def _no_context(context):
    if context is not None:
        raise NotImplementedError("conditional flows have no gfx950 kernel yet")


def _has_batch_norm(flow):
    return any(isinstance(m, M.BatchNorm) for m in flow.modules())


def _has_norm_layers(flow):
    """Batch norm or a not-yet-initialised act norm: layers whose training-mode pass differs."""
    cached = flow.__dict__.get("_norm_layers")
    if cached is None:
        cached = [m for m in flow.modules() if isinstance(m, (M.BatchNorm, M.ActNorm))]
        flow.__dict__["_norm_layers"] = cached
    return any(isinstance(m, M.BatchNorm) or not bool(m.initialized) for m in cached)


def lib_supports(ht, transform, bins):
    """Ask the library whether a kernel instance exists (no GPU needed)."""
    kind = pack.TRANSFORMS[transform]
    return bool(lib.load().nsa_flow_supported(int(ht), int(kind), int(bins if transform == "rq" else 0)))


_STRUCT_WORDS = (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 25, 26, 27, 29)  # nsa_format.h ints


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
    return out
