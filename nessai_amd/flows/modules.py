"""Parameter containers for the transforms of a flow.

These ``torch.nn.Module`` classes hold the parameters and buffers of each
transform under the same attribute names the reference's module tree uses
(``_transform._transforms.N.transform_net.initial_layer.weight`` ...; the tree
is built by ``src/nessai/flows/realnvp.py:186-213`` / ``nsf.py:110-128`` out of
glasflow.nflows classes), so ``state_dict()`` / ``load_state_dict()`` /
``parameters()`` / ``apply()`` behave as ``FlowModel`` expects
(``src/nessai/flowmodel/base.py:636,685,711,729,766-769``).

They contain **no arithmetic**: the maths runs in the gfx950 kernels, fed by
``describe()`` which turns each container into an entry of the flow
description consumed by ``nessai_amd.pack.pack_flow``.  Initial values follow
the published nflows initialisers so that the same ``torch.manual_seed`` gives
the same starting weights as the reference.
"""

import math

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F
from torch.nn import init

ACTIVATION_NAMES = {}


def _register_activations():
    import torch.nn.functional as F

    ACTIVATION_NAMES[F.relu] = "relu"
    ACTIVATION_NAMES[torch.relu] = "relu"
    ACTIVATION_NAMES[F.tanh] = "tanh"
    ACTIVATION_NAMES[torch.tanh] = "tanh"
    ACTIVATION_NAMES[F.silu] = "silu"


_register_activations()


def activation_name(activation):
    """Map the callable the reference passes (``flows/utils.py:200-205``) to a name."""
    if isinstance(activation, str):
        name = activation.lower()
        if name == "swish":
            name = "silu"
        if name not in ("relu", "tanh", "silu"):
            raise ValueError(f"Unknown activation: {activation}")
        return name
    if activation in ACTIVATION_NAMES:
        return ACTIVATION_NAMES[activation]
    name = getattr(activation, "__name__", "")
    if name in ("silu", "swish"):
        return "silu"
    if name in ("relu", "tanh"):
        return name
    raise ValueError(
        f"Activation {activation!r} has no gfx950 kernel; choose relu, tanh or silu"
    )


class _Container(nn.Module):
    """Base: holding parameters only — calling it is a bug."""

    def forward(self, *args, **kwargs):
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; the arithmetic runs "
            "in the nessai_amd HIP kernels (call the flow, not its parts)"
        )


class Linear(nn.Linear):
    """``nn.Linear`` used purely as a (weight, bias) holder with torch's default init."""

    def forward(self, *args, **kwargs):
        raise RuntimeError("Linear layers of a HIP flow are parameter containers")


class ResidualBlock(_Container):
    """``h + W2 act(W1 act(h) + b1) + b2``; with ``context_features`` the block's output is gated,
    ``h + glu(cat(temps, context_layer(context)))`` (nflows; creation order of the layers kept, so that one
    ``torch.manual_seed`` gives the reference's initial weights)."""

    def __init__(self, features, context_features=None, zero_initialization=True, use_batch_norm=False):
        super().__init__()
        self.use_batch_norm = bool(use_batch_norm)
        if use_batch_norm:  # nflows: two BatchNorm1d(features, eps=1e-3); only their eval-mode affine has a kernel
            self.batch_norm_layers = nn.ModuleList([nn.BatchNorm1d(features, eps=1e-3) for _ in range(2)])
        if context_features is not None:
            self.context_layer = Linear(context_features, features)
        self.linear_layers = nn.ModuleList([Linear(features, features) for _ in range(2)])
        if zero_initialization:
            init.uniform_(self.linear_layers[-1].weight, -1e-3, 1e-3)
            init.uniform_(self.linear_layers[-1].bias, -1e-3, 1e-3)

    def reset_parameters(self):
        if hasattr(self, "context_layer"):
            self.context_layer.reset_parameters()
        if self.use_batch_norm:
            for bn in self.batch_norm_layers:
                bn.reset_parameters()
        for layer in self.linear_layers:
            layer.reset_parameters()
        init.uniform_(self.linear_layers[-1].weight, -1e-3, 1e-3)
        init.uniform_(self.linear_layers[-1].bias, -1e-3, 1e-3)


def batch_norm_affines(block):
    """Eval-mode ``(a, c)`` of the two batch norms of a residual block, ``bn(h) = a h + c`` (float64)."""
    out = []
    for bn in block.batch_norm_layers:
        var = bn.running_var.detach().double().cpu()
        a = bn.weight.detach().double().cpu() / torch.sqrt(var + bn.eps)
        out.append((a.numpy(), (bn.bias.detach().double().cpu() - a * bn.running_mean.detach().double().cpu()).numpy()))
    return out


class ResidualNet(_Container):
    """Conditioner: initial layer, pre-activation residual blocks, final layer."""

    kind = "resnet"

    def __init__(self, in_features, out_features, hidden_features, context_features=None, num_blocks=2,
                 activation="relu", dropout_probability=0.0, use_batch_norm=False):
        super().__init__()
        self.activation_name = activation_name(activation)
        self.hidden_features = hidden_features
        self.context_features = context_features
        #: batch norm within the blocks (``batch_norm_within_layers``): eval mode only, like dropout - the inference
        #: kernels apply the running statistics, the training kernels refuse it
        self.use_batch_norm = bool(use_batch_norm)
        #: dropout between the two layers of a block: the identity in eval mode (every inference kernel); the
        #: training kernels have no dropout and refuse a non-zero rate (nessai_amd.train.Trainer)
        self.dropout_probability = float(dropout_probability or 0.0)
        # (conditional: the context joins the inputs of the initial layer and gates every block)
        self.initial_layer = Linear(in_features + (context_features or 0), hidden_features)
        self.blocks = nn.ModuleList([ResidualBlock(hidden_features, context_features, use_batch_norm=use_batch_norm)
                                     for _ in range(num_blocks)])
        self.final_layer = Linear(hidden_features, out_features)

    def layers(self):
        out = [self.initial_layer]
        for block in self.blocks:
            out.extend(block.linear_layers)
        out.append(self.final_layer)
        return out


class MLP(_Container):
    """Plain MLP conditioner (attribute names of ``src/nessai/flows/nets.py:55-67``)."""

    kind = "mlp"

    def __init__(self, in_features, out_features, hidden_sizes, activation="relu"):
        super().__init__()
        self.activation_name = activation_name(activation)
        if len(hidden_sizes) == 0:
            raise ValueError("List of hidden sizes can't be empty.")
        if len(set(hidden_sizes)) != 1:
            raise ValueError("The gfx950 kernels need equal hidden layer widths")
        self._hidden_sizes = list(hidden_sizes)
        self._input_layer = Linear(in_features, hidden_sizes[0])
        self._hidden_layers = nn.ModuleList(
            [Linear(i, o) for i, o in zip(hidden_sizes[:-1], hidden_sizes[1:])]
        )
        self._output_layer = Linear(hidden_sizes[-1], out_features)

    def layers(self):
        return [self._input_layer, *self._hidden_layers, self._output_layer]


class RandomPermutation(_Container):
    def __init__(self, features):
        super().__init__()
        self.register_buffer("_permutation", torch.randperm(features))


class LULinear(_Container):
    """``W = L U`` parameters (unit-diagonal L, ``diag(U) = softplus(.) + eps``)."""

    def __init__(self, features, identity_init=True, eps=1e-3):
        super().__init__()
        self.features = features
        self.eps = eps
        self.bias = nn.Parameter(torch.zeros(features))
        n_tri = ((features - 1) * features) // 2
        self.lower_entries = nn.Parameter(torch.zeros(n_tri))
        self.upper_entries = nn.Parameter(torch.zeros(n_tri))
        self.unconstrained_upper_diag = nn.Parameter(torch.zeros(features))
        self._initialize(identity_init)

    def _initialize(self, identity_init=True):
        init.zeros_(self.bias)
        if identity_init:
            init.zeros_(self.lower_entries)
            init.zeros_(self.upper_entries)
            init.constant_(
                self.unconstrained_upper_diag, math.log(math.exp(1 - self.eps) - 1)
            )
        else:
            stdv = 1.0 / math.sqrt(self.features)
            init.uniform_(self.lower_entries, -stdv, stdv)
            init.uniform_(self.upper_entries, -stdv, stdv)
            init.uniform_(self.unconstrained_upper_diag, -stdv, stdv)

    def host_matrices(self):
        """(W, W^-1, bias, log|det W|) as float32 arrays.

        Built the way the reference's eval-mode cache is (lower @ upper, two
        triangular solves against the identity); a D x D host-side
        precomputation done once per weight update, like that cache.
        """
        with torch.no_grad():
            f = self.features
            dev = self.lower_entries.device
            lower = torch.zeros(f, f, dtype=torch.float32)
            upper = torch.zeros(f, f, dtype=torch.float32)
            li = np.tril_indices(f, k=-1)
            ui = np.triu_indices(f, k=1)
            di = np.diag_indices(f)
            lower[li[0], li[1]] = self.lower_entries.detach().float().cpu()
            lower[di[0], di[1]] = 1.0
            upper[ui[0], ui[1]] = self.upper_entries.detach().float().cpu()
            diag = (
                torch.nn.functional.softplus(self.unconstrained_upper_diag.detach().float().cpu())
                + self.eps
            )
            upper[di[0], di[1]] = diag
            weight = lower @ upper
            identity = torch.eye(f, f, dtype=torch.float32)
            lower_inv = torch.linalg.solve_triangular(
                lower, identity, upper=False, unitriangular=True
            )
            weight_inv = torch.linalg.solve_triangular(
                upper, lower_inv, upper=True, unitriangular=False
            )
            logabsdet = torch.sum(torch.log(diag))
            del dev
            return (
                weight.numpy(),
                weight_inv.numpy(),
                self.bias.detach().float().cpu().numpy(),
                np.float32(logabsdet.item()),
            )


class HouseholderSequence(_Container):
    """The reflection vectors of an orthogonal factor of ``SVDLinear`` (parameter container; nflows' attribute
    name and identity initialisation: pairs of equal unit vectors)."""

    def __init__(self, features, num_transforms):
        super().__init__()
        self.features = features
        self.num_transforms = num_transforms
        qr = torch.eye(num_transforms // 2, features).repeat_interleave(2, dim=0)
        if num_transforms % 2 != 0:
            qr = torch.cat((qr, torch.zeros(1, features)))
            qr[-1, num_transforms // 2] = 1
        self.q_vectors = nn.Parameter(qr)

    def host_matrix(self):
        """Q (float64) with ``x -> x Q`` the sequence of reflections ``x (I - 2 q q^T / |q|^2)`` in order."""
        q = self.q_vectors.detach().double().cpu().numpy()
        Q = np.eye(self.features)
        for v in q:
            Q = Q - np.outer(Q @ v, (2.0 / float(v @ v)) * v)
        return Q


class SVDLinear(_Container):
    """``linear_transform="svd"`` (``src/nessai/flows/utils.py:316-323``): ``y = x Q2 diag(s) Q1 + b`` with
    Householder-sequence factors and ``s = eps + softplus(.)``.  In eval mode it is one dense D x D matrix with a
    known log-determinant - the LU-linear op of the kernels (``host_matrices``, same contract as ``LULinear``);
    there is no training kernel for it (``train`` raises)."""

    def __init__(self, features, num_householder=10, identity_init=True, eps=1e-3):
        super().__init__()
        assert num_householder % 2 == 0
        self.features = features
        self.eps = eps
        self.bias = nn.Parameter(torch.zeros(features))
        self.orthogonal_1 = HouseholderSequence(features, num_householder)
        self.unconstrained_diagonal = nn.Parameter(torch.zeros(features))
        self.orthogonal_2 = HouseholderSequence(features, num_householder)
        self._initialize(identity_init)

    def _initialize(self, identity_init=True):
        init.zeros_(self.bias)
        if identity_init:
            init.constant_(self.unconstrained_diagonal, math.log(math.exp(1 - self.eps) - 1))
        else:
            stdv = 1.0 / math.sqrt(self.features)
            init.uniform_(self.unconstrained_diagonal, -stdv, stdv)

    def host_matrices(self):
        """(W, W^-1, bias, log|det W|) as float32 arrays, ``y = W x + b``: float64 on the host, rounded once
        (the reference applies the 2 x 10 reflections to every batch in float32)."""
        with torch.no_grad():
            diag = (F.softplus(self.unconstrained_diagonal.detach().float().cpu()) + self.eps).double().numpy()
            q1, q2 = self.orthogonal_1.host_matrix(), self.orthogonal_2.host_matrix()
            w_t = q2 @ np.diag(diag) @ q1            # y = x W^T + b
            w_inv_t = q1.T @ np.diag(1.0 / diag) @ q2.T
            logabsdet = float(np.sum(np.log(diag)))
            return (np.ascontiguousarray(w_t.T).astype(np.float32), np.ascontiguousarray(w_inv_t.T).astype(np.float32),
                    self.bias.detach().float().cpu().numpy(), np.float32(logabsdet))


class BatchNorm(_Container):
    """Batch norm between coupling layers (eval-mode statistics on the device)."""

    def __init__(self, features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.momentum = momentum
        self.eps = eps
        constant = math.log(math.exp(1 - eps) - 1)
        self.unconstrained_weight = nn.Parameter(constant * torch.ones(features))
        self.bias = nn.Parameter(torch.zeros(features))
        self.register_buffer("running_mean", torch.zeros(features))
        self.register_buffer("running_var", torch.ones(features))

    def reset_parameters(self):
        """Initial values (the constants ``nessai.flows.utils.reset_weights`` writes for nflows' BatchNorm,
        ``flows/utils.py:265-272``)."""
        with torch.no_grad():
            self.unconstrained_weight.fill_(math.log(math.exp(1 - self.eps) - 1))
            self.bias.zero_()
            self.running_mean.zero_()
            self.running_var.fill_(1)

    def host_tables(self):
        """Per-feature ``((x - a) / b) * c + d`` tables for both directions."""
        with torch.no_grad():
            w = (
                torch.nn.functional.softplus(self.unconstrained_weight.detach().float().cpu())
                + self.eps
            )
            b = self.bias.detach().float().cpu()
            mean = self.running_mean.detach().float().cpu()
            var = self.running_var.detach().float().cpu()
            std = torch.sqrt(var + self.eps)
            ld_fwd = torch.sum(torch.log(w) - 0.5 * torch.log(var + self.eps))
            ld_inv = torch.sum(-torch.log(w) + 0.5 * torch.log(var + self.eps))
            fwd = (mean.numpy(), std.numpy(), w.numpy(), b.numpy(), np.float32(ld_fwd.item()))
            inv = (b.numpy(), w.numpy(), std.numpy(), mean.numpy(), np.float32(ld_inv.item()))
            return fwd, inv


class ActNorm(_Container):
    def __init__(self, features):
        super().__init__()
        self.register_buffer("initialized", torch.tensor(False, dtype=torch.bool))
        self.log_scale = nn.Parameter(torch.zeros(features))
        self.shift = nn.Parameter(torch.zeros(features))

    def host_tables(self):
        with torch.no_grad():
            ls = self.log_scale.detach().float().cpu()
            scale = torch.exp(ls)
            shift = self.shift.detach().float().cpu()
            zeros = torch.zeros_like(scale)
            ones = torch.ones_like(scale)
            s = torch.sum(ls)
            fwd = (zeros.numpy(), ones.numpy(), scale.numpy(), shift.numpy(), np.float32(s.item()))
            inv = (shift.numpy(), scale.numpy(), ones.numpy(), zeros.numpy(), np.float32(-s.item()))
            return fwd, inv


class CouplingTransform(_Container):
    """Mask bookkeeping + conditioner; ``mask <= 0`` identity, ``> 0`` transformed."""

    transform = None  # "rq" | "affine" | "additive"

    def __init__(self, mask, transform_net_create_fn, params_per_feature):
        super().__init__()
        mask = torch.as_tensor(mask)
        if mask.dim() != 1:
            raise ValueError("Mask must be a 1-dim tensor.")
        if mask.numel() <= 0:
            raise ValueError("Mask can't be empty.")
        self.features = len(mask)
        features_vector = torch.arange(self.features)
        self.register_buffer("identity_features", features_vector.masked_select(mask <= 0))
        self.register_buffer("transform_features", features_vector.masked_select(mask > 0))
        self.transform_net = transform_net_create_fn(
            len(self.identity_features), len(self.transform_features) * params_per_feature
        )


class AffineCouplingTransform(CouplingTransform):
    """``scale = sigmoid(raw + scale_shift) + scale_eps`` (2 and 1e-3: nflows' DEFAULT_SCALE_ACTIVATION)."""

    transform = "affine"

    def __init__(self, mask, transform_net_create_fn, scale_shift=2.0, scale_eps=1e-3):
        super().__init__(mask, transform_net_create_fn, 2)
        self.scale_shift = float(scale_shift)
        self.scale_eps = float(scale_eps)


class AdditiveCouplingTransform(CouplingTransform):
    transform = "additive"

    def __init__(self, mask, transform_net_create_fn):
        super().__init__(mask, transform_net_create_fn, 1)


class PiecewiseRationalQuadraticCouplingTransform(CouplingTransform):
    transform = "rq"

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails="linear",
                 tail_bound=1.0, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3):
        if tails != "linear":
            raise ValueError("The gfx950 spline kernel implements linear tails only")
        if min_bin_width * num_bins > 1.0:
            raise ValueError("Minimal bin width too large for the number of bins")
        if min_bin_height * num_bins > 1.0:
            raise ValueError("Minimal bin height too large for the number of bins")
        self.num_bins = num_bins
        self.tails = tails
        self.tail_bound = tail_bound
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        super().__init__(mask, transform_net_create_fn, 3 * num_bins - 1)


class CompositeTransform(_Container):
    def __init__(self, transforms):
        super().__init__()
        self._transforms = nn.ModuleList(transforms)


class StandardNormal(_Container):
    """Base distribution N(0, I); no parameters."""

    var = 1.0

    def __init__(self, shape):
        super().__init__()
        self._shape = torch.Size(shape)


def flatten_transforms(transform):
    """Depth-first list of leaf transforms in forward order."""
    if isinstance(transform, CompositeTransform) or hasattr(transform, "_transforms"):
        out = []
        for t in transform._transforms:
            out.extend(flatten_transforms(t))
        return out
    return [transform]


def _np(t):
    return t.detach().to(torch.float32).cpu().numpy()


def describe(flow, index_offset=None):
    """Module tree -> flow description for ``nessai_amd.pack.pack_flow``.

    Duck-typed on attribute names, so any module tree with the reference's
    structure can be described.

    ``index_offset``: describe the flow's STRUCTURE instead - every conditioner weight / bias array holds
    ``index_offset +`` the positions of its elements in the flat vector of ``flow.parameters()`` (float32),
    for ``nessai_amd.pack.repack_map``; a flow with LU-linear / batch-norm / act-norm transforms, whose image
    words are not plain copies of parameters, raises ``ValueError``.
    """
    offsets = None
    if index_offset is not None:
        offsets, at = {}, 0
        for p in flow.parameters():
            offsets[id(p)] = at
            at += p.numel()

    def values(p):
        if offsets is None:
            return _np(p)
        return (np.arange(p.numel(), dtype=np.float64).reshape(tuple(p.shape)) + (offsets[id(p)] + index_offset)).astype(np.float32)

    leaves = flatten_transforms(flow._transform)
    desc = dict(transforms=[])
    coupling = None
    for t in leaves:
        name = type(t).__name__
        if hasattr(t, "_permutation"):
            desc["transforms"].append(("permutation", t._permutation.detach().cpu().numpy()))
        elif hasattr(t, "lower_entries") or hasattr(t, "orthogonal_1"):
            if offsets is not None:
                raise ValueError("the image words of an LU- / SVD-linear transform are not copies of parameters")
            if hasattr(t, "host_matrices"):
                W, W_inv, bias, ld = t.host_matrices()
            else:  # structurally identical foreign module (tests)
                with torch.no_grad():
                    W = _np(t.weight())
                    W_inv = _np(t.weight_inverse())
                    bias = _np(t.bias)
                    ld = np.float32(t.logabsdet().item())
            desc["transforms"].append(("lu", W, W_inv, bias, ld))
        elif hasattr(t, "transform_net"):
            net = t.transform_net
            if hasattr(net, "initial_layer"):
                kind = "resnet"
                layers = [net.initial_layer]
                for block in net.blocks:
                    layers.extend(block.linear_layers)
                layers.append(net.final_layer)
                hidden = net.initial_layer.out_features
                scale_logits = hasattr(net, "hidden_features")
            elif hasattr(net, "_input_layer"):
                kind = "mlp"
                layers = [net._input_layer, *net._hidden_layers, net._output_layer]
                hidden = net._input_layer.out_features
                scale_logits = hasattr(net, "hidden_features")
            else:
                raise ValueError(f"Unknown conditioner network {type(net).__name__}")
            if "Rational" in name:
                transform = "rq"
            elif "Additive" in name:
                transform = "additive"
            elif "Affine" in name:
                transform = "affine"
            else:
                raise ValueError(f"Unknown coupling transform {name}")
            if hasattr(net, "activation_name"):
                act = net.activation_name
            elif kind == "resnet":
                act = activation_name(net.blocks[0].activation) if len(net.blocks) else "relu"
            else:
                act = activation_name(net._activation)
            this = dict(
                D=int(len(t.identity_features) + len(t.transform_features)),
                hidden=hidden,
                net=kind,
                transform=transform,
                scale_logits=scale_logits,
                activation=act,
            )
            if transform == "affine":
                this.update(affine_shift=float(getattr(t, "scale_shift", 2.0)),
                            affine_eps=float(getattr(t, "scale_eps", 1e-3)))
            if transform == "rq":
                this.update(
                    bins=t.num_bins,
                    tail_bound=float(t.tail_bound),
                    min_bin_width=float(t.min_bin_width),
                    min_bin_height=float(t.min_bin_height),
                    min_derivative=float(t.min_derivative),
                )
            # conditional flows (context_features): the context layers of the residual blocks, in block order
            gates = []
            n_ctx = getattr(net, "context_features", None)
            if n_ctx:
                if kind != "resnet":
                    raise ValueError("only the residual conditioner takes a context")
                gates = [(values(b.context_layer.weight), values(b.context_layer.bias)) for b in net.blocks]
                this["context_features"] = int(n_ctx)
            # batch norm within the blocks (eval mode): the first one stays a per-feature affine in front of the
            # block's first activation, the second one is folded into the block's first linear layer
            block_norms = []
            weights = [(values(l.weight), values(l.bias)) for l in layers]
            if kind == "resnet" and any(getattr(b, "use_batch_norm", False) for b in net.blocks):
                if offsets is not None:
                    raise ValueError("the image words of a conditioner with batch norm are not copies of parameters")
                this["batch_norm_within"] = True
                for b, block in enumerate(net.blocks):
                    (a0, c0), (a1, c1) = batch_norm_affines(block)
                    block_norms.append((a0.astype(np.float32), c0.astype(np.float32)))
                    w, bias = weights[1 + 2 * b]
                    weights[1 + 2 * b] = ((a1[:, None] * np.asarray(w, dtype=np.float64)).astype(np.float32),
                                          (a1 * np.asarray(bias, dtype=np.float64) + c1).astype(np.float32))
            if coupling is None:
                coupling = this
            elif coupling != this:
                raise ValueError("All coupling layers of a flow must share one configuration")
            desc["transforms"].append(
                (
                    "coupling",
                    t.identity_features.detach().cpu().numpy().tolist(),
                    t.transform_features.detach().cpu().numpy().tolist(),
                    weights,
                    gates,
                    block_norms,
                )
            )
        elif hasattr(t, "running_var") or hasattr(t, "log_scale"):
            if offsets is not None:
                raise ValueError("the image words of a batch-norm / act-norm transform are not copies of parameters")
            if hasattr(t, "host_tables"):
                fwd, inv = t.host_tables()
            elif hasattr(t, "running_var"):
                fwd, inv = BatchNorm.host_tables(t)
            else:
                fwd, inv = ActNorm.host_tables(t)
            desc["transforms"].append(("elementwise", fwd, inv))
        else:
            raise ValueError(f"Transform {name} has no gfx950 kernel")
    if coupling is None:
        raise ValueError("The flow has no coupling transform")
    desc.update(coupling)
    dist = getattr(flow, "_distribution", None)
    desc["base_var"] = float(getattr(dist, "_var", getattr(dist, "var", 1.0)))
    return desc
