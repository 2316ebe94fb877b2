"""Invertible transforms used by the reference's RealNVP / NSF, CPU restatement.

Test infrastructure only (see ``oracle/__init__.py``).  Restated from the
public nflows API; the reference call sites that fix which pieces are needed
and with which arguments:

(Algorithm and structure: nflows, MIT License, (c) 2020 Durkan, Bekasov, Murray, Papamakarios - see
``oracle/nflows_restated/__init__.py`` and ``LICENSE.nflows``.)

* ``CompositeTransform`` — ``src/nessai/flows/realnvp.py:211``, ``nsf.py:128``
* ``RandomPermutation`` / ``LULinear(identity_init=True, using_cache=True)``
  — ``src/nessai/flows/utils.py:305-315``; ``SVDLinear(num_householder=10, identity_init=True)`` with its
  ``HouseholderSequence`` factors — ``flows/utils.py:316-323``
* ``BatchNorm`` — ``realnvp.py:205-206``, ``nsf.py:121-122``; attribute names
  pinned by ``flows/utils.py:265-272``
* ``AffineCouplingTransform`` / ``AdditiveCouplingTransform`` —
  ``realnvp.py:109-112,199-203``
* ``PiecewiseRationalQuadraticCouplingTransform`` — ``nsf.py:97-108``
* ``normalization.ActNorm`` — ``realnvp.py:187-190``

Module/attribute names follow nflows so that ``state_dict`` keys look like
those a reference ``model.pt`` would hold (``_transform._transforms.N....``).
"""

import warnings

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F
from torch.nn import init

from . import splines
from .utils import sum_except_batch


class InverseNotAvailable(Exception):
    """Raised when a transform has no inverse in its current mode."""


class InputOutsideDomain(Exception):
    """Raised when inputs fall outside a transform's domain."""


class Transform(nn.Module):
    """Base class: ``forward`` returns ``(outputs, logabsdet)``."""

    def forward(self, inputs, context=None):
        raise NotImplementedError()

    def inverse(self, inputs, context=None):
        raise InverseNotAvailable()


class CompositeTransform(Transform):
    """Apply transforms in order (forward) or reversed with inverses."""

    def __init__(self, transforms):
        super().__init__()
        self._transforms = nn.ModuleList(transforms)

    @staticmethod
    def _cascade(inputs, funcs, context):
        batch_size = inputs.shape[0]
        outputs = inputs
        total_logabsdet = inputs.new_zeros(batch_size)
        for func in funcs:
            outputs, logabsdet = func(outputs, context)
            total_logabsdet += logabsdet
        return outputs, total_logabsdet

    def forward(self, inputs, context=None):
        funcs = self._transforms
        return self._cascade(inputs, funcs, context)

    def inverse(self, inputs, context=None):
        funcs = (transform.inverse for transform in self._transforms[::-1])
        return self._cascade(inputs, funcs, context)


# --------------------------------------------------------------------------
# Permutations
# --------------------------------------------------------------------------


class Permutation(Transform):
    """Column gather with a fixed index buffer; log-det 0."""

    def __init__(self, permutation, dim=1):
        if permutation.ndimension() != 1:
            raise ValueError("Permutation must be a 1D tensor.")
        super().__init__()
        self._dim = dim
        self.register_buffer("_permutation", permutation)

    @property
    def _inverse_permutation(self):
        return torch.argsort(self._permutation)

    @staticmethod
    def _permute(inputs, permutation, dim):
        if dim >= inputs.ndimension():
            raise ValueError("No dimension {} in inputs.".format(dim))
        if inputs.shape[dim] != len(permutation):
            raise ValueError(
                "Dimension {} in inputs must be of size {}.".format(
                    dim, len(permutation)
                )
            )
        batch_size = inputs.shape[0]
        outputs = torch.index_select(inputs, dim, permutation)
        logabsdet = inputs.new_zeros(batch_size)
        return outputs, logabsdet

    def forward(self, inputs, context=None):
        return self._permute(inputs, self._permutation, self._dim)

    def inverse(self, inputs, context=None):
        return self._permute(inputs, self._inverse_permutation, self._dim)


class RandomPermutation(Permutation):
    def __init__(self, features, dim=1):
        super().__init__(torch.randperm(features), dim)


class ReversePermutation(Permutation):
    def __init__(self, features, dim=1):
        super().__init__(torch.arange(features - 1, -1, -1), dim)


# --------------------------------------------------------------------------
# Linear (LU) transform
# --------------------------------------------------------------------------


class LinearCache:
    """Weight, inverse and log-det cached while in eval mode."""

    def __init__(self):
        self.weight = None
        self.inverse = None
        self.logabsdet = None

    def invalidate(self):
        self.weight = None
        self.inverse = None
        self.logabsdet = None


class Linear(Transform):
    """``y = W x + b`` with an eval-mode cache (flushed by ``train()``).

    The cache is why the reference toggles ``model.train(); model.eval()``
    after training (``src/nessai/flowmodel/base.py:681-682``).
    """

    def __init__(self, features, using_cache=False):
        super().__init__()
        self.features = features
        self.bias = nn.Parameter(torch.zeros(features))
        self.using_cache = using_cache
        self.cache = LinearCache()

    def forward(self, inputs, context=None):
        if not self.training and self.using_cache:
            self._check_forward_cache()
            outputs = F.linear(inputs, self.cache.weight, self.bias)
            logabsdet = self.cache.logabsdet * outputs.new_ones(outputs.shape[0])
            return outputs, logabsdet
        return self.forward_no_cache(inputs)

    def _check_forward_cache(self):
        if self.cache.weight is None and self.cache.logabsdet is None:
            self.cache.weight, self.cache.logabsdet = self.weight_and_logabsdet()
        elif self.cache.weight is None:
            self.cache.weight = self.weight()
        elif self.cache.logabsdet is None:
            self.cache.logabsdet = self.logabsdet()

    def inverse(self, inputs, context=None):
        if not self.training and self.using_cache:
            self._check_inverse_cache()
            outputs = F.linear(inputs - self.bias, self.cache.inverse)
            logabsdet = (-self.cache.logabsdet) * outputs.new_ones(outputs.shape[0])
            return outputs, logabsdet
        return self.inverse_no_cache(inputs)

    def _check_inverse_cache(self):
        if self.cache.inverse is None and self.cache.logabsdet is None:
            (
                self.cache.inverse,
                self.cache.logabsdet,
            ) = self.weight_inverse_and_logabsdet()
        elif self.cache.inverse is None:
            self.cache.inverse = self.weight_inverse()
        elif self.cache.logabsdet is None:
            self.cache.logabsdet = self.logabsdet()

    def train(self, mode=True):
        if mode:
            self.cache.invalidate()
        return super().train(mode)

    def use_cache(self, mode=True):
        self.using_cache = mode

    def weight_and_logabsdet(self):
        return self.weight(), self.logabsdet()

    def weight_inverse_and_logabsdet(self):
        return self.weight_inverse(), self.logabsdet()


class LULinear(Linear):
    """``W = L U`` with unit-diagonal ``L`` and ``diag(U) = softplus(.) + eps``."""

    def __init__(self, features, using_cache=False, identity_init=True, eps=1e-3):
        super().__init__(features, using_cache)
        self.eps = eps
        self.lower_indices = np.tril_indices(features, k=-1)
        self.upper_indices = np.triu_indices(features, k=1)
        self.diag_indices = np.diag_indices(features)

        n_triangular_entries = ((features - 1) * features) // 2
        self.lower_entries = nn.Parameter(torch.zeros(n_triangular_entries))
        self.upper_entries = nn.Parameter(torch.zeros(n_triangular_entries))
        self.unconstrained_upper_diag = nn.Parameter(torch.zeros(features))
        self._initialize(identity_init)

    def _initialize(self, identity_init):
        init.zeros_(self.bias)
        if identity_init:
            init.zeros_(self.lower_entries)
            init.zeros_(self.upper_entries)
            constant = np.log(np.exp(1 - self.eps) - 1)
            init.constant_(self.unconstrained_upper_diag, constant)
        else:
            stdv = 1.0 / np.sqrt(self.features)
            init.uniform_(self.lower_entries, -stdv, stdv)
            init.uniform_(self.upper_entries, -stdv, stdv)
            init.uniform_(self.unconstrained_upper_diag, -stdv, stdv)

    def _create_lower_upper(self):
        lower = self.lower_entries.new_zeros(self.features, self.features)
        lower[self.lower_indices[0], self.lower_indices[1]] = self.lower_entries
        lower[self.diag_indices[0], self.diag_indices[1]] = 1.0

        upper = self.upper_entries.new_zeros(self.features, self.features)
        upper[self.upper_indices[0], self.upper_indices[1]] = self.upper_entries
        upper[self.diag_indices[0], self.diag_indices[1]] = self.upper_diag
        return lower, upper

    def forward_no_cache(self, inputs):
        lower, upper = self._create_lower_upper()
        outputs = F.linear(inputs, upper)
        outputs = F.linear(outputs, lower, self.bias)
        logabsdet = self.logabsdet() * inputs.new_ones(outputs.shape[0])
        return outputs, logabsdet

    def inverse_no_cache(self, inputs):
        lower, upper = self._create_lower_upper()
        outputs = inputs - self.bias
        outputs = torch.linalg.solve_triangular(
            lower, outputs.t(), upper=False, unitriangular=True
        )
        outputs = torch.linalg.solve_triangular(
            upper, outputs, upper=True, unitriangular=False
        )
        outputs = outputs.t()
        logabsdet = -self.logabsdet()
        logabsdet = logabsdet * inputs.new_ones(outputs.shape[0])
        return outputs, logabsdet

    def weight(self):
        lower, upper = self._create_lower_upper()
        return lower @ upper

    def weight_inverse(self):
        lower, upper = self._create_lower_upper()
        identity = torch.eye(self.features, self.features, dtype=lower.dtype)
        lower_inverse = torch.linalg.solve_triangular(
            lower, identity, upper=False, unitriangular=True
        )
        weight_inverse = torch.linalg.solve_triangular(
            upper, lower_inverse, upper=True, unitriangular=False
        )
        return weight_inverse

    @property
    def upper_diag(self):
        return F.softplus(self.unconstrained_upper_diag) + self.eps

    def logabsdet(self):
        return torch.sum(torch.log(self.upper_diag))


# --------------------------------------------------------------------------
# Normalisation transforms
# --------------------------------------------------------------------------


class HouseholderSequence(Transform):
    """A product of Householder reflections (published nflows, ``nflows/transforms/orthogonal.py``): a row vector x
    goes to ``x (I - 2 q q^T / |q|^2)`` for each vector q in turn; the inverse applies them in reverse order.
    Initialised to pairs of equal unit vectors, i.e. the identity."""

    def __init__(self, features, num_transforms):
        super().__init__()
        self.features = features
        self.num_transforms = num_transforms
        qr = torch.eye(num_transforms // 2, features).repeat_interleave(2, dim=0)
        if num_transforms % 2 != 0:  # odd number of transforms, including 1
            qr = torch.cat((qr, torch.zeros(1, features)))
            qr[-1, num_transforms // 2] = 1
        self.q_vectors = nn.Parameter(qr)

    @staticmethod
    def _apply_transforms(inputs, q_vectors):
        squared_norms = torch.sum(q_vectors ** 2, dim=-1)
        outputs = inputs
        for q_vector, squared_norm in zip(q_vectors, squared_norms):
            temp = outputs @ q_vector  # inner product
            temp = torch.outer(temp, (2.0 / squared_norm) * q_vector)
            outputs = outputs - temp
        return outputs, inputs.new_zeros(inputs.shape[0])

    def forward(self, inputs, context=None):
        return self._apply_transforms(inputs, self.q_vectors)

    def inverse(self, inputs, context=None):
        reverse_idx = torch.arange(self.num_transforms - 1, -1, -1)
        return self._apply_transforms(inputs, self.q_vectors[reverse_idx])


class SVDLinear(Linear):
    """``W = U S V^T`` with Householder-sequence U, V^T and a positive diagonal (published nflows,
    ``nflows/transforms/svd.py``; the reference: ``src/nessai/flows/utils.py:316-323``, 10 reflections each)."""

    def __init__(self, features, num_householder, using_cache=False, identity_init=True, eps=1e-3):
        super().__init__(features, using_cache)
        assert num_householder % 2 == 0
        self.eps = eps
        self.orthogonal_1 = HouseholderSequence(features=features, num_transforms=num_householder)
        self.unconstrained_diagonal = nn.Parameter(torch.zeros(features))
        self.orthogonal_2 = HouseholderSequence(features=features, num_transforms=num_householder)
        self.identity_init = identity_init
        self._initialize()

    @property
    def diagonal(self):
        return self.eps + F.softplus(self.unconstrained_diagonal)

    @property
    def log_diagonal(self):
        return torch.log(self.diagonal)

    def _initialize(self):
        init.zeros_(self.bias)
        if self.identity_init:
            init.constant_(self.unconstrained_diagonal, np.log(np.exp(1 - self.eps) - 1))
        else:
            stdv = 1.0 / np.sqrt(self.features)
            init.uniform_(self.unconstrained_diagonal, -stdv, stdv)

    def forward_no_cache(self, inputs):
        outputs, _ = self.orthogonal_2(inputs)
        outputs = outputs * self.diagonal
        outputs, _ = self.orthogonal_1(outputs)
        outputs = outputs + self.bias
        return outputs, self.logabsdet() * outputs.new_ones(outputs.shape[0])

    def inverse_no_cache(self, inputs):
        outputs = inputs - self.bias
        outputs, _ = self.orthogonal_1.inverse(outputs)
        outputs = outputs / self.diagonal
        outputs, _ = self.orthogonal_2.inverse(outputs)
        return outputs, -self.logabsdet() * outputs.new_ones(outputs.shape[0])

    def weight(self):
        weight, _ = self.orthogonal_2.inverse(torch.diag(self.diagonal))
        weight, _ = self.orthogonal_1(weight.t())
        return weight.t()

    def weight_inverse(self):
        weight_inv, _ = self.orthogonal_1(torch.diag(torch.reciprocal(self.diagonal)))
        weight_inv, _ = self.orthogonal_2.inverse(weight_inv.t())
        return weight_inv.t()

    def logabsdet(self):
        return torch.sum(self.log_diagonal)


class BatchNorm(Transform):
    """Batch norm as an invertible transform (inverse only in eval mode)."""

    def __init__(self, features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__()
        self.momentum = momentum
        self.eps = eps
        constant = np.log(np.exp(1 - eps) - 1)
        self.unconstrained_weight = nn.Parameter(constant * torch.ones(features))
        self.bias = nn.Parameter(torch.zeros(features))
        self.register_buffer("running_mean", torch.zeros(features))
        self.register_buffer("running_var", torch.ones(features))

    @property
    def weight(self):
        return F.softplus(self.unconstrained_weight) + self.eps

    def forward(self, inputs, context=None):
        if inputs.dim() != 2:
            raise ValueError(
                "Expected 2-dim inputs, got inputs of shape: {}".format(inputs.shape)
            )
        if self.training:
            mean, var = inputs.mean(0), inputs.var(0)
            self.running_mean.mul_(1 - self.momentum).add_(
                mean.detach() * self.momentum
            )
            self.running_var.mul_(1 - self.momentum).add_(
                var.detach() * self.momentum
            )
        else:
            mean, var = self.running_mean, self.running_var

        outputs = (
            self.weight * ((inputs - mean) / torch.sqrt((var + self.eps)))
            + self.bias
        )
        logabsdet_ = torch.log(self.weight) - 0.5 * torch.log(var + self.eps)
        logabsdet = torch.sum(logabsdet_) * inputs.new_ones(inputs.shape[0])
        return outputs, logabsdet

    def inverse(self, inputs, context=None):
        if self.training:
            raise InverseNotAvailable(
                "Batch norm inverse is only available in eval mode, not in "
                "training mode."
            )
        if inputs.dim() != 2:
            raise ValueError(
                "Expected 2-dim inputs, got inputs of shape: {}".format(inputs.shape)
            )
        outputs = (
            torch.sqrt(self.running_var + self.eps)
            * ((inputs - self.bias) / self.weight)
            + self.running_mean
        )
        logabsdet_ = -torch.log(self.weight) + 0.5 * torch.log(
            self.running_var + self.eps
        )
        logabsdet = torch.sum(logabsdet_) * inputs.new_ones(inputs.shape[0])
        return outputs, logabsdet


class ActNorm(Transform):
    """Per-feature affine with data-dependent initialisation (Glow)."""

    def __init__(self, features):
        super().__init__()
        self.register_buffer("initialized", torch.tensor(False, dtype=torch.bool))
        self.log_scale = nn.Parameter(torch.zeros(features))
        self.shift = nn.Parameter(torch.zeros(features))

    @property
    def scale(self):
        return torch.exp(self.log_scale)

    def forward(self, inputs, context=None):
        if inputs.dim() != 2:
            raise ValueError("Expecting inputs to be a 2D tensor.")
        if self.training and not self.initialized:
            self._initialize(inputs)
        scale, shift = self.scale.view(1, -1), self.shift.view(1, -1)
        outputs = scale * inputs + shift
        logabsdet = torch.sum(self.log_scale) * outputs.new_ones(inputs.shape[0])
        return outputs, logabsdet

    def inverse(self, inputs, context=None):
        if inputs.dim() != 2:
            raise ValueError("Expecting inputs to be a 2D tensor.")
        scale, shift = self.scale.view(1, -1), self.shift.view(1, -1)
        outputs = (inputs - shift) / scale
        logabsdet = -torch.sum(self.log_scale) * outputs.new_ones(inputs.shape[0])
        return outputs, logabsdet

    def _initialize(self, inputs):
        with torch.no_grad():
            std = inputs.std(dim=0)
            mu = (inputs / std).mean(dim=0)
            self.log_scale.data = -torch.log(std)
            self.shift.data = -mu
            self.initialized.data = torch.tensor(True, dtype=torch.bool)


class _Normalization:
    """Namespace standing in for ``nflows.transforms.normalization``."""

    BatchNorm = BatchNorm
    ActNorm = ActNorm


normalization = _Normalization()


# --------------------------------------------------------------------------
# Coupling transforms
# --------------------------------------------------------------------------


class CouplingTransform(Transform):
    """Split features by mask; transform one half conditioned on the other.

    ``mask <= 0`` marks identity features, ``mask > 0`` transformed features
    (SURVEY.md §8 a2).
    """

    def __init__(self, mask, transform_net_create_fn, unconditional_transform=None):
        mask = torch.as_tensor(mask)
        if mask.dim() != 1:
            raise ValueError("Mask must be a 1-dim tensor.")
        if mask.numel() <= 0:
            raise ValueError("Mask can't be empty.")

        super().__init__()
        self.features = len(mask)
        features_vector = torch.arange(self.features)

        self.register_buffer(
            "identity_features", features_vector.masked_select(mask <= 0)
        )
        self.register_buffer(
            "transform_features", features_vector.masked_select(mask > 0)
        )
        assert self.num_identity_features + self.num_transform_features == self.features

        self.transform_net = transform_net_create_fn(
            self.num_identity_features,
            self.num_transform_features * self._transform_dim_multiplier(),
        )
        if unconditional_transform is None:
            self.unconditional_transform = None
        else:
            self.unconditional_transform = unconditional_transform(
                features=self.num_identity_features
            )

    @property
    def num_identity_features(self):
        return len(self.identity_features)

    @property
    def num_transform_features(self):
        return len(self.transform_features)

    def forward(self, inputs, context=None):
        if inputs.dim() not in [2, 4]:
            raise ValueError("Inputs must be a 2D or a 4D tensor.")
        if inputs.shape[1] != self.features:
            raise ValueError(
                "Expected features = {}, got {}.".format(self.features, inputs.shape[1])
            )

        identity_split = inputs[:, self.identity_features, ...]
        transform_split = inputs[:, self.transform_features, ...]

        transform_params = self.transform_net(identity_split, context)
        transform_split, logabsdet = self._coupling_transform_forward(
            inputs=transform_split, transform_params=transform_params
        )

        if self.unconditional_transform is not None:
            identity_split, logabsdet_identity = self.unconditional_transform(
                identity_split, context
            )
            logabsdet += logabsdet_identity

        outputs = torch.empty_like(inputs)
        outputs[:, self.identity_features, ...] = identity_split
        outputs[:, self.transform_features, ...] = transform_split
        return outputs, logabsdet

    def inverse(self, inputs, context=None):
        if inputs.dim() not in [2, 4]:
            raise ValueError("Inputs must be a 2D or a 4D tensor.")
        if inputs.shape[1] != self.features:
            raise ValueError(
                "Expected features = {}, got {}.".format(self.features, inputs.shape[1])
            )

        identity_split = inputs[:, self.identity_features, ...]
        transform_split = inputs[:, self.transform_features, ...]

        logabsdet = 0.0
        if self.unconditional_transform is not None:
            identity_split, logabsdet = self.unconditional_transform.inverse(
                identity_split, context
            )

        transform_params = self.transform_net(identity_split, context)
        transform_split, logabsdet_split = self._coupling_transform_inverse(
            inputs=transform_split, transform_params=transform_params
        )
        logabsdet += logabsdet_split

        outputs = torch.empty_like(inputs)
        outputs[:, self.identity_features] = identity_split
        outputs[:, self.transform_features] = transform_split
        return outputs, logabsdet

    def _transform_dim_multiplier(self):
        raise NotImplementedError()

    def _coupling_transform_forward(self, inputs, transform_params):
        raise NotImplementedError()

    def _coupling_transform_inverse(self, inputs, transform_params):
        raise NotImplementedError()


class AffineCouplingTransform(CouplingTransform):
    """``y = x * scale + shift`` with ``scale = sigmoid(s + 2) + 1e-3``.

    Conditioner output layout: shift = first ``d_tr`` columns, raw scale =
    last ``d_tr`` columns.  ``scale_activation`` is a
    constructor kwarg (``tests/test_flows/test_specific_flows.py:32``).
    """

    DEFAULT_SCALE_ACTIVATION = staticmethod(lambda x: torch.sigmoid(x + 2) + 1e-3)
    GENERAL_SCALE_ACTIVATION = staticmethod(lambda x: (F.softplus(x) + 1e-3).clamp(0, 3))

    def __init__(
        self,
        mask,
        transform_net_create_fn,
        unconditional_transform=None,
        scale_activation=None,
    ):
        if scale_activation is None:
            scale_activation = self.DEFAULT_SCALE_ACTIVATION
        self.scale_activation = scale_activation
        super().__init__(mask, transform_net_create_fn, unconditional_transform)

    def _transform_dim_multiplier(self):
        return 2

    def _scale_and_shift(self, transform_params):
        unconstrained_scale = transform_params[:, self.num_transform_features :, ...]
        shift = transform_params[:, : self.num_transform_features, ...]
        scale = self.scale_activation(unconstrained_scale)
        return scale, shift

    def _coupling_transform_forward(self, inputs, transform_params):
        scale, shift = self._scale_and_shift(transform_params)
        log_scale = torch.log(scale)
        outputs = inputs * scale + shift
        logabsdet = sum_except_batch(log_scale, num_batch_dims=1)
        return outputs, logabsdet

    def _coupling_transform_inverse(self, inputs, transform_params):
        scale, shift = self._scale_and_shift(transform_params)
        log_scale = torch.log(scale)
        outputs = (inputs - shift) / scale
        logabsdet = -sum_except_batch(log_scale, num_batch_dims=1)
        return outputs, logabsdet


class AdditiveCouplingTransform(AffineCouplingTransform):
    """Volume-preserving coupling: shift only (``realnvp.py:109-110``)."""

    def _transform_dim_multiplier(self):
        return 1

    def _scale_and_shift(self, transform_params):
        shift = transform_params
        scale = torch.ones_like(shift)
        return scale, shift


class PiecewiseRationalQuadraticCouplingTransform(CouplingTransform):
    """Element-wise monotonic RQ spline on the transformed half.

    Conditioner output is reshaped to ``(N, d_tr, 3K-1)`` (linear tails) and
    split ``[widths(K) | heights(K) | derivatives(K-1)]``; width/height logits
    are divided by ``sqrt(hidden_features)`` when the conditioner exposes that
    attribute (ResidualNet does) — SURVEY.md §8 a5.
    """

    def __init__(
        self,
        mask,
        transform_net_create_fn,
        num_bins=10,
        tails=None,
        tail_bound=1.0,
        apply_unconditional_transform=False,
        img_shape=None,
        min_bin_width=splines.DEFAULT_MIN_BIN_WIDTH,
        min_bin_height=splines.DEFAULT_MIN_BIN_HEIGHT,
        min_derivative=splines.DEFAULT_MIN_DERIVATIVE,
    ):
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.tail_bound = tail_bound

        if apply_unconditional_transform:
            raise NotImplementedError(
                "apply_unconditional_transform is outside the restated scope"
            )
        super().__init__(mask, transform_net_create_fn, unconditional_transform=None)

    def _transform_dim_multiplier(self):
        if self.tails == "linear":
            return self.num_bins * 3 - 1
        return self.num_bins * 3 + 1

    def _coupling_transform_forward(self, inputs, transform_params):
        return self._coupling_transform(inputs, transform_params, inverse=False)

    def _coupling_transform_inverse(self, inputs, transform_params):
        return self._coupling_transform(inputs, transform_params, inverse=True)

    def _coupling_transform(self, inputs, transform_params, inverse=False):
        if inputs.dim() != 2:
            raise ValueError("Inputs must be a 2D tensor.")
        b, d = inputs.shape
        transform_params = transform_params.reshape(b, d, -1)
        outputs, logabsdet = self._piecewise_cdf(inputs, transform_params, inverse)
        return outputs, sum_except_batch(logabsdet)

    def _piecewise_cdf(self, inputs, transform_params, inverse=False):
        unnormalized_widths = transform_params[..., : self.num_bins]
        unnormalized_heights = transform_params[..., self.num_bins : 2 * self.num_bins]
        unnormalized_derivatives = transform_params[..., 2 * self.num_bins :]

        if hasattr(self.transform_net, "hidden_features"):
            unnormalized_widths = unnormalized_widths / np.sqrt(
                self.transform_net.hidden_features
            )
            unnormalized_heights = unnormalized_heights / np.sqrt(
                self.transform_net.hidden_features
            )
        elif hasattr(self.transform_net, "hidden_channels"):
            unnormalized_widths = unnormalized_widths / np.sqrt(
                self.transform_net.hidden_channels
            )
            unnormalized_heights = unnormalized_heights / np.sqrt(
                self.transform_net.hidden_channels
            )
        else:
            warnings.warn(
                "Inputs to the softmax are not scaled down: initialization "
                "might be bad."
            )

        if self.tails is None:
            spline_fn = splines.rational_quadratic_spline
            spline_kwargs = {}
        else:
            spline_fn = splines.unconstrained_rational_quadratic_spline
            spline_kwargs = {"tails": self.tails, "tail_bound": self.tail_bound}

        return spline_fn(
            inputs=inputs,
            unnormalized_widths=unnormalized_widths,
            unnormalized_heights=unnormalized_heights,
            unnormalized_derivatives=unnormalized_derivatives,
            inverse=inverse,
            min_bin_width=self.min_bin_width,
            min_bin_height=self.min_bin_height,
            min_derivative=self.min_derivative,
            **spline_kwargs,
        )
