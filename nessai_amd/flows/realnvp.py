"""RealNVP (affine / additive coupling) on the gfx950 kernels.

Drop-in for the reference's ``nessai.flows.RealNVP``
(``src/nessai/flows/realnvp.py:20-214``): same constructor signature and the
same per-layer transform sequence — optional act norm, linear transform
(permutation, by default followed by an identity-initialised LU-linear layer),
affine (or volume-preserving additive) coupling with a sign-alternating mask,
batch norm between layers (on by default).
"""

import numpy as np
import math

import torch

from . import modules as M
from .base import HipFlow
from .nsf import _reject_unsupported
from .utils import create_linear_transform, create_pre_transform


def identify_scale_activation(fn):
    """``(a, b)`` such that ``fn(x) == sigmoid(x + a) + b``.

    The reference passes ``scale_activation`` (a callable on tensors) straight to nflows'
    ``AffineCouplingTransform`` (``src/nessai/flows/realnvp.py:112, 199-203``; exercised with
    ``lambda x: torch.sigmoid(x + 2) + 1e-3`` in ``tests/test_flows/test_specific_flows.py:32``).  A
    kernel cannot call Python, so the callable is probed on a grid and must belong to the family the
    gfx950 affine transform implements - shifted sigmoid plus a floor, of which nflows' default is
    ``a = 2, b = 1e-3``; anything else raises."""
    x = torch.linspace(-12.0, 12.0, 193, dtype=torch.float64)
    try:
        with torch.no_grad():
            y = torch.as_tensor(fn(x), dtype=torch.float64)
            floor = float(torch.as_tensor(fn(torch.tensor([-200.0], dtype=torch.float64)), dtype=torch.float64))
    except Exception as exc:
        raise NotImplementedError(f"scale_activation could not be evaluated on a float64 grid: {exc}") from exc
    if y.shape != x.shape or not torch.isfinite(y).all() or not (0.0 <= floor < 1.0):
        raise NotImplementedError("scale_activation is not of the form sigmoid(x + a) + b (no gfx950 kernel)")
    mid = float(y[96]) - floor  # x = 0
    if not 0.0 < mid < 1.0:
        raise NotImplementedError("scale_activation is not of the form sigmoid(x + a) + b (no gfx950 kernel)")
    shift = math.log(mid / (1.0 - mid))
    if float((torch.sigmoid(x + shift) + floor - y).abs().max()) > 1e-9:
        raise NotImplementedError(
            "scale_activation is not of the form sigmoid(x + a) + b: only that family (nflows' default is "
            "a = 2, b = 1e-3) has a gfx950 kernel"
        )
    return round(shift, 9), round(floor, 12)


class RealNVP(HipFlow):
    def __init__(
        self,
        features,
        hidden_features,
        num_layers,
        num_blocks_per_layer,
        mask=None,
        context_features=None,
        net="resnet",
        use_volume_preserving=False,
        activation="relu",
        dropout_probability=0.0,
        batch_norm_within_layers=False,
        batch_norm_between_layers=True,
        linear_transform="lu",
        pre_transform=None,
        pre_transform_kwargs=None,
        actnorm=False,
        distribution=None,
        **kwargs,
    ):
        if features <= 1:
            raise ValueError(
                "RealNVP requires at least 2 dimensions. "
                f"Specified dimensions: {features}."
            )
        if actnorm and batch_norm_between_layers:
            raise RuntimeError(
                "Cannot enable actnorm and batchnorm between layers simultaneously."
            )
        # (the reference's MLP conditioner ignores both options with a warning: flows/realnvp.py:151-159)
        _reject_unsupported(context_features, dropout_probability if net.lower() != "mlp" else 0.0,
                            batch_norm_within_layers and net.lower() != "mlp")
        scale_kwargs = {}
        scale_activation = kwargs.pop("scale_activation", None)
        if scale_activation is not None and not use_volume_preserving:
            shift, eps = identify_scale_activation(scale_activation)
            scale_kwargs = dict(scale_shift=shift, scale_eps=eps)
        if kwargs:
            raise TypeError(f"Unexpected coupling arguments: {sorted(kwargs)}")

        coupling = M.AdditiveCouplingTransform if use_volume_preserving else M.AffineCouplingTransform

        # +1 transformed / -1 identity, flipped every layer unless given per layer
        if mask is None:
            mask = torch.ones(features)
            mask[::2] = -1
        else:
            mask = np.array(mask)
            if not mask.shape[-1] == features:
                raise ValueError("Mask does not match number of features")
            if mask.ndim == 2 and not mask.shape[0] == num_layers:
                raise ValueError("Mask does not match number of layers")
            mask = torch.from_numpy(mask).type(torch.get_default_dtype())
        if mask.dim() == 1:
            per_layer = torch.empty([num_layers, features])
            for i in range(num_layers):
                per_layer[i] = mask * (-1.0) ** i
            mask = per_layer

        if net.lower() == "resnet":

            def conditioner(n_in, n_out):
                return M.ResidualNet(
                    n_in, n_out, hidden_features, context_features=context_features,
                    num_blocks=num_blocks_per_layer, activation=activation,
                    dropout_probability=dropout_probability, use_batch_norm=batch_norm_within_layers,
                )

        elif net.lower() == "mlp":
            if context_features is not None:  # (the reference's MLP raises at its first call: flows/nets.py:90-101)
                raise NotImplementedError("MLP with conditional inputs is not implemented.")

            def conditioner(n_in, n_out):
                return M.MLP(
                    n_in, n_out, num_blocks_per_layer * [hidden_features], activation=activation
                )

        else:
            raise ValueError(f"Unknown nn type: {net}. Choose from: {{resnet, mlp}}.")

        layers = []
        if pre_transform is not None:
            layers.append(
                create_pre_transform(pre_transform, features, **(pre_transform_kwargs or {}))
            )
        for i in range(num_layers):
            if actnorm:
                layers.append(M.ActNorm(features=features))
            if linear_transform is not None and linear_transform.lower() != "none":
                layers.append(create_linear_transform(linear_transform, features))
            layers.append(coupling(mask=mask[i], transform_net_create_fn=conditioner, **scale_kwargs))
            if batch_norm_between_layers:
                layers.append(M.BatchNorm(features=features))

        if distribution is None:
            distribution = M.StandardNormal([features])
        super().__init__(M.CompositeTransform(layers), distribution)
