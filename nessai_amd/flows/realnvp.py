"""RealNVP (affine / additive coupling) on the gfx950 kernels.

Drop-in for the reference's ``nessai.flows.RealNVP``
(``src/nessai/flows/realnvp.py:20-214``): same constructor signature and the
same per-layer transform sequence — optional act norm, linear transform
(permutation, by default followed by an identity-initialised LU-linear layer),
affine (or volume-preserving additive) coupling with a sign-alternating mask,
batch norm between layers (on by default).
"""

import numpy as np
import torch

from . import modules as M
from .base import HipFlow
from .nsf import _reject_unsupported
from .utils import create_linear_transform, create_pre_transform


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
        _reject_unsupported(context_features, dropout_probability and net.lower() != "mlp",
                            batch_norm_within_layers and net.lower() != "mlp")
        if "scale_activation" in kwargs and kwargs["scale_activation"] is not None:
            raise NotImplementedError(
                "custom scale_activation has no gfx950 kernel (default: sigmoid(x + 2) + 1e-3)"
            )
        kwargs.pop("scale_activation", None)
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
                    n_in, n_out, hidden_features, num_blocks=num_blocks_per_layer,
                    activation=activation,
                )

        elif net.lower() == "mlp":

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
            layers.append(coupling(mask=mask[i], transform_net_create_fn=conditioner))
            if batch_norm_between_layers:
                layers.append(M.BatchNorm(features=features))

        if distribution is None:
            distribution = M.StandardNormal([features])
        super().__init__(M.CompositeTransform(layers), distribution)
