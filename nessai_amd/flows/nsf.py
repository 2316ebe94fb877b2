"""Neural spline flow (rational-quadratic coupling) on the gfx950 kernels.

Drop-in for the reference's ``nessai.flows.NeuralSplineFlow``
(``src/nessai/flows/nsf.py:20-130``): same constructor signature, same
transform sequence per coupling layer — linear transform (random permutation,
optionally followed by an LU-linear layer), RQ-spline coupling with an
alternating binary mask (even-indexed features transformed on even layers),
optional batch norm — and a standard-normal base distribution.
"""

import torch

from . import modules as M
from .base import HipFlow
from .utils import create_linear_transform


def alternating_binary_mask(features, even=True):
    """0/1 mask, ones mark transformed features (``nsf.py:99-101``)."""
    mask = torch.zeros(features, dtype=torch.uint8)
    mask[(0 if even else 1) :: 2] = 1
    return mask


class NeuralSplineFlow(HipFlow):
    def __init__(
        self,
        features,
        hidden_features,
        num_layers,
        num_blocks_per_layer,
        num_bins=8,
        context_features=None,
        activation="relu",
        dropout_probability=0.0,
        batch_norm_within_layers=False,
        batch_norm_between_layers=False,
        apply_unconditional_transform=False,
        linear_transform="permutation",
        tails="linear",
        tail_bound=5.0,
        distribution=None,
        **kwargs,
    ):
        if features <= 1:
            raise ValueError(
                "Coupling based Neural Spline flow requires at least 2 "
                f"dimensions. Specified dimensions: {features}."
            )
        _reject_unsupported(
            context_features, dropout_probability, batch_norm_within_layers,
            apply_unconditional_transform,
        )

        def conditioner(n_in, n_out):
            return M.ResidualNet(
                n_in, n_out, hidden_features, context_features=context_features,
                num_blocks=num_blocks_per_layer, activation=activation,
                dropout_probability=dropout_probability, use_batch_norm=batch_norm_within_layers,
            )

        layers = []
        for i in range(num_layers):
            if linear_transform is not None and linear_transform.lower() != "none":
                layers.append(create_linear_transform(linear_transform, features))
            layers.append(
                M.PiecewiseRationalQuadraticCouplingTransform(
                    mask=alternating_binary_mask(features, even=(i % 2 == 0)),
                    transform_net_create_fn=conditioner,
                    num_bins=num_bins,
                    tails=tails,
                    tail_bound=tail_bound,
                    **kwargs,
                )
            )
            if batch_norm_between_layers:
                layers.append(M.BatchNorm(features=features))

        if distribution is None:
            distribution = M.StandardNormal([features])
        super().__init__(M.CompositeTransform(layers), distribution)


def _reject_unsupported(context_features, dropout_probability, batch_norm_within_layers,
                        apply_unconditional_transform=False):
    if context_features is not None and (isinstance(context_features, bool) or int(context_features) < 1):
        raise ValueError("context_features must be a positive integer or None")
    if dropout_probability and not 0.0 <= float(dropout_probability) < 1.0:
        raise ValueError("dropout_probability must lie in [0, 1)")
    # (batch_norm_within_layers: eval mode on the kernels - the running statistics as per-feature affines;
    # nessai_amd.train.Trainer refuses such a flow, like one with dropout)
    if apply_unconditional_transform:
        raise NotImplementedError("apply_unconditional_transform has no gfx950 kernel")
