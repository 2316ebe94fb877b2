"""CPU restatement of the reference's flow assembly (``nessai.flows``).

Test infrastructure only (see ``oracle/__init__.py``).  Each piece cites the
reference code it follows; the arithmetic underneath comes from
``oracle.nflows_restated``.  ``tests/test_golden.py`` checks this assembly
against the reference's own classes (same seed → same ``state_dict`` keys,
same permutations/masks, same outputs) using the fixtures produced by
``tests/golden/make_golden.py``.
"""

import copy

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .nflows_restated import transforms
from .nflows_restated.distributions import Distribution, StandardNormal
from .nflows_restated.nets import ResidualNet
from .nflows_restated.utils import create_alternating_binary_mask


class NFlow(nn.Module):
    """Flow = transform + base distribution.

    Follows ``src/nessai/flows/base.py:170-316`` (``NFlow``) and the abstract
    API of ``BaseFlow`` (``base.py:11-167``).
    """

    device = None

    def __init__(self, transform, distribution):
        super().__init__()
        if not isinstance(transform, transforms.Transform):
            raise TypeError("transform must inherit from Transform")
        if not isinstance(distribution, Distribution):
            raise TypeError("distribution must inherit from Distribution")
        self._transform = transform
        self._distribution = distribution

    def to(self, device):  # base.py:20-23 (returns None)
        self.device = device
        super().to(device)

    def forward(self, x, context=None):  # base.py:209-214
        return self._transform.forward(x, context=context)

    def inverse(self, z, context=None):  # base.py:216-221
        return self._transform.inverse(z, context=context)

    def sample(self, num_samples, context=None):  # base.py:223-234
        noise = self._distribution.sample(num_samples)
        samples, _ = self._transform.inverse(noise, context=context)
        return samples

    def log_prob(self, inputs, context=None):  # base.py:236-245
        noise, logabsdet = self._transform(inputs, context=context)
        log_prob = self._distribution.log_prob(noise)
        return log_prob + logabsdet

    def sample_latent_distribution(self, n, context=None):  # base.py:247-250
        if context is not None:
            raise NotImplementedError
        return self._distribution.sample(n)

    def base_distribution_log_prob(self, z, context=None):  # base.py:252-257
        return self._distribution.log_prob(z)

    def forward_and_log_prob(self, x, context=None):  # base.py:259-273
        z, log_J = self.forward(x, context=context)
        log_prob = self.base_distribution_log_prob(z)
        return z, log_prob + log_J

    def sample_and_log_prob(self, N, context=None):  # base.py:275-287
        z, log_prob = self._distribution.sample_and_log_prob(N)
        samples, logabsdet = self._transform.inverse(z, context=context)
        return samples, log_prob - logabsdet

    def finalise(self):
        pass

    def end_iteration(self):
        pass

    def freeze_transform(self):  # base.py:310-312
        self._transform.requires_grad_(False)

    def unfreeze_transform(self):  # base.py:314-316
        self._transform.requires_grad_(True)


class MLP(nn.Module):
    """Plain MLP conditioner — follows ``src/nessai/flows/nets.py:12-126``.

    ``Linear(in,H0) → act → [Linear(Hi,Hi+1) → act → dropout] → Linear(H,out)``.
    It has no ``hidden_features`` attribute, so the RQ coupling would not scale
    its logits; the reference only pairs it with RealNVP (``realnvp.py:148-168``).
    """

    def __init__(self, in_shape, out_shape, hidden_sizes, activation=F.relu):
        super().__init__()
        self._in_shape = torch.Size(in_shape)
        self._out_shape = torch.Size(out_shape)
        self._hidden_sizes = hidden_sizes
        self._activation = activation
        if len(hidden_sizes) == 0:
            raise ValueError("List of hidden sizes can't be empty.")
        self._input_layer = nn.Linear(int(np.prod(in_shape)), hidden_sizes[0])
        self._hidden_layers = nn.ModuleList(
            [
                nn.Linear(i, o)
                for i, o in zip(hidden_sizes[:-1], hidden_sizes[1:])
            ]
        )
        self._output_layer = nn.Linear(hidden_sizes[-1], int(np.prod(out_shape)))

    def forward(self, inputs, context=None):
        if context is not None:
            raise ValueError("MLP with conditional inputs is not implemented.")
        if inputs.shape[1:] != self._in_shape:
            raise ValueError(
                "Expected inputs of shape {}, got {}.".format(
                    self._in_shape, inputs.shape[1:]
                )
            )
        outputs = self._input_layer(inputs.reshape(-1, int(np.prod(self._in_shape))))
        outputs = self._activation(outputs)
        for hidden_layer in self._hidden_layers:
            outputs = hidden_layer(outputs)
            outputs = self._activation(outputs)
        outputs = self._output_layer(outputs)
        return outputs.reshape(-1, *self._out_shape)


def create_linear_transform(linear_transform, features):
    """Follows ``src/nessai/flows/utils.py:295-329``."""
    if linear_transform.lower() == "permutation":
        return transforms.RandomPermutation(features=features)
    elif linear_transform.lower() == "lu":
        return transforms.CompositeTransform(
            [
                transforms.RandomPermutation(features=features),
                transforms.LULinear(features, identity_init=True, using_cache=True),
            ]
        )
    elif linear_transform.lower() == "svd":
        return transforms.CompositeTransform(
            [
                transforms.RandomPermutation(features=features),
                transforms.SVDLinear(features, num_householder=10, identity_init=True),
            ]
        )
    raise ValueError(
        f"Unknown linear transform: {linear_transform}. "
        "Choose from: {permutation, lu, svd}."
    )


class RealNVP(NFlow):
    """Follows ``src/nessai/flows/realnvp.py:76-214``."""

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
        activation=F.relu,
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
        if use_volume_preserving:  # realnvp.py:109-112
            coupling_constructor = transforms.AdditiveCouplingTransform
        else:
            coupling_constructor = transforms.AffineCouplingTransform

        if mask is None:  # realnvp.py:114-131
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
            mask_array = torch.empty([num_layers, features])
            for i in range(num_layers):
                mask_array[i] = mask
                mask *= -1
            mask = mask_array

        if net.lower() == "resnet":  # realnvp.py:133-146

            def create_net(in_features, out_features):
                return ResidualNet(
                    in_features,
                    out_features,
                    hidden_features=hidden_features,
                    context_features=context_features,
                    num_blocks=num_blocks_per_layer,
                    activation=activation,
                    dropout_probability=dropout_probability,
                    use_batch_norm=batch_norm_within_layers,
                )

        elif net.lower() == "mlp":  # realnvp.py:148-168
            hidden_sizes = num_blocks_per_layer * [hidden_features]

            def create_net(in_features, out_features):
                return MLP(
                    (in_features,), (out_features,), hidden_sizes, activation=activation
                )

        else:
            raise ValueError(f"Unknown nn type: {net}. Choose from: {{resnet, mlp}}.")

        layers = []
        if pre_transform is not None:  # realnvp.py:177-184
            if pre_transform != "batch_norm":
                raise ValueError(f"Unknown pre-transform: {pre_transform}")
            layers.append(
                transforms.BatchNorm(features=features, **(pre_transform_kwargs or {}))
            )
        for i in range(num_layers):  # realnvp.py:186-206
            if actnorm:
                layers.append(transforms.ActNorm(features=features))
            if linear_transform is not None and linear_transform.lower() != "none":
                layers.append(create_linear_transform(linear_transform, features))
            layers.append(
                coupling_constructor(
                    mask=mask[i], transform_net_create_fn=create_net, **kwargs
                )
            )
            if batch_norm_between_layers:
                layers.append(transforms.BatchNorm(features=features))

        if distribution is None:
            distribution = StandardNormal([features])
        super().__init__(
            transform=transforms.CompositeTransform(layers),
            distribution=distribution,
        )


class NeuralSplineFlow(NFlow):
    """Follows ``src/nessai/flows/nsf.py:60-130``."""

    def __init__(
        self,
        features,
        hidden_features,
        num_layers,
        num_blocks_per_layer,
        num_bins=8,
        context_features=None,
        activation=F.relu,
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

        def create_resnet(in_features, out_features):  # nsf.py:85-95
            return ResidualNet(
                in_features,
                out_features,
                hidden_features=hidden_features,
                context_features=context_features,
                num_blocks=num_blocks_per_layer,
                activation=activation,
                dropout_probability=dropout_probability,
                use_batch_norm=batch_norm_within_layers,
            )

        def spline_constructor(i):  # nsf.py:97-108
            return transforms.PiecewiseRationalQuadraticCouplingTransform(
                mask=create_alternating_binary_mask(
                    features=features, even=(i % 2 == 0)
                ),
                transform_net_create_fn=create_resnet,
                num_bins=num_bins,
                apply_unconditional_transform=apply_unconditional_transform,
                tails=tails,
                tail_bound=tail_bound,
                **kwargs,
            )

        transforms_list = []
        for i in range(num_layers):  # nsf.py:110-122
            if linear_transform is not None and linear_transform.lower() != "none":
                transforms_list.append(
                    create_linear_transform(linear_transform, features)
                )
            transforms_list.append(spline_constructor(i))
            if batch_norm_between_layers:
                transforms_list.append(transforms.BatchNorm(features=features))

        if distribution is None:
            distribution = StandardNormal([features])
        super().__init__(
            transform=transforms.CompositeTransform(transforms_list),
            distribution=distribution,
        )


def silu(x):  # flows/utils.py:24-32
    return torch.mul(x, torch.sigmoid(x))


ACTIVATIONS = {"relu": F.relu, "tanh": torch.tanh, "swish": silu, "silu": silu}
FLOWS = {
    "realnvp": RealNVP,
    "frealnvp": RealNVP,
    "spline": NeuralSplineFlow,
    "nsf": NeuralSplineFlow,
}


def configure_model(config):
    """dict → flow; follows ``src/nessai/flows/utils.py:208-246``."""
    config = copy.deepcopy(config)
    if not isinstance(config["n_inputs"], int):
        raise TypeError("Number of inputs (n_inputs) must be an int")
    if "activation" in config:
        name = config["activation"]
        if name not in ACTIVATIONS:
            raise ValueError(f"Unknown activation: {name}")
        config["activation"] = ACTIVATIONS[name]
    config.pop("distribution_kwargs", None)
    if config.get("distribution", None) is None:
        config.pop("distribution", None)
    FlowClass = config.pop("flow", None)
    ftype = config.pop("ftype", None)
    if FlowClass is None and ftype is None:
        raise RuntimeError("Must specify either 'flow' or 'ftype'.")
    if FlowClass is None:
        if ftype.lower() not in FLOWS:
            raise ValueError(f"Unknown flow: {ftype}")
        FlowClass = FLOWS[ftype.lower()]
    return FlowClass(
        config.pop("n_inputs"),
        config.pop("n_neurons"),
        config.pop("n_blocks"),
        config.pop("n_layers"),
        **config,
    )
