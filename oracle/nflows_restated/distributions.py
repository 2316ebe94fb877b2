"""Base distributions, CPU restatement of the published nflows classes.

Test infrastructure only (see ``oracle/__init__.py``).  Reference call sites:
``src/nessai/flows/realnvp.py:208-209`` / ``nsf.py:124-125``
(``StandardNormal([features])``) and ``src/nessai/flows/base.py:236-287``
(``log_prob``, ``sample``, ``sample_and_log_prob``).

(Algorithm and structure: nflows, MIT License, (c) 2020 Durkan, Bekasov, Murray, Papamakarios - see
``oracle/nflows_restated/__init__.py`` and ``LICENSE.nflows``.)
"""

import numpy as np
import torch
from torch import nn

from .utils import sum_except_batch


class Distribution(nn.Module):
    """Base class: subclasses implement ``_log_prob`` and ``_sample``."""

    def forward(self, *args):
        raise RuntimeError("Forward method cannot be called for a Distribution object.")

    def log_prob(self, inputs, context=None):
        inputs = torch.as_tensor(inputs)
        if context is not None:
            context = torch.as_tensor(context)
            if inputs.shape[0] != context.shape[0]:
                raise ValueError(
                    "Number of input items must be equal to number of context items."
                )
        return self._log_prob(inputs, context)

    def _log_prob(self, inputs, context):
        raise NotImplementedError()

    def sample(self, num_samples, context=None, batch_size=None):
        if not (isinstance(num_samples, int) and num_samples > 0):
            raise TypeError("Number of samples must be a positive integer.")
        if context is not None:
            context = torch.as_tensor(context)
        if batch_size is None:
            return self._sample(num_samples, context)
        num_batches = num_samples // batch_size
        num_leftover = num_samples % batch_size
        samples = [self._sample(batch_size, context) for _ in range(num_batches)]
        if num_leftover > 0:
            samples.append(self._sample(num_leftover, context))
        return torch.cat(samples, dim=0)

    def _sample(self, num_samples, context):
        raise NotImplementedError()

    def sample_and_log_prob(self, num_samples, context=None):
        samples = self.sample(num_samples, context=context)
        log_prob = self.log_prob(samples, context=context)
        return samples, log_prob

    def mean(self, context=None):
        if context is not None:
            context = torch.as_tensor(context)
        return self._mean(context)

    def _mean(self, context):
        raise NotImplementedError()


class StandardNormal(Distribution):
    """N(0, I): ``log p = -0.5 * sum(z^2) - 0.5 * D * log(2 pi)``.

    The normaliser is a float64 buffer (``_log_z``), as in
    ``src/nessai/flows/distributions.py:36-43``; subtracting the 0-dim float64
    buffer from a float32 tensor keeps float32.
    """

    def __init__(self, shape):
        super().__init__()
        self._shape = torch.Size(shape)
        self.register_buffer(
            "_log_z",
            torch.tensor(0.5 * np.prod(shape) * np.log(2 * np.pi), dtype=torch.float64),
            persistent=False,
        )

    def _log_prob(self, inputs, context):
        if inputs.shape[1:] != self._shape:
            raise ValueError(
                "Expected input of shape {}, got {}".format(
                    self._shape, inputs.shape[1:]
                )
            )
        neg_energy = -0.5 * sum_except_batch(inputs**2, num_batch_dims=1)
        return neg_energy - self._log_z

    def _sample(self, num_samples, context):
        if context is None:
            return torch.randn(num_samples, *self._shape, device=self._log_z.device)
        raise NotImplementedError("conditional sampling is outside the restated scope")

    def _mean(self, context):
        if context is None:
            return self._log_z.new_zeros(self._shape)
        raise NotImplementedError()
