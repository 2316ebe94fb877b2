"""Small tensor helpers restated from the published nflows API.

Test infrastructure only (see ``oracle/__init__.py``).

(Algorithm and structure: nflows, MIT License, (c) 2020 Durkan, Bekasov, Murray, Papamakarios - see
``oracle/nflows_restated/__init__.py`` and ``LICENSE.nflows``.)

Reference call sites that fix the semantics:
* ``create_alternating_binary_mask`` — ``src/nessai/flows/nsf.py:99-101``
  (``even=(i % 2 == 0)``; mask value 1 marks a *transformed* feature).
* ``torchutils.sum_except_batch`` — ``src/nessai/flows/distributions.py:53-55``.
"""

import torch


def create_alternating_binary_mask(features, even=True):
    """0/1 mask with ones at even (``even=True``) or odd positions."""
    mask = torch.zeros(features).byte()
    start = 0 if even else 1
    mask[start::2] += 1
    return mask


def sum_except_batch(x, num_batch_dims=1):
    """Sum over every dimension after the first ``num_batch_dims``."""
    if num_batch_dims < 0:
        raise TypeError("Number of batch dimensions must be non-negative.")
    reduce_dims = list(range(num_batch_dims, x.ndimension()))
    if not reduce_dims:
        return x
    return torch.sum(x, dim=reduce_dims)


def searchsorted(bin_locations, inputs, eps=1e-6):
    """Index of the bin holding each input.

    The last knot is nudged by ``eps`` so an input sitting exactly on the
    right edge lands in the last bin.  Mutates
    ``bin_locations`` in place, as the published implementation does.
    """
    bin_locations[..., -1] += eps
    return torch.sum(inputs[..., None] >= bin_locations, dim=-1) - 1


class _TorchUtils:
    """Namespace standing in for the ``nflows.utils.torchutils`` module."""

    sum_except_batch = staticmethod(sum_except_batch)
    searchsorted = staticmethod(searchsorted)


torchutils = _TorchUtils()
