"""Normalising flows whose arithmetic runs in the gfx950 kernels.

Public names mirror the reference's ``nessai.flows`` for the two families of the hot path
(``RealNVP``, ``NeuralSplineFlow``) and its factory / reset helpers; ``HipFlow`` is the common
base that owns the packed device image."""

from . import base as _base
from . import nsf as _nsf
from . import realnvp as _realnvp
from . import utils as _utils

BaseFlow, HipFlow = _base.BaseFlow, _base.HipFlow
RealNVP = _realnvp.RealNVP
NeuralSplineFlow = _nsf.NeuralSplineFlow
configure_model = _utils.configure_model
reset_weights, reset_permutations = _utils.reset_weights, _utils.reset_permutations

__all__ = ["BaseFlow", "HipFlow", "RealNVP", "NeuralSplineFlow", "configure_model", "reset_weights", "reset_permutations"]
