"""Normalising flows whose arithmetic runs in the gfx950 kernels."""

from .base import BaseFlow, HipFlow
from .nsf import NeuralSplineFlow
from .realnvp import RealNVP
from .utils import configure_model, reset_permutations, reset_weights

__all__ = [
    "BaseFlow",
    "HipFlow",
    "NeuralSplineFlow",
    "RealNVP",
    "configure_model",
    "reset_permutations",
    "reset_weights",
]
