"""Mirror of the reference package network/libs/base for the hot path: the pixel-adaptive convolution op (pac)."""
from . import pac

__all__ = ["pac"]
