"""Pieces of the reference's network/ package that sit on the hot path's doorstep."""
from . import conv_tuning, up_pooling
from .conv_tuning import use_tuned_conv_db
from .up_pooling import MyBlock

__all__ = ["conv_tuning", "up_pooling", "MyBlock", "use_tuned_conv_db"]
