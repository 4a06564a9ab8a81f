"""Pieces of the reference's network/ package that sit on the hot path's doorstep."""
from . import up_pooling
from .up_pooling import MyBlock

__all__ = ["up_pooling", "MyBlock"]
