"""Batched counterparts of PyFlyt.core (core/__init__.py:3-4)."""
from .aviary import Aviary, AviaryInitException
from .mixed import MixedAviary

__all__ = ["Aviary", "AviaryInitException", "MixedAviary"]
