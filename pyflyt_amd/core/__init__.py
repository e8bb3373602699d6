"""Batched counterparts of PyFlyt.core (core/__init__.py:3-4)."""
from .aviary import Aviary, AviaryInitException

__all__ = ["Aviary", "AviaryInitException"]
