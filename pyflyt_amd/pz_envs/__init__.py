"""Batched counterparts of PyFlyt.pz_envs (pz_envs/__init__.py:3-6), hot-path env only."""
from .ma_quadx_hover import MAQuadXHoverEnv

__all__ = ["MAQuadXHoverEnv"]
