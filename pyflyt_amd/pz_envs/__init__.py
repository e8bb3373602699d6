"""Batched counterparts of PyFlyt.pz_envs (pz_envs/__init__.py:3-6), hot-path envs only."""
from .ma_fixedwing_dogfight import MAFixedwingDogfightEnv
from .ma_quadx_hover import MAQuadXHoverEnv

__all__ = ["MAQuadXHoverEnv", "MAFixedwingDogfightEnv"]
