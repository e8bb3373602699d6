"""Batched counterparts of PyFlyt.gym_envs (gym_envs/__init__.py:8-43), hot-path ids only."""
from .vector_envs import FixedwingWaypointsVecEnv, QuadXHoverVecEnv, QuadXWaypointsVecEnv, SingleEnv, make, make_vec

__all__ = ["QuadXHoverVecEnv", "QuadXWaypointsVecEnv", "FixedwingWaypointsVecEnv", "SingleEnv", "make", "make_vec"]
