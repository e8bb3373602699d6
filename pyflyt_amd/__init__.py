"""pyflyt_amd -- MI355X-native batched UAV-physics step behind PyFlyt-shaped APIs.

Only the hot path of jjshoots/PyFlyt is implemented: Aviary.step() + the
per-drone control/physics/state loop + the 6-DoF integrator, as hand-written HIP kernels for
gfx950 called through a C ABI (include/pyflyt_amd.h). There is no CPU fallback.
"""
from . import _lib
from ._lib import PyFlytAmdError
from .params import build_params

__all__ = ["_lib", "PyFlytAmdError", "build_params"]
__version__ = "0.1.0"
