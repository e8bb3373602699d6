"""Observation/action space descriptions. gymnasium's own `spaces` are used when the package is
importable; otherwise these minimal duck types provide the attributes RL code reads
(`shape`, `low`, `high`, `dtype`, `sample()`, `contains()`), so that pyflyt_amd has no hard
dependency on gymnasium (it is absent on the GPU box)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium import spaces as _gs

    Box, Dict = _gs.Box, _gs.Dict
    HAVE_GYMNASIUM = True
except Exception:  # gymnasium not installed
    HAVE_GYMNASIUM = False

    class Box:  # type: ignore[no-redef]
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            if shape is None:
                shape = np.shape(low)
            self.shape = tuple(shape)
            self.dtype = np.dtype(dtype)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Dict(dict):  # type: ignore[no-redef]
        def __init__(self, spaces):
            super().__init__(spaces)

        def sample(self):
            return {k: v.sample() for k, v in self.items()}

        def contains(self, x):
            return all(k in x and v.contains(x[k]) for k, v in self.items())


def batch_box(space: "Box", n: int) -> "Box":
    """The batched counterpart of a single-env Box (gymnasium.vector.utils.batch_space)."""
    low = np.broadcast_to(space.low, (n,) + tuple(space.shape)).copy()
    high = np.broadcast_to(space.high, (n,) + tuple(space.shape)).copy()
    return Box(low=low, high=high, dtype=space.dtype)
