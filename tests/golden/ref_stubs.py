"""ref_stubs.py -- only used by gen_goldens.py, in the build container (where /root/reference exists).

Installs minimal stand-ins for the third-party modules the reference imports at module load
(numba, pybullet, pybullet_data, pybullet_utils, gymnasium, pettingzoo -- all absent here, SURVEY.md
section 8(c)) so that the reference's own Python can be imported and *run* to capture golden
vectors. numba is used by the reference without fastmath (core/utils/compile_helpers.py:13), so
running its functions un-jitted gives the same IEEE results.

`pybullet` is replaced by oracle/fake_bullet.py (our own restatement, not real Bullet).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REFERENCE = "/root/reference"


class RecordingRNG:
    """Wraps a numpy Generator and logs every normal()/uniform() draw in call order.
    F32 (class switch, set by the generators of the cascaded-flight-mode fixtures): every draw is handed to the reference ROUNDED to the
    nearest float32 value -- still a float64 number, still the reference's arithmetic on it. The device's interface takes float32
    actions and float32 injected draws; a cascaded controller turns the 6e-8 by which a float64 draw differs from its float32 rounding
    into 1e-3 within an episode, which says nothing about the kernel. With float32-exact inputs both sides
    start from identical numbers."""

    F32 = False

    def __init__(self, gen):
        self._gen = gen
        self.log = []  # (kind, ndarray)

    def _q(self, out):
        if not RecordingRNG.F32:
            return out
        r = np.asarray(out, dtype=np.float64).astype(np.float32).astype(np.float64)
        return float(r) if np.ndim(out) == 0 else r

    def normal(self, *args, **kwargs):
        out = self._q(self._gen.normal(*args, **kwargs))
        self.log.append(("normal", np.atleast_1d(np.asarray(out, dtype=np.float64)).copy()))
        return out

    def uniform(self, *args, **kwargs):
        out = self._q(self._gen.uniform(*args, **kwargs))
        self.log.append(("uniform", np.atleast_1d(np.asarray(out, dtype=np.float64)).copy()))
        return out

    def __getattr__(self, name):
        return getattr(self._gen, name)

    def drain(self, kind):
        vals = [v for k, v in self.log if k == kind]
        self.log = [(k, v) for k, v in self.log if k != kind]
        return np.concatenate(vals) if vals else np.zeros(0)


def install():
    sys.path.insert(0, REPO)
    from oracle import fake_bullet

    # ---------------- numba
    nb = types.ModuleType("numba")
    nb.njit = lambda f=None, **kw: f if f is not None else (lambda g: g)

    class _T:
        def __getitem__(self, k):
            return self

    nb.float64 = _T()
    exp = types.ModuleType("numba.experimental")
    exp.jitclass = lambda spec=None: (lambda cls: cls)
    nb.experimental = exp
    sys.modules["numba"] = nb
    sys.modules["numba.experimental"] = exp

    # ---------------- pybullet family
    pb = types.ModuleType("pybullet")
    for name in ("DIRECT", "GUI", "LINK_FRAME", "WORLD_FRAME", "URDF_USE_INERTIA_FROM_FILE"):
        setattr(pb, name, getattr(fake_bullet.BulletClient, name))
    pb.getQuaternionFromEuler = fake_bullet.BulletClient.getQuaternionFromEuler
    pb.getEulerFromQuaternion = fake_bullet.BulletClient.getEulerFromQuaternion
    pb.getMatrixFromQuaternion = fake_bullet.BulletClient.getMatrixFromQuaternion
    pb.isNumpyEnabled = lambda: True
    sys.modules["pybullet"] = pb
    pbd = types.ModuleType("pybullet_data")
    pbd.getDataPath = lambda: ""
    sys.modules["pybullet_data"] = pbd
    pbu = types.ModuleType("pybullet_utils")
    bc = types.ModuleType("pybullet_utils.bullet_client")
    bc.BulletClient = fake_bullet.BulletClient
    pbu.bullet_client = bc
    sys.modules["pybullet_utils"] = pbu
    sys.modules["pybullet_utils.bullet_client"] = bc

    # ---------------- gymnasium
    gym = types.ModuleType("gymnasium")

    class Space:
        pass

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float64):
            if shape is None:
                shape = np.shape(low)
            self.low = np.broadcast_to(np.asarray(low, dtype=np.float64), shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=np.float64), shape).copy()
            self.shape = tuple(shape)
            self.dtype = dtype

    class Dict(Space, dict):
        def __init__(self, spaces):
            dict.__init__(self, spaces)

    class Sequence(Space):
        def __init__(self, space, stack=False):
            self.feature_space = space
            self.stack = stack

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box, spaces.Dict, spaces.Sequence, spaces.Space = Box, Dict, Sequence, Space

    class Env:
        metadata = {"render_modes": []}
        rng_factory = staticmethod(lambda seed: RecordingRNG(np.random.default_rng(seed)))

        @property
        def np_random(self):
            if getattr(self, "_np_random", None) is None:
                self._np_random = Env.rng_factory(None)
            return self._np_random

        @np_random.setter
        def np_random(self, v):
            self._np_random = v

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random = Env.rng_factory(seed)

        @property
        def unwrapped(self):
            return self

    class ObservationWrapper(Env):
        def __init__(self, env):
            self.env = env

    core = types.ModuleType("gymnasium.core")
    core.Env, core.ObservationWrapper = Env, ObservationWrapper
    utils = types.ModuleType("gymnasium.utils")
    utils.colorize = lambda s, **kw: s
    envs = types.ModuleType("gymnasium.envs")
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = lambda **kw: None
    envs.registration = reg
    gym.Env, gym.Space, gym.spaces, gym.core, gym.utils, gym.envs = Env, Space, spaces, core, utils, envs
    gym.ObservationWrapper = ObservationWrapper
    for name, mod in (
        ("gymnasium", gym), ("gymnasium.spaces", spaces), ("gymnasium.core", core),
        ("gymnasium.utils", utils), ("gymnasium.envs", envs), ("gymnasium.envs.registration", reg),
    ):
        sys.modules[name] = mod

    # ---------------- pettingzoo
    pz = types.ModuleType("pettingzoo")

    class ParallelEnv:
        pass

    pz.ParallelEnv = ParallelEnv
    sys.modules["pettingzoo"] = pz

    sys.path.insert(0, REFERENCE)
    return gym
