"""Gymnasium-VectorEnv-shaped façades over the fused HIP env step.

They mirror, per lane, the reference's single-env classes (file:line under /root/reference/PyFlyt/):
  QuadXHoverVecEnv        <- gym_envs/quadx_envs/quadx_hover_env.py:10-138 (+ quadx_base_env.py)
  QuadXWaypointsVecEnv    <- gym_envs/quadx_envs/quadx_waypoints_env.py:11-204
  FixedwingWaypointsVecEnv<- gym_envs/fixedwing_envs/fixedwing_waypoints_env.py:11-190
with the same constructor keywords, action/observation layouts, reward, termination and info keys.
The reference has no vector env; the batch dimension and auto-reset follow gymnasium.vector
(num_envs, single_*_space, reset(seed=, options=), step(actions) -> 5-tuple, autoreset_mode).

Tensors are torch.float32 / torch.bool on the ROCm device and are views of buffers the kernels
write in place: copy them if you need them after the next step().

What step() costs the host (round 6): with an action tensor the env has seen before (a policy writing into a fixed buffer, an
action ring) it is one foreign call -- no tensor check, no pointer conversion, no torch kernel, no allocation, no host
synchronisation --, so a closed loop `policy(obs) -> env.step(actions)` can be captured whole in a HIP graph. `infos` is a
LazyInfos: its entries are computed from the flag words of the state when they are READ (like every other value step() returns
they describe the most recent step).
"""
from __future__ import annotations

from typing import Any

import numpy as np
import torch

from .. import _lib as L
from ..engine import BatchEngine
from ..params import build_params
from ..spaces import Box, Dict, batch_box


class LazyInfos(dict):
    """The `infos` of a vector env step as a read-only dict whose values are computed when they are read: each is one or two
    element-wise torch kernels over the lanes' flag words, and a training loop reads them on a small fraction of its steps --
    computed eagerly they were eight kernel launches per step() against the one launch of the step itself. A key maps to a
    zero-argument function; the first read calls it and keeps the result until the env's next step() / reset(), which clears
    the cache (the values describe the MOST RECENT step: read them, or .materialize(), before stepping on if they are to be kept)."""

    __slots__ = ("_make",)

    def __init__(self, makers: dict):
        super().__init__()
        self._make = dict(makers)

    # -- the Mapping protocol over the lazy keys
    def __missing__(self, key):
        fn = self._make.get(key)
        if fn is None:
            raise KeyError(key)
        v = fn()
        dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self._make else default

    def __contains__(self, key):
        return key in self._make

    def __iter__(self):
        return iter(self._make)

    def __len__(self):
        return len(self._make)

    def keys(self):
        return self._make.keys()

    def values(self):
        return [self[k] for k in self._make]

    def items(self):
        return [(k, self[k]) for k in self._make]

    def __repr__(self):
        return "LazyInfos(" + ", ".join(f"{k!r}: " + (repr(dict.__getitem__(self, k)) if dict.__contains__(self, k) else "<lazy>") for k in self._make) + ")"

    def __setitem__(self, key, value):  # (an eager entry: kept across invalidate() only if it is set again)
        self._make[key] = lambda v=value: v
        dict.__setitem__(self, key, value)

    def invalidate(self):
        """Forget the computed values (the env calls this at every step / reset)."""
        dict.clear(self)

    def materialize(self) -> dict:
        """A plain dict of freshly computed tensors that the next step() does not touch."""
        return {k: (v.clone() if torch.is_tensor(v) else (v.materialize() if isinstance(v, LazyInfos) else v)) for k, v in self.items()}


class _VecEnvBase:
    metadata = {"render_modes": [], "autoreset_mode": "next_step"}
    _vehicle = "quadx"
    _task = "hover"

    def __init__(self, num_envs: int, *, device="cuda:0", seed: int = 0, autoreset_mode: str = "next_step",
                 motor_noise: bool = True, lane_offset: int = 0, render_mode=None, **task_kwargs):
        if render_mode is not None:
            raise ValueError("rendering is out of scope for the batched GPU path (SURVEY.md section 2: Camera)")
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self._kwargs = dict(task_kwargs)
        self._autoreset_mode = {"next_step": "next_step", "same_step": "same_step", "disabled": "off", "off": "off"}[str(autoreset_mode).lower()]
        self.metadata = dict(self.metadata, autoreset_mode=autoreset_mode)
        self._noise = "philox" if motor_noise else "off"
        self._lane_offset = int(lane_offset)
        self._seed = int(seed)
        self._build(self._seed)
        P = self.engine.params
        att = (13 if P.angle_repr else 12)
        aux = 4 if P.vehicle == L.QUADX else 6
        self.attitude_dim = att + 4 + aux
        self.single_action_space = Box(low=np.array(list(P.action_low), dtype=np.float32),
                                       high=np.array(list(P.action_high), dtype=np.float32), dtype=np.float32)
        self.action_space = batch_box(self.single_action_space, self.num_envs)
        self._make_obs_space()
        self._needs_reset = True
        self._coerced = None  # (what the last foreign action array was converted into: kept alive until the next one)

    # ------------------------------------------------------------------ construction
    def _build(self, seed):
        P = build_params(self._vehicle, self._task, noise=self._noise, autoreset=self._autoreset_mode, seed=seed, **self._kwargs)
        self.engine = BatchEngine(P, self.num_envs, device=self.device, lane_offset=self._lane_offset)
        # what step() hands back is built ONCE per engine: views of the tensors the kernels write in place + the lazy infos
        self._ret = None
        self._infos_obj = None
        self._final_infos = None

    def _make_obs_space(self):
        self.single_observation_space = Box(low=-np.inf, high=np.inf, shape=(self.attitude_dim,), dtype=np.float32)
        self.observation_space = batch_box(self.single_observation_space, self.num_envs)

    # ------------------------------------------------------------------ gymnasium.vector API
    def reset(self, *, seed: int | None = None, options: dict | None = None):
        """Reset every env (or the subset in options['reset_mask']). A new `seed` re-keys the
        counter-based RNG (motor noise, waypoint sampling); same seed => bit-identical rollouts
        (mirrors tests/test_gym_envs.py:92-112 of the reference)."""
        mask = None if not options else options.get("reset_mask")
        if seed is not None and int(seed) != self._seed:
            if mask is not None:
                raise ValueError("cannot re-seed and partially reset in one call")
            self._seed = int(seed)
            self.engine.close()
            self._build(self._seed)
        elif seed is not None and mask is None:
            # same seed: restart the event counters so that the rollout repeats exactly
            self.engine.state.zero_()
        if mask is not None and not torch.is_tensor(mask):
            mask = torch.as_tensor(np.asarray(mask), dtype=torch.bool, device=self.device)
        self.engine.env_reset(mask=mask)
        self._needs_reset = False
        if self._ret is None:
            self._make_ret()
        self._infos_obj.invalidate()
        return self._obs(self.engine.obs) if self._obs_copies else self._ret[0], self._infos_obj

    def _make_ret(self):
        eng = self.engine
        infos = self._infos()
        if eng.final_obs is not None:
            # SAME_STEP (gymnasium.vector AutoresetMode.SAME_STEP): lanes that finished were re-initialised inside the
            # step, so their terminal observation AND info travel in final_obs / final_info (rows of other lanes are stale)
            fin = self._infos(eng.final_info[:, 0], eng.final_info[:, 1])
            infos._make["final_obs"] = lambda: self._obs(eng.final_obs)
            infos._make["final_info"] = lambda: fin
            infos._make["_final_info"] = lambda: eng.terminated | eng.truncated
            self._final_infos = fin
        else:
            self._final_infos = None
        self._infos_obj = infos
        self._ret = (self._obs(eng.obs), eng.reward, eng.terminated, eng.truncated, infos)

    # (FlattenWaypointEnv with a context longer than the target list pads with zeros: that observation is a fresh tensor per step)
    _obs_copies = False

    def step(self, actions):
        """gymnasium.vector.VectorEnv.step: (obs, reward, terminated, truncated, infos) for every env. The five values are the SAME
        objects every call (views of the buffers the kernel writes, and the lazy infos); nothing here synchronises with the device."""
        if self._needs_reset:
            raise RuntimeError("call reset() before step()")
        if not torch.is_tensor(actions):  # (numpy arrays, lists: one host-to-device copy per step)
            actions = self._coerced = torch.as_tensor(np.asarray(actions), dtype=torch.float32, device=self.device)
        elif actions.dtype != torch.float32 or actions.device != self.device or not actions.is_contiguous():
            actions = self._coerced = actions.to(device=self.device, dtype=torch.float32).contiguous()
        self.engine.env_step(actions)
        self._infos_obj.invalidate()
        if self._final_infos is not None:
            self._final_infos.invalidate()
        if self._obs_copies:
            r = self._ret
            return self._obs(self.engine.obs), r[1], r[2], r[3], r[4]
        return self._ret

    def close(self):
        self.engine.close()

    def sample_actions(self, step_index: int = 0):
        """Device-side uniform sample of the action box (the role of action_space.sample())."""
        out = torch.empty(self.num_envs, 4, dtype=torch.float32, device=self.device)
        return self.engine.sample_actions(out, step_index)

    # ------------------------------------------------------------------ helpers
    def _obs(self, buf):
        return buf

    def _infos(self, flags=None, n_left=None) -> LazyInfos:
        f = self.engine.flags() if flags is None else flags  # (a view of the state's int group: read when an entry is)
        return LazyInfos({
            "out_of_bounds": lambda: (f & L.F_INFO_OOB) != 0,      # quadx_base_env.py:266
            "collision": lambda: (f & L.F_INFO_COLLISION) != 0,    # quadx_base_env.py:260
            "env_complete": lambda: (f & L.F_INFO_COMPLETE) != 0,  # quadx_waypoints_env.py:203
            "nonfinite": lambda: (f & L.F_NONFINITE) != 0,         # NaN/Inf guard (not in the reference, which carries NaNs on silently)
        })

    @property
    def step_count(self):
        return self.engine.ints()[:, 0]


class QuadXHoverVecEnv(_VecEnvBase):
    """PyFlyt/QuadX-Hover-v4, batched. Keywords as quadx_hover_env.py:32-41."""
    _vehicle, _task = "quadx", "hover"

    def __init__(self, num_envs: int, *, sparse_reward: bool = False, flight_mode: int = 0, flight_dome_size: float = 3.0,
                 max_duration_seconds: float = 10.0, angle_representation: str = "quaternion", agent_hz: int = 40, **kw):
        super().__init__(num_envs, sparse_reward=sparse_reward, flight_mode=flight_mode, flight_dome_size=flight_dome_size,
                         max_duration_seconds=max_duration_seconds, angle_representation=angle_representation,
                         agent_hz=agent_hz, **kw)


class _WaypointsMixin:
    def _make_obs_space(self):
        nt = self.engine.params.num_targets
        dome = self.engine.params.dome
        self.num_targets = nt
        tw = self.target_width = 4 if self.engine.params.use_yaw_targets else 3  # quadx_waypoints_env.py:99
        if self._flatten:
            width = self.attitude_dim + tw * self._context_length
            self.single_observation_space = Box(low=-np.inf, high=np.inf, shape=(width,), dtype=np.float32)
            self.observation_space = batch_box(self.single_observation_space, self.num_envs)
        else:
            self.single_observation_space = Dict({
                "attitude": Box(low=-np.inf, high=np.inf, shape=(self.attitude_dim,), dtype=np.float32),
                "target_deltas": Box(low=-2 * dome, high=2 * dome, shape=(nt, tw), dtype=np.float32),
            })
            self.observation_space = Dict({
                "attitude": batch_box(self.single_observation_space["attitude"], self.num_envs),
                "target_deltas": batch_box(self.single_observation_space["target_deltas"], self.num_envs),
            })

    def _obs(self, buf):
        a, tw = self.attitude_dim, self.target_width
        if self._flatten:  # gym_envs/utils/flatten_waypoint_env.py:42-62
            ctx = self._context_length
            have = min(ctx, self.num_targets)
            if have == ctx:
                return buf[:, : a + tw * ctx]
            pad = torch.zeros(buf.shape[0], tw * (ctx - have), dtype=buf.dtype, device=buf.device)
            return torch.cat([buf[:, : a + tw * have], pad], dim=1)
        # the reference's variable-length Sequence becomes a fixed [num_targets, 3] block whose rows
        # past the remaining targets are zero (flatten_waypoint_env.py:49-56 padding convention)
        return {"attitude": buf[:, :a], "target_deltas": buf[:, a:].view(-1, self.num_targets, tw)}

    def _infos(self, flags=None, n_left=None):
        infos = super()._infos(flags)
        if n_left is None:
            n_left = self.engine.ints()[:, 3]
        infos._make["num_targets_reached"] = lambda: self.num_targets - n_left  # quadx_waypoints_env.py:204
        return infos

    @property
    def _obs_copies(self):
        return self._flatten and self._context_length > self.num_targets


class QuadXWaypointsVecEnv(_WaypointsMixin, _VecEnvBase):
    """PyFlyt/QuadX-Waypoints-v4, batched. Keywords as quadx_waypoints_env.py:36-49, incl. `use_yaw_targets` /
    `goal_reach_angle` (target deltas 4 wide: body-frame delta + wrapped yaw error; a waypoint counts only when
    both gates hold). flatten=True returns FlattenWaypointEnv-style rows [attitude, ctx deltas]."""
    _vehicle, _task = "quadx", "waypoints"

    def __init__(self, num_envs: int, *, sparse_reward: bool = False, num_targets: int = 4, use_yaw_targets: bool = False,
                 goal_reach_distance: float = 0.2, goal_reach_angle: float = 0.1, flight_mode: int = 0,
                 flight_dome_size: float = 5.0, max_duration_seconds: float = 10.0, angle_representation: str = "quaternion",
                 agent_hz: int = 30, flatten: bool = False, context_length: int = 2, **kw):
        self._flatten, self._context_length = bool(flatten), int(context_length)
        super().__init__(num_envs, sparse_reward=sparse_reward, num_targets=num_targets, goal_reach_distance=goal_reach_distance,
                         use_yaw_targets=use_yaw_targets, goal_reach_angle=goal_reach_angle, flight_mode=flight_mode, flight_dome_size=flight_dome_size, max_duration_seconds=max_duration_seconds,
                         angle_representation=angle_representation, agent_hz=agent_hz, **kw)


class FixedwingWaypointsVecEnv(_WaypointsMixin, _VecEnvBase):
    """PyFlyt/Fixedwing-Waypoints-v4, batched. Keywords as fixedwing_waypoints_env.py:36-47."""
    _vehicle, _task = "fixedwing", "waypoints"

    def __init__(self, num_envs: int, *, sparse_reward: bool = False, num_targets: int = 4, goal_reach_distance: float = 2.0,
                 flight_mode: int = 0, flight_dome_size: float = 100.0, max_duration_seconds: float = 120.0,
                 angle_representation: str = "quaternion", agent_hz: int = 30, flatten: bool = False,
                 context_length: int = 2, **kw):
        self._flatten, self._context_length = bool(flatten), int(context_length)
        super().__init__(num_envs, sparse_reward=sparse_reward, num_targets=num_targets, goal_reach_distance=goal_reach_distance,
                         flight_mode=flight_mode, flight_dome_size=flight_dome_size, max_duration_seconds=max_duration_seconds,
                         angle_representation=angle_representation, agent_hz=agent_hz, **kw)


_REGISTRY = {
    # ids as registered by the reference, gym_envs/__init__.py:8-43
    "PyFlyt/QuadX-Hover-v4": QuadXHoverVecEnv,
    "PyFlyt/QuadX-Waypoints-v4": QuadXWaypointsVecEnv,
    "PyFlyt/Fixedwing-Waypoints-v4": FixedwingWaypointsVecEnv,
}


def make_vec(env_id: str, num_envs: int, **kwargs):
    """The batched counterpart of `gymnasium.make_vec(id, num_envs=...)` for the supported ids."""
    if env_id not in _REGISTRY:
        raise KeyError(f"{env_id!r} is not on the batched hot path; available: {sorted(_REGISTRY)}")
    return _REGISTRY[env_id](num_envs, **kwargs)


class SingleEnv:
    """One environment with the reference's single-env calling convention -- what
    `gymnasium.make("PyFlyt/QuadX-Hover-v4")` hands back (numpy observations, Python scalars, no
    auto-reset; quadx_base_env.py:128-301) -- on top of a 1-lane batch. For existing scripts and tests;
    throughput comes from `make_vec`."""

    def __init__(self, env_id: str, **kwargs):
        kwargs.pop("autoreset_mode", None)
        self.vec = make_vec(env_id, 1, autoreset_mode="disabled", **kwargs)
        self.observation_space = self.vec.single_observation_space
        self.action_space = self.vec.single_action_space
        self.metadata = dict(self.vec.metadata)
        self.unwrapped = self

    @staticmethod
    def _np(x):
        if isinstance(x, dict):
            return {k: SingleEnv._np(v) for k, v in x.items()}
        return x[0].detach().cpu().numpy()

    def _info(self, infos):
        return {k: (SingleEnv._np(v) if torch.is_tensor(v) or isinstance(v, dict) else v) for k, v in infos.items()}

    def reset(self, *, seed: int | None = None, options: dict | None = None):
        obs, infos = self.vec.reset(seed=seed)
        return self._np(obs), {k: (bool(v) if getattr(v, "dtype", None) == np.bool_ else v) for k, v in self._info(infos).items()}

    def step(self, action):
        a = torch.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, 4), device=self.vec.device)
        obs, rew, term, trunc, infos = self.vec.step(a)
        info = {k: (bool(v) if getattr(v, "dtype", None) == np.bool_ else v) for k, v in self._info(infos).items()}
        return self._np(obs), float(rew[0]), bool(term[0]), bool(trunc[0]), info

    def close(self):
        self.vec.close()


def make(env_id: str, **kwargs) -> SingleEnv:
    """The counterpart of `gymnasium.make(id, **kwargs)` for the supported ids."""
    return SingleEnv(env_id, **kwargs)
