"""PettingZoo-ParallelEnv-shaped façade of the multi-agent QuadX hover env
(pz_envs/quadx_envs/ma_quadx_base_env.py:17-371, ma_quadx_hover_env.py:14-205).

Same constructor keywords, agent naming ("uav_i"), dict-in/dict-out `reset()` / `step()`, observation
layout [ang_vel, quat|rpy, lin_vel, lin_pos, throttle(4), past action(4), start_pos(3)], additive
-100 penalties, per-call term/trunc flags, culling of finished agents.

`shared_world`: the reference puts all agents of an env in ONE Bullet world (ma_quadx_base_env.py:206-241). With
shared_world=True so does this env: the agents of a copy are adjacent lanes of one wavefront, a hit between two drones
enters both contact arrays and ends both episodes (ma_quadx_hover_env.py:181), and a contact point anywhere in the world --
e.g. a finished drone lying on the floor -- switches off the rotational drag of every drone (quadx.py:509). The contact
response is then on (a finished drone must come to rest on the floor, not fall through it); between drones there is detection
only (no impulses). The number of agents must divide 64. Default False: every agent an independent lane (round 1 semantics,
and the specialised kernel).

`num_envs` independent copies of the whole multi-agent env are stepped at once: dict values are
tensors of shape [num_envs, ...] (squeezed to the reference's per-agent vectors when num_envs == 1).

What step() costs (round 6). The per-agent values it returns are views, built once, of the tensors the kernel writes; the
per-agent infos are lazy (gym_envs.vector_envs.LazyInfos). Actions: `env.action_buffers()` hands out, per agent, a view of
the env's own [num_envs, num_agents, 4] action tensor -- a policy that writes its actions THERE and passes those views back
costs step() no copy; any other tensor / array is copied into place (one small kernel per agent). Culling (`cull_agents`,
default True: ma_quadx_base_env.py:365-369) needs the flags on the host: ONE synchronisation per step. With
`cull_agents=False` the agent list stays whole (the per-copy flags are in terminations / truncations anyway), step() makes
one foreign call and nothing else, and a whole policy -> step loop can be captured in a HIP graph.
"""
from __future__ import annotations

from typing import Any

import numpy as np
import torch

from .. import _lib as L
from ..engine import BatchEngine
from ..gym_envs.vector_envs import LazyInfos
from ..params import build_params, quat_from_euler
from ..spaces import Box


class MAQuadXHoverEnv:
    metadata = {"render_modes": [], "name": "ma_quadx_hover"}

    def __init__(self, start_pos=np.array([[-1.0, -1.0, 1.0], [1.0, -1.0, 1.0], [-1.0, 1.0, 1.0], [1.0, 1.0, 1.0]]),
                 start_orn=np.zeros((4, 3)), sparse_reward: bool = False, flight_mode: int = 0, flight_dome_size: float = 10.0,
                 max_duration_seconds: float = 30.0, angle_representation: str = "quaternion", agent_hz: int = 40,
                 render_mode=None, num_envs: int = 1, device="cuda:0", seed: int = 0, motor_noise: bool = True,
                 shared_world: bool = False, cull_agents: bool = True):
        if render_mode is not None:
            raise ValueError("rendering is out of scope for the batched GPU path")
        start_pos, start_orn = np.asarray(start_pos, dtype=np.float64), np.asarray(start_orn, dtype=np.float64)
        assert len(start_pos.shape) == 2 and start_pos.shape[-1] == 3, f"Expected `start_pos` to be of shape [num_agents, 3], got {start_pos.shape}."
        assert start_pos.shape == start_orn.shape, f"Expected `start_pos` to be of shape [num_agents, 3], got {start_pos.shape}."
        if 120 % agent_hz != 0:  # ma_quadx_base_env.py:60-65
            lowest, highest = int(120 / (int(120 / agent_hz) + 1)), int(120 / int(120 / agent_hz))
            raise AssertionError(f"`agent_hz` must be round denominator of 120, try {lowest} or {highest}.")
        self.start_pos, self.start_orn = start_pos, start_orn
        self.num_possible_agents = len(start_pos)
        self.possible_agents = ["uav_" + str(r) for r in range(self.num_possible_agents)]
        self.agent_name_mapping = dict(zip(self.possible_agents, range(self.num_possible_agents)))
        self.agents: list[str] = []
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self._kw = dict(flight_mode=flight_mode, flight_dome_size=flight_dome_size, max_duration_seconds=max_duration_seconds,
                        angle_representation=angle_representation, agent_hz=agent_hz, sparse_reward=sparse_reward)
        self._noise = "philox" if motor_noise else "off"
        self._seed = int(seed)
        self.shared_world = bool(shared_world)
        self.cull_agents = bool(cull_agents)
        if self.shared_world:
            if 64 % self.num_possible_agents != 0:
                raise ValueError("shared_world=True needs a number of agents that divides 64 (the agents of a world share a wavefront)")
            self._kw.update(agents_per_world=self.num_possible_agents, world_options=dict(contact_response=True))
        self._build(self._seed)
        att = 13 if angle_representation == "quaternion" else 12
        xyz, thr = np.pi, 0.8
        self._action_space = Box(low=np.array([-xyz, -xyz, -xyz, 0.0], dtype=np.float32),
                                 high=np.array([xyz, xyz, xyz, thr], dtype=np.float32), dtype=np.float32)
        self._observation_space = Box(low=-np.inf, high=np.inf, shape=(att + 4 + 4 + 3,), dtype=np.float32)
        self.max_steps = self.engine.params.max_steps
        self.step_count = 0

    def _build(self, seed):
        A, E = self.num_possible_agents, self.num_envs
        # The kernels take every agent's spawn from the state's side block; the parameter block's start
        # pose only tells pf_ctx_create whether the specialised kernel's level-spawn settle applies
        # (quadx_fast.hpp), so it carries the lowest position and the least level orientation of the team.
        worst_orn = self.start_orn[np.argmax(np.abs(self.start_orn[:, :2]).sum(axis=1))]
        P = build_params("quadx", "ma_hover", noise=self._noise, autoreset="off", seed=seed,
                         start_pos=self.start_pos[np.argmin(self.start_pos[:, 2])], start_orn=worst_orn, **self._kw)
        self.engine = BatchEngine(P, A * E, device=self.device)
        # lane = env * A + agent; the per-lane spawn lives in the state's side block
        pose = np.concatenate([self.start_pos, np.stack([quat_from_euler(o) for o in self.start_orn])], axis=1)  # [A,7]
        side = np.zeros((A * E, 12), dtype=np.float32)
        side[:, :7] = np.tile(pose, (E, 1))
        st = self.engine.state
        st[12:15] = torch.tensor(side, device=self.device).view(A * E, 3, 4).permute(1, 0, 2)
        # what step() hands back, built once per engine: per-agent views of the kernel's output tensors, lazy per-agent infos
        self._act = torch.zeros(E, A, 4, dtype=torch.float32, device=self.device)
        self._act_flat = self._act.view(E * A, 4)
        self._act_views = self._split(self._act_flat)
        o, r = self._split(self.engine.obs), self._split(self.engine.reward)
        t, u = self._split(self.engine.terminated), self._split(self.engine.truncated)
        f = self._split(self.engine.flags())
        self._out = []
        for i in range(A):
            infos = LazyInfos({"collision": (lambda fi=f[i]: (fi & L.F_INFO_COLLISION) != 0),
                               "out_of_bounds": (lambda fi=f[i]: (fi & L.F_INFO_OOB) != 0)})
            self._out.append((o[i], r[i], t[i], u[i], infos))
        self._done_all = self.engine.terminated.view(E, A), self.engine.truncated.view(E, A)

    def observation_space(self, agent: Any = None):
        return self._observation_space

    def action_space(self, agent: Any = None):
        return self._action_space

    def close(self):
        self.engine.close()

    def action_buffers(self) -> dict:
        """{agent: view [num_envs, 4] (or [4]) of the env's own action tensor}: write the actions here and pass the views to step()
        -- no copy on the way in."""
        return {ag: self._act_views[i] for i, ag in enumerate(self.possible_agents)}

    def _split(self, t):
        """[E*A, ...] -> per-agent views [E, ...] (squeezed when E == 1)."""
        A, E = self.num_possible_agents, self.num_envs
        t = t.view(E, A, *t.shape[1:])
        return [t[:, i].squeeze(0) if E == 1 else t[:, i] for i in range(A)]

    # ------------------------------------------------------------------ ma_quadx_hover_env.py:99-121
    def reset(self, seed=None, options=None):
        if seed is not None and int(seed) != self._seed:
            self._seed = int(seed)
            keep = self.engine.state[12:16].clone()  # spawn poses and the action memories survive
            self.engine.close()
            self._build(self._seed)
            self.engine.state[12:16] = keep
        self.step_count = 0
        self.agents = self.possible_agents[:]
        self.engine.env_reset()
        obs = self._split(self.engine.obs)
        return {ag: obs[i] for i, ag in enumerate(self.possible_agents)}, {ag: dict() for ag in self.agents}

    # ------------------------------------------------------------------ ma_quadx_base_env.py:309-371
    def step(self, actions: dict):
        A, E = self.num_possible_agents, self.num_envs
        act, views = self._act, self._act_views
        # current_actions *= 0, then the live agents' actions (:329-332): an agent that is not in `actions` flies on zeros
        missing = A - len(actions)
        for k, v in actions.items():
            i = self.agent_name_mapping[k]
            if v is views[i]:
                continue  # (written in place by the caller: action_buffers())
            if not torch.is_tensor(v):
                v = torch.as_tensor(np.asarray(v), dtype=torch.float32)
            act[:, i] = v.to(self.device).view(E, 4)
        if missing:
            for ag, i in self.agent_name_mapping.items():
                if ag not in actions:
                    act[:, i] = 0.0
        self.engine.env_step(self._act_flat)
        out = self._out
        observations, rewards, terminations, truncations, infos = {}, {}, {}, {}, {}
        for ag in self.agents:
            o = out[self.agent_name_mapping[ag]]
            observations[ag], rewards[ag], terminations[ag], truncations[ag], infos[ag] = o
            o[4].invalidate()
        self.step_count += 1
        if self.cull_agents:
            # cull finished agents (:365-369); with num_envs > 1 an agent stays listed until it has finished in every copy (its
            # per-copy flags are in terminations / truncations). One reduction and one host synchronisation per step for all agents.
            t, u = self._done_all
            gone = (t | u).all(dim=0).tolist()
            self.agents = [ag for ag in self.agents if not gone[self.agent_name_mapping[ag]]]
        return observations, rewards, terminations, truncations, infos
