"""PettingZoo-ParallelEnv-shaped façade of the team dogfight env
(pz_envs/fixedwing_envs/ma_fixedwing_dogfight_env.py:14-833, ma_fixedwing_base_env.py:17-408).

Same constructor keywords and defaults, agent naming ("uav_i", agents [0, team_size) one team, the rest the other),
dict-in / dict-out `reset()` / `step()`, the flattened observation [attitude 12, surfaces + throttle 6, health, past action 4] +
14 per other active aircraft (its attitude in the own body frame, health, same-team flag) zero padded, the accumulate-then-pop
rewards, culling of finished agents (their aircraft fly on with zero commands), infos (`health`, `received_hits`, `dead`,
`collision`, `out_of_bounds`, `team_win`). All aircraft of an env copy share one world (drone-drone collisions, contact
response on); the whole env step is ONE kernel launch (pyflyt_amd/csrc/dogfight.hpp).

Differences from the reference, by construction of the batched path:
  * `num_envs` independent copies of the whole multi-agent env are stepped at once: dict values are tensors of shape
    [num_envs, ...] (squeezed to the reference's per-agent vectors when num_envs == 1);
  * the spawn circle of every copy is drawn from the counter-based RNG (Philox keyed by seed and the copy's first lane), not
    from `np.random.RandomState(seed)`: the same distribution (:176-213), a different stream;
  * `flatten_observation=False`: the device always writes the flattened, zero-padded vector; the Dict form is a host-side
    view of it -- {"self": [23], "others": [k, 14]} with the k active others exactly as in the reference when num_envs == 1,
    and {"self": [E, 23], "others": [E, A - 1, 14] zero padded, "others_mask": [E, A - 1]} for num_envs > 1 (a Sequence
    space has no batched form);
  * `freeze_wrecks=True` (default False = the reference's behaviour) stops an aircraft where it hits the ground instead of
    letting it tumble to rest: it leaves the others' observations one update later, and the step time of a world with
    wrecks stays that of a world in flight (the contact solve of a tumbling airframe is the most expensive thing here);
  * `assisted_flight=False` does what the reference does with it: the action is six numbers wide, the Aviary stays in mode 0
    (ma_fixedwing_base_env.py:229) and reads the first four, entries 4 and 5 only show up in the observed past action, and the
    thrust remap lands on the unused sixth entry, so the thrust command is action[3] as given (pinned by a reference trajectory).
"""
from __future__ import annotations

from typing import Any

import numpy as np
import torch

from .. import _lib as L
from ..engine import BatchEngine
from ..params import build_params
from ..spaces import Box

_DF_ALIVE, _DF_DEAD, _DF_COLLISION, _DF_OOB, _DF_TEAM_WIN = 1, 16, 32, 64, 128  # dogfight.hpp, state group 6 word 3


class MAFixedwingDogfightEnv:
    metadata = {"render_modes": [], "name": "ma_fixedwing_team_dogfight"}

    def __init__(self, team_size: int = 2, spawn_min_radius: float = 10.0, spawn_max_radius: float = 50.0,
                 spawn_min_height: float = 20.0, spawn_max_height: float = 50.0, damage_per_hit: float = 0.003,
                 lethal_distance: float = 20.0, lethal_angle_radians: float = 0.07, assisted_flight: bool = True,
                 aggressiveness: float = 0.5, cooperativeness: float = 0.5, sparse_reward: bool = False,
                 flatten_observation: bool = True, flight_dome_size: float = 800.0, max_duration_seconds: float = 60.0,
                 agent_hz: int = 30, render_mode=None, num_envs: int = 1, device="cuda:0", seed: int = 0, motor_noise: bool = True,
                 freeze_wrecks: bool = False):
        if render_mode is not None:
            raise ValueError("rendering is out of scope for the batched GPU path")
        self.assisted_flight = bool(assisted_flight)
        self.flatten_observation = bool(flatten_observation)
        if 120 % agent_hz != 0:  # ma_fixedwing_base_env.py:52-57
            lowest, highest = int(120 / (int(120 / agent_hz) + 1)), int(120 / int(120 / agent_hz))
            raise AssertionError(f"`agent_hz` must be round denominator of 120, try {lowest} or {highest}.")
        if not 1 <= team_size <= 4:
            raise ValueError("team_size must be in 1..4 (the aircraft of a world share a wavefront: 2 * team_size <= 8)")
        # (spawn_min_height / spawn_max_height are accepted and unused, as in the reference: :197-201 draws the height
        #  between the RADIUS bounds)
        self.team_size = int(team_size)
        self.num_possible_agents = 2 * self.team_size
        self.possible_agents = ["uav_" + str(r) for r in range(self.num_possible_agents)]
        self.agent_name_mapping = dict(zip(self.possible_agents, range(self.num_possible_agents)))
        self.team_flag = np.concatenate((np.zeros(self.team_size, dtype=bool), np.ones(self.team_size, dtype=bool)))
        self.agents: list[str] = []
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        self._df = dict(team_size=self.team_size, spawn_min_radius=spawn_min_radius, spawn_max_radius=spawn_max_radius,
                        damage_per_hit=damage_per_hit, lethal_distance=lethal_distance, lethal_angle=lethal_angle_radians,
                        aggressiveness=aggressiveness, cooperativeness=cooperativeness, sample_spawn=True, assisted_flight=self.assisted_flight,
                        freeze_wrecks=bool(freeze_wrecks))
        self._kw = dict(flight_dome_size=flight_dome_size, max_duration_seconds=max_duration_seconds, agent_hz=agent_hz,
                        sparse_reward=sparse_reward)
        self._noise = "philox" if motor_noise else "off"
        self._seed = int(seed)
        self._build(self._seed)
        ad = self.engine.action_dim  # 4, or 6 with assisted_flight=False (ma_fixedwing_base_env.py:69)
        self._action_space = Box(low=-np.ones(ad, dtype=np.float32), high=np.ones(ad, dtype=np.float32), dtype=np.float32)
        self._observation_space = Box(low=-np.inf, high=np.inf, shape=(self.engine.obs_dim,), dtype=np.float32)
        self.max_steps = self.engine.params.max_steps
        self.step_count = 0

    def _build(self, seed):
        # Aviary(world_scale=5.0, drone_type="fixedwing", drone_model="acrowing") (ma_fixedwing_base_env.py:193-210)
        P = build_params("fixedwing", "dogfight", noise=self._noise, autoreset="off", seed=seed, angle_representation="euler",
                         vehicle_options=dict(drone_model="acrowing"), world_options=dict(world_scale=5.0), dogfight=self._df, **self._kw)
        self.engine = BatchEngine(P, self.num_possible_agents * self.num_envs, device=self.device)

    def observation_space(self, agent: Any = None):
        return self._observation_space

    def action_space(self, agent: Any = None):
        return self._action_space

    def close(self):
        self.engine.close()

    def _split(self, t):
        """[E*A, ...] -> per-agent views [E, ...] (squeezed when E == 1)."""
        A, E = self.num_possible_agents, self.num_envs
        t = t.view(E, A, *t.shape[1:])
        return [t[:, i].squeeze(0) if E == 1 else t[:, i] for i in range(A)]

    def _obs_out(self, o):
        """The observation of one agent in the form `flatten_observation` asks for (pop_obs_by_id, :724-752)."""
        if self.flatten_observation:
            return o
        A, S = self.num_possible_agents, 19 + self.engine.action_dim  # self block: attitude 12, aux 6, health, past action
        if self.num_envs == 1:
            rows = o[S:].view(A - 1, 14)
            return {"self": o[:S], "others": rows[rows.abs().sum(dim=1) > 0]}  # active others only, in index order (:523-541)
        rows = o[:, S:].view(self.num_envs, A - 1, 14)
        return {"self": o[:, :S], "others": rows, "others_mask": rows.abs().sum(dim=2) > 0}

    @property
    def healths(self):
        """[num_envs, agents] float32 (self.healths of the reference)."""
        return self.engine.state[6, :, 0].view(self.num_envs, self.num_possible_agents)

    @property
    def start_pos(self):
        """[num_envs, agents, 3]: the spawn positions drawn at the last reset."""
        return self.engine.state[13, :, :3].view(self.num_envs, self.num_possible_agents, 3)

    # ------------------------------------------------------------------ ma_fixedwing_dogfight_env.py:215-322
    def reset(self, seed=None, options=None):
        if seed is not None:
            if int(seed) != self._seed:
                self._seed = int(seed)
                # current / past actions are created in __init__ and survive resets: groups 7-8, and 15 (entries 4, 5 of the
                # six-wide action space, assisted_flight=False)
                keep, keep45 = self.engine.state[7:9].clone(), self.engine.state[15].clone()
                self.engine.close()
                self._build(self._seed)
                self.engine.state[7:9] = keep
                self.engine.state[15] = keep45
            # a seeded reset replays the seed's stream from its start (np.random.RandomState(seed) in the reference, :187):
            # the counter RNG's event counters go back to zero; reset(seed=None) continues the stream
            self.engine.state[5, :, 2] = 0.0
        self.step_count = 0
        self.agents = self.possible_agents[:]
        self.engine.env_reset()
        self._done = torch.zeros(self.num_envs, self.num_possible_agents, dtype=torch.bool, device=self.device)
        obs = self._split(self.engine.obs)
        return {ag: self._obs_out(obs[i]) for i, ag in enumerate(self.possible_agents)}, {ag: dict() for ag in self.agents}

    def reset_envs(self, env_mask):
        """num_envs > 1: reset the copies selected by `env_mask` ([num_envs] bool) and leave the others running -- every agent
        of a selected copy starts a new episode (the reference's reset() of that env instance). Returns the observations of
        all copies (fresh ones for the selected copies)."""
        env_mask = torch.as_tensor(env_mask, dtype=torch.bool, device=self.device).view(self.num_envs)
        lane_mask = env_mask.repeat_interleave(self.num_possible_agents)
        self.engine.env_reset(mask=lane_mask)
        self._done[env_mask] = False
        self.agents = self.possible_agents[:]
        obs = self._split(self.engine.obs)
        return {ag: self._obs_out(obs[i]) for i, ag in enumerate(self.possible_agents)}

    @property
    def done(self):
        """[num_envs, agents] bool: latched per copy -- True from the step an agent's episode ended (termination or truncation)
        until that copy is reset. With num_envs > 1 the per-step `terminations` / `truncations` report the ending step only;
        this is what a caller culls by."""
        return self._done

    # ------------------------------------------------------------------ ma_fixedwing_base_env.py:272-334
    def step(self, actions: dict):
        A, E = self.num_possible_agents, self.num_envs
        ad = self.engine.action_dim
        act = torch.zeros(E, A, ad, dtype=torch.float32, device=self.device)
        for k, v in actions.items():
            v = v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v), dtype=torch.float32)
            act[:, self.agent_name_mapping[k]] = v.to(self.device).view(E, ad)
        obs, rew, term, trunc = self.engine.env_step(act.view(E * A, ad))
        side = self.engine.state[6]
        o, r, t, u = self._split(obs), self._split(rew), self._split(term), self._split(trunc)
        health, hits, bits = self._split(side[:, 0]), self._split(side[:, 2].view(torch.int32)), self._split(side[:, 3].view(torch.int32))
        observations, rewards, terminations, truncations, infos = {}, {}, {}, {}, {}
        for ag in self.agents:
            i = self.agent_name_mapping[ag]
            observations[ag], rewards[ag], terminations[ag], truncations[ag] = self._obs_out(o[i]), r[i], t[i], u[i]
            infos[ag] = {"health": health[i], "received_hits": hits[i], "dead": (bits[i] & _DF_DEAD) != 0,
                         "collision": (bits[i] & _DF_COLLISION) != 0, "out_of_bounds": (bits[i] & _DF_OOB) != 0,
                         "team_win": (bits[i] & _DF_TEAM_WIN) != 0}
        self._done |= (term | trunc).view(E, A)
        self.step_count += 1
        # cull finished agents (:326-328); with num_envs > 1 an agent stays listed until it has finished in every copy (its
        # per-copy flags are in terminations / truncations; a copy's finished agent reports reward 0 from then on)
        alive = self._split((side[:, 3].view(torch.int32) & _DF_ALIVE) != 0)
        self.agents = [ag for ag in self.agents if bool(alive[self.agent_name_mapping[ag]].any())]
        return observations, rewards, terminations, truncations, infos
