"""Several drone types in one `Aviary` (core/aviary.py:150-175, tests/test_core.py:228-259): the batched
path keeps one engine per vehicle type (the kernels are specialised per vehicle) and this class presents
them under the reference's single-Aviary surface, indexed by the caller's drone order. Every drone
still lives in its own world (no drone-drone contact)."""
from __future__ import annotations

from typing import Any, Sequence

import numpy as np
import torch


class MixedAviary:
    def __init__(self, start_pos, start_orn, drone_type: Sequence[str], drone_options=None, **kw):
        from .aviary import Aviary, AviaryInitException

        start_pos = np.asarray(start_pos, dtype=np.float64)
        start_orn = np.asarray(start_orn, dtype=np.float64)
        if len(start_pos.shape) != 2 or start_pos.shape[-1] != 3:
            raise AviaryInitException(f"start_pos must be shape (n, 3), currently {start_pos.shape}.")
        if start_orn.shape != start_pos.shape:
            raise AviaryInitException(f"start_orn must be same shape as start_pos, currently {start_orn.shape}.")
        self.num_drones = n = start_pos.shape[0]
        if len(drone_type) != n:  # core/aviary.py:139-143
            raise AviaryInitException(f"If multiple `drone_types` are used, must have same number of `drone_types` ({len(drone_type)}) as number of drones ({n}).")
        if drone_options is None or isinstance(drone_options, dict):
            drone_options = [dict(drone_options or {}) for _ in range(n)]
        if len(drone_options) != n:  # :156-160
            raise AviaryInitException(f"If multiple `drone_options` ({len(drone_options)}) are used, must have same number of `drone_options` as number of drones ({n}).")
        self.drone_type = list(drone_type)
        kinds = list(dict.fromkeys(self.drone_type))  # first-seen order
        self._idx = {k: [i for i, t in enumerate(self.drone_type) if t == k] for k in kinds}
        self._where = {}  # global drone index -> (kind, local index)
        for k, idx in self._idx.items():
            for j, i in enumerate(idx):
                self._where[i] = (k, j)
        base_offset = int(kw.pop("lane_offset", 0))
        self.parts: dict[str, Any] = {}
        done = 0
        for k, idx in self._idx.items():
            opts = [dict(drone_options[i] or {}) for i in idx]
            same = all(o == opts[0] for o in opts)
            # RNG lanes: the counter-based generator is keyed by (seed, lane); every part gets its own DISJOINT lane range
            # [base + drones of the earlier parts, ...) so that no two drones of the fleet -- of this shard or, with
            # lane_offset advanced by num_drones per shard, of another -- ever share a key (interleaved types would
            # otherwise overlap: ['quadx', 'fixedwing', 'quadx'] used lanes {0, 1} and {1})
            self.parts[k] = Aviary(start_pos[idx], start_orn[idx], drone_type=k, drone_options=opts[0] if same else opts,
                                   lane_offset=base_offset + done, **kw)
            done += len(idx)
        any_part = next(iter(self.parts.values()))
        self.physics_hz = any_part.physics_hz
        self.device = any_part.device
        # the world advances by the slowest controller's period (core/aviary.py:288-289)
        self.updates_per_step = max(p.updates_per_step for p in self.parts.values())
        for p in self.parts.values():
            if self.updates_per_step % p.updates_per_step != 0:  # :292-297
                raise AssertionError("Looprates must form common multiples of each other.")
        self.step_period = self.updates_per_step / self.physics_hz
        self.physics_steps = 0
        self.aviary_steps = 0
        self.elapsed_time = 0.0

    # ------------------------------------------------------------------ core/aviary.py surface
    def reset(self) -> None:
        for p in self.parts.values():
            p.reset()
        self.physics_steps = self.aviary_steps = 0
        self.elapsed_time = 0.0

    def set_mode(self, flight_modes) -> None:  # :440-458
        if isinstance(flight_modes, (list, tuple, np.ndarray)):
            assert len(flight_modes) == self.num_drones, f"Expected {self.num_drones} flight_modes, got {len(flight_modes)}."
            for k, idx in self._idx.items():
                self.parts[k].set_mode([int(flight_modes[i]) for i in idx])  # (per-drone modes within a type: QuadX only)
        else:
            for p in self.parts.values():
                p.set_mode(int(flight_modes))

    def set_setpoint(self, index: int, setpoint) -> None:
        k, j = self._where[int(index)]
        self.parts[k].set_setpoint(j, setpoint)

    def set_all_setpoints(self, setpoints) -> None:
        for i in range(self.num_drones):
            self.set_setpoint(i, setpoints[i])

    def set_armed(self, settings) -> None:
        for k, idx in self._idx.items():
            self.parts[k].set_armed([settings[i] for i in idx] if isinstance(settings, (list, tuple, np.ndarray)) else settings)

    def register_wind_field_function(self, wind_field) -> None:
        for p in self.parts.values():
            p.register_wind_field_function(wind_field)

    def step(self) -> None:
        for p in self.parts.values():
            p.step(n_steps=self.updates_per_step // p.updates_per_step)
        self.physics_steps += self.updates_per_step
        self.aviary_steps += 1
        self.elapsed_time = self.physics_steps / self.physics_hz

    def state(self, index: int) -> torch.Tensor:
        k, j = self._where[int(index)]
        return self.parts[k].state(j)

    def aux_state(self, index: int) -> torch.Tensor:
        k, j = self._where[int(index)]
        return self.parts[k].aux_state(j)

    @property
    def all_states(self) -> torch.Tensor:
        """[N, 4, 3] in the caller's drone order."""
        out = torch.empty(self.num_drones, 4, 3, dtype=torch.float32, device=self.device)
        for k, idx in self._idx.items():
            out[torch.as_tensor(idx, device=self.device)] = self.parts[k].all_states
        return out

    @property
    def all_aux_states(self) -> list[torch.Tensor]:
        """One vector per drone (their lengths differ between vehicle types), as the reference's list."""
        return [self.aux_state(i) for i in range(self.num_drones)]

    @property
    def contact_array(self) -> torch.Tensor:
        out = torch.zeros(self.num_drones, dtype=torch.bool, device=self.device)
        for k, idx in self._idx.items():
            out[torch.as_tensor(idx, device=self.device)] = self.parts[k].contact_array
        return out

    def disconnect(self) -> None:
        for p in self.parts.values():
            p.disconnect()

    close = disconnect
