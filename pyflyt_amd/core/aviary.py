"""Batched `Aviary`: the reference's simulation orchestrator (core/aviary.py:47-531) for N drones
that each live in their OWN world (independent lanes), stepped by the HIP kernels.

Same surface as the reference where it is on the hot path:
  Aviary(start_pos[N,3], start_orn[N,3], drone_type, drone_options=..., physics_hz=240,
         world_scale=1.0, seed=...)                           core/aviary.py:69-216
         drone_type: "quadx" | "fixedwing" | "rocket" (one type per Aviary)
         drone_options: control_hz, starting_velocity, drone_model ("cf2x" | "primitive_drone",
         quadx.py:29), starting_fuel_ratio (rocket.py:47), or any entry of the parameter tables in
         pyflyt_amd/params.py
  reset()                                                     :218-312
  set_mode(int) / set_setpoint(i, sp) / set_all_setpoints(sp) :440-478
  step()                                                      :480-531
  state(i) (4,3) / aux_state(i) / all_states / all_aux_states :335-421
  contact_array  -> per-drone bool "touches the floor" (the reference's body-pair matrix collapses
                    to this because every drone is alone in its world)
  wind_type / wind_options / register_wind_field_function()   :266-285,324-333 -- the field is a
                    function of (time, positions[M,3] device tensor) -> wind[M,3] device tensor,
                    sampled after every physics tick exactly where and when the reference samples
                    it (boring_bodies.py:93-96, lifting_surfaces.py:88-93)
Several drone types in one Aviary are composed from one engine per type (core/mixed.py).
Custom controllers (register_controller) run batch-wide on device tensors.
Not carried over (out of scope): rendering/cameras, drone-drone contact.
"""
from __future__ import annotations

from typing import Any, Sequence

import numpy as np
import torch

from ..engine import BatchEngine
from ..params import build_params, quat_from_euler


class AviaryInitException(Exception):
    """Mirrors core/aviary.py:21-44."""


class _DroneProxy:
    def __init__(self, aviary, index):
        self._aviary, self.index = aviary, index

    def register_controller(self, controller_id: int, controller_constructor, base_mode: int) -> None:
        self._aviary.register_controller(controller_id=controller_id, controller_constructor=controller_constructor, base_mode=base_mode)

    @property
    def state(self):
        return self._aviary.state(self.index)

    @property
    def aux_state(self):
        return self._aviary.aux_state(self.index)

    @property
    def setpoint(self):
        return self._aviary.setpoints[self.index]


class Aviary:
    def __new__(cls, start_pos=None, start_orn=None, drone_type="quadx", *args, **kwargs):
        # several drone types in one Aviary (core/aviary.py:139-175): one engine per type behind one surface
        if cls is Aviary and not isinstance(drone_type, str) and len(set(drone_type)) > 1:
            from .mixed import MixedAviary

            return MixedAviary(start_pos, start_orn, drone_type, *args, **kwargs)
        return super().__new__(cls)

    def __init__(self, start_pos, start_orn, drone_type: str | Sequence[str] = "quadx", drone_options: dict | None = None,
                 wind_type=None, wind_options=None, render: bool = False, physics_hz: int = 240, world_scale: float = 1.0,
                 seed: None | int = None, device="cuda:0", motor_noise: bool = True, lane_offset: int = 0):
        start_pos = np.asarray(start_pos, dtype=np.float64)
        start_orn = np.asarray(start_orn, dtype=np.float64)
        if len(start_pos.shape) != 2 or start_pos.shape[-1] != 3:  # core/aviary.py:125-128
            raise AviaryInitException(f"start_pos must be shape (n, 3), currently {start_pos.shape}.")
        if start_orn.shape != start_pos.shape:  # :129-136
            raise AviaryInitException(f"start_orn must be same shape as start_pos, currently {start_orn.shape}.")
        if not isinstance(drone_type, str):
            if len(drone_type) != start_pos.shape[0]:  # core/aviary.py:139-143
                raise AviaryInitException(f"If multiple `drone_types` are used, must have same number of `drone_types` ({len(drone_type)}) as number of drones ({start_pos.shape[0]}).")
            drone_type = drone_type[0]  # (several distinct types are handled by MixedAviary, see __new__)
        if drone_type not in ("quadx", "fixedwing", "rocket"):
            raise AviaryInitException(f"drone_type {drone_type!r} is not on the batched path (quadx, fixedwing, rocket)")
        if render:
            raise AviaryInitException("rendering is out of scope for the batched GPU path")
        if wind_type is not None and not callable(wind_type):  # core/aviary.py:270-285
            if isinstance(wind_type, str):
                raise AssertionError(f"Unknown wind field model {wind_type}.")
            raise LookupError("Invalid setting for wind field.")
        self.wind_type, self.wind_options = wind_type, dict(wind_options or {})
        self._registered_controllers: dict[int, Any] = {}
        self._registered_base_modes: dict[int, int] = {}
        self.wind_field = None
        self.num_drones = start_pos.shape[0]
        self.drone_type = drone_type
        self.device = torch.device(device)
        self.physics_hz = int(physics_hz)
        # drone_options: one dict for the whole batch, or one per drone (core/aviary.py:150-163). Per-drone
        # dicts may differ in `control_hz` only (tests/test_core.py:34-62): the Aviary then steps at the
        # slowest controller's rate (aviary.py:288-289) and every drone's controller fires at its own.
        per_drone_hz = None
        per_drone_vel = None
        if isinstance(drone_options, (list, tuple)):
            if len(drone_options) != self.num_drones:
                raise AviaryInitException(f"drone_options must be a dict or a sequence of {self.num_drones} dicts")
            dicts = [dict(d or {}) for d in drone_options]
            if any("control_hz" in d for d in dicts):
                per_drone_hz = [int(d.pop("control_hz", 120)) for d in dicts]
            if any("starting_velocity" in d for d in dicts):  # fixedwing.py:35, ma_fixedwing_dogfight_env.py:218-222
                default_v = (20.0, 0.0, 0.0) if drone_type == "fixedwing" else (0.0, 0.0, 0.0)
                per_drone_vel = np.array([np.asarray(d.pop("starting_velocity", default_v), dtype=np.float64) for d in dicts])
            if any(d != dicts[0] for d in dicts):
                raise AviaryInitException("per-drone drone_options may differ in `control_hz` and `starting_velocity` only")
            opts = dicts[0]
            if per_drone_hz is not None and len(set(per_drone_hz)) > 1:
                for hz in per_drone_hz:
                    if self.physics_hz % hz != 0:  # base_drone.py:95-98
                        raise ValueError(f"`physics_hz` ({self.physics_hz}) must be multiple of `control_hz` ({hz}).")
                rates = sorted(set(per_drone_hz))
                if any(b % a != 0 for a, b in zip(rates[:-1], rates[1:])):  # aviary.py:292-297
                    raise AssertionError("Looprates must form common multiples of each other.")
                opts["control_hz"] = min(per_drone_hz)
            elif per_drone_hz is not None:
                opts["control_hz"], per_drone_hz = per_drone_hz[0], None
        else:
            opts = dict(drone_options or {})
        vopts: dict[str, Any] = {}
        if "control_hz" in opts:
            vopts["control_hz"] = int(opts.pop("control_hz"))
        if "starting_velocity" in opts:
            vopts["starting_velocity"] = tuple(opts.pop("starting_velocity"))
        if "starting_fuel_ratio" in opts:  # rocket.py:47
            vopts["starting_fuel_ratio"] = float(opts.pop("starting_fuel_ratio"))
        for k in ("use_camera", "use_gimbal", "camera_fps"):
            opts.pop(k, None)
        vopts.update(opts)
        P = build_params(drone_type, "none", noise="philox" if motor_noise else "off", autoreset="off",
                         seed=0 if seed is None else int(seed), vehicle_options=vopts,
                         world_options={"physics_hz": self.physics_hz, "world_scale": float(world_scale)})
        self._seed = 0 if seed is None else int(seed)
        self.engine = BatchEngine(P, self.num_drones, device=self.device, lane_offset=lane_offset)
        if per_drone_hz is not None:
            self.engine.ctrl_ratio = torch.tensor([self.physics_hz // hz for hz in per_drone_hz], dtype=torch.int32, device=self.device)
        if per_drone_vel is not None:
            # world-frame linear velocity handed to resetBaseVelocity as is (fixedwing.py:201)
            self.engine.start_vel = torch.tensor(per_drone_vel, dtype=torch.float32, device=self.device).contiguous()
        pose = np.concatenate([start_pos, np.stack([quat_from_euler(o) for o in start_orn])], axis=1)
        self._start_pose = torch.tensor(pose, dtype=torch.float32, device=self.device).contiguous()
        self.start_pos, self.start_orn = start_pos, start_orn
        self.updates_per_step = P.ticks_per_control  # core/aviary.py:288-289
        self.step_period = 1.0 / (self.physics_hz / P.ticks_per_control)
        self._sp_dim = 7 if drone_type == "rocket" else 4  # rocket.py:228
        self.setpoints = torch.zeros(self.num_drones, self._sp_dim, dtype=torch.float32, device=self.device)
        self.reset()

    # ------------------------------------------------------------------ core/aviary.py:218-312
    def reset(self) -> None:
        self.physics_steps = 0
        self.aviary_steps = 0
        self.elapsed_time = 0.0
        self.engine.state.zero_()
        self.engine.armed = None   # core/aviary.py:305-307: arm everything
        self.engine.modes = None   # drone.reset() -> set_mode(0) on every drone
        self.engine.aviary_reset(self._start_pose)
        self.mode = 0
        self._set_sp_dim(7 if self.drone_type == "rocket" else 4)
        self.setpoints.zero_()
        self._contact_acc = None
        self._controller = None
        self._base_sp_dim = self._sp_dim
        # wind field given to the constructor (core/aviary.py:266-285): built per reset with the
        # Aviary's generator and options, and sampled by the reset's update_state at time 0
        self.wind_field = None
        self._wind = None
        if self.wind_type is not None:
            field = self.wind_type(np_random=np.random.default_rng(self._seed), **self.wind_options)
            self._check_wind_field_validity(field)
            self.wind_field = field
            self._wind = self._sample_wind()

    # ------------------------------------------------------------------ :324-333, base_wind_field.py:57-69
    def _check_wind_field_validity(self, wind_field) -> None:
        test = wind_field(0.0, torch.tensor([[0.0, 0.0, 1.0]] * 5, dtype=torch.float32, device=self.device))
        assert torch.is_tensor(test), f"Returned wind velocity must be a torch.Tensor, got {type(test)}."
        assert test.is_floating_point(), f"Returned wind velocity must be type float, got {test.dtype}."
        assert tuple(test.shape) == (5, 3), f"Returned wind velocity must be array of shape (n, 3), got {tuple(test.shape)}."

    def register_wind_field_function(self, wind_field) -> None:
        """`wind_field(time: float, position: Tensor[M, 3]) -> Tensor[M, 3]`, device tensors. As in the
        reference, a field registered after construction first acts on the tick after the next
        update_state: the velocities left by reset() are wind-free."""
        assert callable(wind_field), "`wind_field` function must be callable."
        self._check_wind_field_validity(wind_field)
        self.wind_field = wind_field
        if self._wind is None:
            self.engine._aviary_outputs()
            self._wind = torch.zeros(self.num_drones, self.engine.wind_links, 3, dtype=torch.float32, device=self.device)

    def _sample_wind(self) -> torch.Tensor:
        pos = self.engine.link_pos
        w = self.wind_field(self.elapsed_time, pos.view(-1, 3))
        return w.to(dtype=torch.float32).reshape(pos.shape).contiguous()

    def _set_sp_dim(self, d):
        if d != self._sp_dim:
            self._sp_dim = d
            self.setpoints = torch.zeros(self.num_drones, d, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ quadx.py:375-399, base_drone.py:261-283
    def register_controller(self, controller_id: int, controller_constructor, base_mode: int) -> None:
        """Custom controllers (tests/test_core.py:141-192): `controller_constructor()` must give an object
        with `reset()` and `step(state, setpoint)`, here called ONCE PER CONTROL STEP FOR THE WHOLE BATCH with
        device tensors -- state [N, 4, 3] (ang_vel, ang_pos, lin_vel, lin_pos rows), setpoint [N, k] -- and
        returning the [N, k_base] setpoints of `base_mode`, which the kernels then run."""
        lo, hi = {"quadx": (-1, 7), "fixedwing": (-1, 0), "rocket": (0, 0)}[self.drone_type]
        assert controller_id < lo or controller_id > hi, f"`controller_id` must not be a default flight mode ({lo}..{hi}), got {controller_id}."
        assert lo <= base_mode <= hi, f"`base_mode` must be a default flight mode ({lo}..{hi}), got {base_mode}."
        if self.engine.ctrl_ratio is not None:
            raise NotImplementedError("custom controllers with per-drone control rates")
        self._registered_controllers[int(controller_id)] = controller_constructor
        self._registered_base_modes[int(controller_id)] = int(base_mode)

    @property
    def drones(self):
        """`env.drones[i].register_controller(...)` as in the reference: the registration is batch-wide."""
        return [_DroneProxy(self, i) for i in range(self.num_drones)]

    # ------------------------------------------------------------------ :440-478
    def set_mode(self, flight_modes: int) -> None:
        if isinstance(flight_modes, (list, tuple, np.ndarray)):
            assert len(flight_modes) == self.num_drones, f"Expected {self.num_drones} flight_modes, got {len(flight_modes)}."
            if len(set(int(m) for m in flight_modes)) == 1:
                flight_modes = int(flight_modes[0])
            else:
                return self._set_modes_per_drone([int(m) for m in flight_modes])
        self.engine.modes = None
        mode = int(flight_modes)
        self._controller = None
        if mode in self._registered_controllers:  # quadx.py:266-269: instantiate, then behave as the base mode
            self._controller = self._registered_controllers[mode]()
            mode = self._registered_base_modes[mode]
        lo, hi = {"quadx": (-1, 7), "fixedwing": (-1, 0), "rocket": (0, 0)}[self.drone_type]
        if mode < lo or mode > hi:
            raise ValueError(f"`mode` must be between {lo} and {hi}, got {mode}.")  # quadx.py:260-263
        self._set_sp_dim(7 if self.drone_type == "rocket" else (6 if (self.drone_type == "fixedwing" and mode == -1) else 4))
        self.engine.aviary_set_mode(mode, self.setpoints)
        self.mode = mode
        self._base_sp_dim = self._sp_dim

    def _set_modes_per_drone(self, modes) -> None:
        """A different flight mode per drone (core/aviary.py:449-455): QuadX only -- every QuadX mode takes four
        setpoint values; the modes live in a per-lane buffer the Aviary-level kernels read."""
        if self.drone_type != "quadx":
            raise NotImplementedError("per-drone flight modes are supported for QuadX (Fixedwing modes differ in setpoint width)")
        if any(m in self._registered_controllers for m in modes):
            raise NotImplementedError("custom controllers with per-drone flight modes")
        for m in modes:
            if m < -1 or m > 7:
                raise ValueError(f"`mode` must be between -1 and 7, got {m}.")
        self._controller = None
        self._set_sp_dim(4)
        self.engine.modes = torch.tensor(modes, dtype=torch.int32, device=self.device)
        self.engine.aviary_set_mode(0, self.setpoints)
        self.mode = list(modes)
        self._base_sp_dim = 4

    def set_setpoint(self, index: int, setpoint) -> None:
        self.setpoints[index] = torch.as_tensor(np.asarray(setpoint), dtype=torch.float32, device=self.device)

    def set_all_setpoints(self, setpoints) -> None:
        sp = setpoints if torch.is_tensor(setpoints) else torch.as_tensor(np.asarray(setpoints))
        self.setpoints.copy_(sp.to(device=self.device, dtype=torch.float32).reshape(self.num_drones, self._sp_dim))

    def set_armed(self, settings) -> None:  # core/aviary.py:423-438
        """Arm / disarm drones: a disarmed drone gets no controller update, no forces and no state read-back
        (its `state` / `aux_state` keep their last values), but still falls under gravity."""
        if isinstance(settings, (list, tuple, np.ndarray)):
            assert len(settings) == self.num_drones, f"Expected {self.num_drones} settings, got {len(settings)}."
            flags = [bool(x) for x in settings]
        else:
            flags = [bool(settings)] * self.num_drones
        self.engine.armed = None if all(flags) else torch.tensor(flags, dtype=torch.bool, device=self.device)

    # ------------------------------------------------------------------ :480-531
    def step(self, n_steps: int = 1) -> None:
        """One (or n fused) `Aviary.step()`: control + ticks_per_control physics ticks per drone."""
        if self.wind_field is not None:
            return self._step_with_wind(n_steps)
        self._contact_acc = None
        if self._controller is not None:  # quadx.py:417-429: the custom controller runs first, on the last update_state's state
            for _ in range(n_steps):
                sp = self._controller.step(self.all_states, self.setpoints)
                sp = sp.to(device=self.device, dtype=torch.float32).reshape(self.num_drones, -1).contiguous()
                assert sp.shape[1] == self._base_sp_dim, f"custom controller outputting wrong shape, expected (N, {self._base_sp_dim}) but got {tuple(sp.shape)}."
                self.engine.aviary_step(sp, n_steps=1)
            self.physics_steps += n_steps * self.updates_per_step
            self.aviary_steps += n_steps
            self.elapsed_time = self.physics_steps / self.physics_hz
            return
        self.engine.aviary_step(self.setpoints, n_steps=n_steps)
        self.physics_steps += n_steps * self.updates_per_step
        self.aviary_steps += n_steps
        self.elapsed_time = self.physics_steps / self.physics_hz

    def _step_with_wind(self, n_steps: int) -> None:
        # the reference's per-tick order (aviary.py:510-531): control + forces (with the wind sampled by
        # the previous update_state) -> stepSimulation -> update_state (samples the field at the new link
        # positions with the not-yet-advanced elapsed_time) -> contact splice -> counters
        for _ in range(n_steps):
            acc = torch.zeros_like(self.engine.out_contact)
            sp = self.setpoints
            if self._controller is not None:
                # quadx.py:417-429: a registered custom controller runs at the control tick (tick 0 of the Aviary step) on the
                # state the last update_state left, wind field or not; its output is the base mode's setpoint for this step
                sp = self._controller.step(self.all_states, self.setpoints)
                sp = sp.to(device=self.device, dtype=torch.float32).reshape(self.num_drones, -1).contiguous()
                assert sp.shape[1] == self._base_sp_dim, f"custom controller outputting wrong shape, expected (N, {self._base_sp_dim}) but got {tuple(sp.shape)}."
            for t in range(self.updates_per_step):
                self.engine.aviary_tick(sp, t, wind=self._wind)
                acc |= self.engine.out_contact
                self._wind = self._sample_wind()
                self.physics_steps += 1
                self.elapsed_time = self.physics_steps / self.physics_hz
            self._contact_acc = acc
            self.aviary_steps += 1

    # ------------------------------------------------------------------ :335-421
    @property
    def all_states(self) -> torch.Tensor:
        """[N, 4, 3]: ang_vel, ang_pos, lin_vel (body frame), lin_pos rows, as DroneClass.state."""
        return self.engine.out_state.view(self.num_drones, 4, 3)

    @property
    def all_aux_states(self) -> torch.Tensor:
        return self.engine.out_aux

    def state(self, index: int) -> torch.Tensor:
        return self.all_states[index]

    def aux_state(self, index: int) -> torch.Tensor:
        return self.all_aux_states[index]

    @property
    def contact_array(self) -> torch.Tensor:
        return self.engine.out_contact if self._contact_acc is None else self._contact_acc

    def disconnect(self) -> None:
        self.engine.close()

    close = disconnect
