"""BatchEngine: owns the PyTorch-ROCm tensors of one batched simulation and drives the HIP kernels
through the C ABI (include/pyflyt_amd.h). PyTorch is plumbing here -- device memory and streams;
all arithmetic happens in libpyflyt_amd.so.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


try:  # the raw handle of torch's current stream without building a torch.cuda.Stream object (0.2 us against 1.5 us per call)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream

_PREPARED_MAX, _PREPARED_BYTES = 256, 256 << 20  # action tensors whose prepared buffer blocks an engine keeps alive, at most (an action ring; a policy's output buffer)


class BatchEngine:
    """N independent drones ("lanes") on one GPU.

    Tensors (all on `device`):
      state      [groups, n, 4] float32  persistent SoA state (float4 groups)
      obs        [n, obs_dim]   float32
      final_obs  [n, obs_dim]   float32  (SAME_STEP auto-reset only)
      reward     [n]            float32
      terminated [n], truncated [n]  bool
    """

    def __init__(self, params: L.PfParams, num_lanes: int, device="cuda:0", lane_offset: int = 0):
        if not torch.cuda.is_available():
            raise L.PyFlytAmdError("pyflyt_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.lib = L.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.PyFlytAmdError(f"device must be a ROCm 'cuda' device, got {device}")
        self.params = params
        self.n = int(num_lanes)
        self.lane_offset = int(lane_offset)
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._ctx = C.c_void_p()
        L.check(self.lib.pf_ctx_create(C.byref(params), self.n, index, self.lane_offset, C.byref(self._ctx)))
        self.groups = self.lib.pf_state_groups(self._ctx)
        self.obs_dim = self.lib.pf_obs_dim(self._ctx)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.state = torch.zeros(self.groups, self.n, 4, **f32)
        self.obs = torch.zeros(self.n, self.obs_dim, **f32)
        self.final_obs = torch.zeros(self.n, self.obs_dim, **f32) if params.autoreset == L.AUTORESET_SAME_STEP else None
        # [n, 2] int32 (flags, targets left) of the episode that just ended, before the SAME_STEP re-initialisation
        self.final_info = torch.zeros(self.n, 2, dtype=torch.int32, device=self.device) if params.autoreset == L.AUTORESET_SAME_STEP else None
        self.reward = torch.zeros(self.n, **f32)
        self.terminated = torch.zeros(self.n, dtype=torch.bool, device=self.device)
        self.truncated = torch.zeros(self.n, dtype=torch.bool, device=self.device)
        self.out_state = None
        self.link_pos = None
        self.wind_links = 0
        self.ctrl_ratio = None  # [n] int32: physics ticks per controller update of each drone, or None = uniform
        self.modes = None       # [n] int32: per-drone flight modes (QuadX), or None = the context's mode
        self.start_vel = None   # [n, 3] float32: per-drone spawn velocity for aviary_reset, or None = params
        self.armed = None       # [n] bool: Aviary.set_armed, or None = all armed
        self.out_aux = None
        self.out_contact = None
        self._buf = L.PfBuffers()
        self._index = index
        # env_step's hot path: {id(action tensor): (the tensor, its filled pf_buffers block)} -- see env_step
        self._prepared: dict[int, tuple] = {}
        self._step_fn = self.lib.pf_env_step

    def close(self):
        if self._ctx:
            self.lib.pf_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _buffers(self, actions=None, xi=None, xi_reset=None, u_targets=None, setpoints=None, start_pose=None, wind=None,
                 wrench=None, actions_out=None):
        # (ctrl_ratio: per-drone control rate, set once by the batched Aviary)
        b = self._buf
        b.state = _ptr(self.state)
        b.actions = _ptr(actions)
        b.obs = _ptr(self.obs)
        b.final_obs = _ptr(self.final_obs)
        b.reward = _ptr(self.reward)
        b.terminated = _ptr(self.terminated)
        b.truncated = _ptr(self.truncated)
        b.xi = _ptr(xi)
        b.xi_reset = _ptr(xi_reset)
        b.u_targets = _ptr(u_targets)
        b.setpoints = _ptr(setpoints)
        b.out_state = _ptr(self.out_state)
        b.out_aux = _ptr(self.out_aux)
        b.out_contact = _ptr(self.out_contact)
        b.start_pose = _ptr(start_pose)
        b.wind = _ptr(wind)
        b.out_link_pos = _ptr(self.link_pos)
        b.ctrl_ratio = _ptr(self.ctrl_ratio)
        b.modes = _ptr(self.modes)
        b.start_vel = _ptr(self.start_vel)
        b.armed = _ptr(self.armed)
        b.final_info = _ptr(self.final_info)
        b.actions_out = _ptr(actions_out)
        b.wrench = _ptr(wrench)
        return b

    def _check_f32(self, t, shape, name):
        if t is None:
            return None
        if t.dtype != torch.float32 or t.device != self.device or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} must be a contiguous float32 tensor of shape {tuple(shape)} on {self.device}, "
                             f"got {t.dtype} {tuple(t.shape)} on {t.device}")
        return t

    def _check(self, t, shape, dtypes, name):
        """Every tensor whose data_ptr() crosses the C ABI is checked here: the kernels index raw pointers, so a wrongly
        sized, typed, placed or strided tensor would read or write out of bounds silently."""
        if t is None:
            return None
        if t.dtype not in dtypes or t.device != self.device or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} must be a contiguous {'/'.join(str(d) for d in dtypes)} tensor of shape {tuple(shape)} on {self.device}, "
                             f"got {t.dtype} {tuple(t.shape)} on {t.device}")
        return t

    def _sp_dim(self, mode=None):
        mode = self.params.flight_mode if mode is None else mode
        return 7 if self.params.vehicle == L.ROCKET else (6 if (self.params.vehicle == L.FIXEDWING and mode == -1) else 4)

    def _check_per_lane(self):
        self._check(self.ctrl_ratio, (self.n,), (torch.int32,), "ctrl_ratio")
        self._check(self.modes, (self.n,), (torch.int32,), "modes")
        self._check(self.start_vel, (self.n, 3), (torch.float32,), "start_vel")
        self._check(self.armed, (self.n,), (torch.bool, torch.uint8), "armed")

    @property
    def ticks_per_step(self):
        return self.params.env_step_ratio * self.params.ticks_per_control

    @property
    def settle_ticks(self):
        return self.params.settle_steps * self.params.ticks_per_control

    # ------------------------------------------------------------------ env level
    def env_reset(self, mask=None, xi_reset=None, u_targets=None):
        if mask is not None:
            if mask.dtype != torch.bool and mask.dtype != torch.uint8:
                raise ValueError("mask must be a bool/uint8 tensor")
            mask = mask.to(device=self.device).contiguous()
            self._check(mask, (self.n,), (torch.bool, torch.uint8), "mask")
            # (shared worlds: the agents of a world are reset together; the kernels widen a partial selection to the whole world --
            #  no host-side check, which would cost a device synchronisation per masked reset)
        self._check_targets(u_targets)
        self._check_f32(xi_reset, (self.settle_ticks, self.n), "xi_reset")
        b = self._buffers(xi_reset=xi_reset, u_targets=u_targets)
        with torch.cuda.device(self.device):
            L.check(self.lib.pf_env_reset(self._ctx, C.byref(b), _ptr(mask), self._stream()), self._ctx)
        return self.obs

    @property
    def action_dim(self):
        """Width of the env action: 4, or 6 for the dogfight task with assisted_flight=False (pf_params.df_action_dim)."""
        return 6 if (self.params.task == L.TASK_DOGFIGHT and self.params.df_action_dim == 6) else 4

    def env_step(self, actions, xi=None, xi_reset=None, u_targets=None):
        """One env step of every lane: pf_env_step on torch's current stream. Results in self.obs / reward / terminated / truncated
        (the same tensors every call). The host cost of a call whose `actions` TENSOR OBJECT this engine has seen before -- a
        policy writing into a fixed buffer, an action ring -- is one dictionary lookup, the raw stream handle and the foreign call:
        the tensor checks and the ~25 pointer conversions are done once per tensor (the prepared block keeps the tensor alive, so
        its id cannot be recycled; its address is compared on every call). Capturable in a HIP graph (no host synchronisation, no allocation)."""
        if xi is None and xi_reset is None and u_targets is None:
            hit = self._prepared.get(id(actions))
            if hit is None or hit[3] != actions.data_ptr():  # (a tensor re-pointed in place -- t.data = ..., set_() -- is prepared again)
                hit = self._prepare(actions)
            rc = self._step_fn(self._ctx, hit[1], _raw_stream(self._index))  # (the library selects the context's device itself)
            if rc:
                L.check(rc, self._ctx)
            return self.obs, self.reward, self.terminated, self.truncated
        self._check_f32(actions, (self.n, self.action_dim), "actions")
        self._check_f32(xi, (self.ticks_per_step, self.n), "xi")
        self._check_f32(xi_reset, (self.settle_ticks, self.n), "xi_reset")
        self._check_targets(u_targets)
        b = self._buffers(actions=actions, xi=xi, xi_reset=xi_reset, u_targets=u_targets)
        with torch.cuda.device(self.device):
            L.check(self.lib.pf_env_step(self._ctx, C.byref(b), self._stream()), self._ctx)
        return self.obs, self.reward, self.terminated, self.truncated

    def _prepare(self, actions):
        if not torch.is_tensor(actions):
            raise ValueError(f"actions must be a float32 tensor of shape {(self.n, self.action_dim)} on {self.device}, got {type(actions).__name__}")
        self._check_f32(actions, (self.n, self.action_dim), "actions")
        b = L.PfBuffers()
        C.memmove(C.byref(b), C.byref(self._buffers(actions=actions)), C.sizeof(L.PfBuffers))
        if len(self._prepared) >= max(8, min(_PREPARED_MAX, _PREPARED_BYTES // (16 * self.n))):
            self._prepared.clear()
        hit = self._prepared[id(actions)] = (actions, C.byref(b), b, actions.data_ptr())
        return hit

    def prepare_step(self, actions):
        """A prepared env step: validates `actions` ([n, action_dim] float32 on the device) and fills the C buffer block ONCE,
        returns launch(stream_ptr) -- one pf_env_step call on that stream (a ctypes.c_void_p, e.g.
        C.c_void_p(torch.cuda.current_stream().cuda_stream)), results in self.obs / reward / terminated / truncated as for
        env_step. For launch-bound inner loops that re-use their action tensors (a policy writing into a fixed buffer): the
        per-call host cost drops from the tensor checks and ~25 pointer conversions of env_step to the one foreign call.
        PF_NOISE_PHILOX / PF_NOISE_OFF only (the injected-noise protocol passes new tensors every step)."""
        if self.params.noise_mode == L.NOISE_INJECT:
            raise ValueError("prepare_step: PF_NOISE_INJECT passes per-step noise tensors; use env_step")
        self._check_f32(actions, (self.n, self.action_dim), "actions")
        b = L.PfBuffers()
        C.memmove(C.byref(b), C.byref(self._buffers(actions=actions)), C.sizeof(L.PfBuffers))
        fn, ctx, ref, check = self.lib.pf_env_step, self._ctx, C.byref(b), L.check

        def launch(stream_ptr):
            rc = fn(ctx, ref, stream_ptr)
            if rc:
                check(rc, ctx)

        launch._keep = (b, actions)  # (the buffer block and the action tensor stay alive with the closure)
        return launch

    def _check_targets(self, u_targets):
        if u_targets is not None:
            rows = (4 if self.params.use_yaw_targets else 3) * self.params.num_targets
            self._check_f32(u_targets, (rows, self.n), "u_targets")

    def sample_actions(self, out, step_index: int):
        self._check_f32(out, (self.n, 4), "out")
        with torch.cuda.device(self.device):
            L.check(self.lib.pf_sample_actions(self._ctx, _ptr(out), int(step_index) & 0xFFFFFFFF, self._stream()), self._ctx)
        return out

    def rollout(self, k_steps: int, step_index0: int = 0, actions=None, store_actions: bool = True):
        """pf_rollout: `k_steps` env steps in one launch, the lanes' state resident in registers between them (every env kernel
        since round 4: the specialised ones, the generic one, the dogfight on either aircraft model). Returns the trajectory
        tensors (obs [k, n, D], reward [k, n], terminated [k, n], truncated [k, n], actions [k, n, 4] or None);
        bit-identical to k x (sample_actions(step_index0 + s) + env_step). `actions`: an open-loop sequence
        [k, n, 4] instead of on-device sampling."""
        k = int(k_steps)
        t = getattr(self, "_traj", None)
        if t is None or t["k"] != k:
            f32 = dict(dtype=torch.float32, device=self.device)
            t = dict(k=k, obs=torch.empty(k, self.n, self.obs_dim, **f32), reward=torch.empty(k, self.n, **f32),
                     terminated=torch.empty(k, self.n, dtype=torch.bool, device=self.device),
                     truncated=torch.empty(k, self.n, dtype=torch.bool, device=self.device),
                     actions=torch.empty(k, self.n, 4, **f32),
                     final_obs=torch.zeros(k, self.n, self.obs_dim, **f32) if self.final_obs is not None else None,
                     final_info=torch.zeros(k, self.n, 2, dtype=torch.int32, device=self.device) if self.final_info is not None else None)
            self._traj = t
        self._check_f32(actions, (k, self.n, self.action_dim), "actions")
        keep = store_actions and actions is None  # (the kernels sample in registers; the draws are written out only on request)
        b = self._buffers(actions=actions, actions_out=t["actions"] if keep else None)
        b.obs, b.reward, b.terminated, b.truncated = _ptr(t["obs"]), _ptr(t["reward"]), _ptr(t["terminated"]), _ptr(t["truncated"])
        b.final_obs, b.final_info = _ptr(t["final_obs"]), _ptr(t["final_info"])
        with torch.cuda.device(self.device):
            L.check(self.lib.pf_rollout(self._ctx, C.byref(b), k, int(step_index0) & 0xFFFFFFFF, self._stream()), self._ctx)
        return t["obs"], t["reward"], t["terminated"], t["truncated"], (t["actions"] if actions is None and store_actions else actions)

    def body_tick(self, wrench, n_ticks: int = 1):
        """pf_body_tick: the free-body tick alone under a held body-frame wrench [n, 6] (force, torque)."""
        self._aviary_outputs()
        self._check_f32(wrench, (self.n, 6), "wrench")
        b = self._buffers(wrench=wrench)
        with torch.cuda.device(self.device):
            L.check(self.lib.pf_body_tick(self._ctx, C.byref(b), int(n_ticks), self._stream()), self._ctx)
        return self.out_state

    # ------------------------------------------------------------------ Aviary level
    def _aviary_outputs(self):
        if self.out_state is None:
            aux = {L.QUADX: 4, L.FIXEDWING: 6, L.ROCKET: 9}[self.params.vehicle]
            self.out_state = torch.zeros(self.n, 12, dtype=torch.float32, device=self.device)
            self.out_aux = torch.zeros(self.n, aux, dtype=torch.float32, device=self.device)
            self.out_contact = torch.zeros(self.n, dtype=torch.bool, device=self.device)
            # world positions of the links a wind field is sampled at (QuadX: body link; Fixedwing: 5 surfaces)
            self.wind_links = int(self.lib.pf_wind_links(self._ctx))
            self.link_pos = torch.zeros(self.n, self.wind_links, 3, dtype=torch.float32, device=self.device)

    def aviary_reset(self, start_pose=None):
        self._aviary_outputs()
        self._check_f32(start_pose, (self.n, 7), "start_pose")
        self._check_per_lane()
        b = self._buffers(start_pose=start_pose)
        with torch.cuda.device(self.device):
            L.check(self.lib.pf_aviary_reset(self._ctx, C.byref(b), self._stream()), self._ctx)
        self.params.flight_mode = 0

    def aviary_set_mode(self, mode: int, setpoints):
        self._aviary_outputs()
        self._check_f32(setpoints, (self.n, self._sp_dim(int(mode))), "setpoints")
        self._check_per_lane()
        b = self._buffers()
        with torch.cuda.device(self.device):
            L.check(self.lib.pf_aviary_set_mode(self._ctx, C.byref(b), int(mode), _ptr(setpoints), self._stream()), self._ctx)
        self.params.flight_mode = int(mode)

    def aviary_step(self, setpoints, n_steps: int = 1, xi=None):
        self._aviary_outputs()
        self._check_f32(setpoints, (self.n, self._sp_dim()), "setpoints")
        self._check_f32(xi, (int(n_steps) * self.params.ticks_per_control, self.n), "xi")
        self._check_per_lane()
        b = self._buffers(setpoints=setpoints, xi=xi)
        with torch.cuda.device(self.device):
            L.check(self.lib.pf_aviary_step(self._ctx, C.byref(b), int(n_steps), self._stream()), self._ctx)
        return self.out_state, self.out_aux

    def aviary_tick(self, setpoints, tick_index: int, wind=None, xi=None):
        """One physics tick of Aviary.step (pf_aviary_tick). `wind`: [n, wind_links, 3] world-frame wind
        as sampled after the previous tick, or None. Fills out_state / out_aux / out_contact (this
        tick's contact verdict) / link_pos (where to sample the field next)."""
        self._aviary_outputs()
        self._check_f32(wind, (self.n, self.wind_links, 3), "wind")
        self._check_f32(xi, (self.n,), "xi")
        self._check_f32(setpoints, (self.n, self._sp_dim()), "setpoints")
        self._check_per_lane()
        b = self._buffers(setpoints=setpoints, xi=xi, wind=wind)
        with torch.cuda.device(self.device):
            L.check(self.lib.pf_aviary_tick(self._ctx, C.byref(b), int(tick_index), self._stream()), self._ctx)
        return self.out_state, self.out_aux

    # ------------------------------------------------------------------ state views
    def ints(self):
        """[n, 4] int32 view: step_count, flags, rng_ctr, n_targets_left."""
        g = 5 if self.params.vehicle == L.FIXEDWING else 6
        return self.state[g].view(torch.int32)

    def flags(self):
        return self.ints()[:, 1]
