"""oracle.py -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libuav_oracle.so (fp64 CPU
restatement, see uav_oracle.h). May be imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by pyflyt_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libuav_oracle.so")

MAX_TARGETS, MAX_BOXES, MAX_SURF, MAX_LINKS = 8, 12, 5, 12
QUADX, FIXEDWING = 0, 1
TASK_NONE, TASK_HOVER, TASK_WAYPOINTS, TASK_MA_HOVER = 0, 1, 2, 3
NOISE_OFF, NOISE_INJECT, NOISE_PHILOX = 0, 1, 2

d3 = C.c_double * 3
d33 = d3 * 3


class World(C.Structure):
    _fields_ = [
        ("dt", C.c_double), ("gravity_z", C.c_double), ("use_gyro_term", C.c_int),
        ("max_coord_vel", C.c_double), ("plane_half_xy", C.c_double), ("plane_half_z", C.c_double),
        ("ticks_per_control", C.c_int),
        ("contact_response", C.c_int), ("contact_restitution", C.c_double), ("contact_friction", C.c_double),
        ("contact_erp", C.c_double), ("contact_iters", C.c_int), ("contact_margin", C.c_double), ("contact_slop", C.c_double), ("pair_response", C.c_int),
        ("contact_report_distance", C.c_double), ("contact_residual_threshold", C.c_double), ("contact_manifold_points", C.c_int), ("contact_break_distance", C.c_double),
    ]


class PidGains(C.Structure):
    _fields_ = [("kp", d3), ("ki", d3), ("kd", d3), ("lim", d3)]


class Box(C.Structure):
    _fields_ = [("c", d3), ("h", d3), ("kind", C.c_int), ("yaw", C.c_double)]


class Surface(C.Structure):
    _fields_ = [
        ("r", d3), ("lift_unit", d3), ("drag_unit", d3), ("torque_unit", d3),
        ("Cl_alpha_2D", C.c_double), ("chord", C.c_double), ("span", C.c_double),
        ("flap_to_chord", C.c_double), ("eta", C.c_double),
        ("alpha_0_base", C.c_double), ("alpha_stall_P_base", C.c_double), ("alpha_stall_N_base", C.c_double),
        ("Cd_0", C.c_double), ("deflection_limit", C.c_double), ("tau", C.c_double),
        ("half_rho", C.c_double), ("area", C.c_double), ("aspect", C.c_double),
        ("Cl_alpha_3D", C.c_double), ("theta_f", C.c_double), ("aero_tau", C.c_double),
    ]


class Params(C.Structure):
    _fields_ = [
        ("vehicle", C.c_int), ("world", World),
        ("mass", C.c_double), ("com", d3), ("I_own", d33), ("I_pa", d33), ("I_inv", d33),
        ("n_boxes", C.c_int), ("boxes", Box * MAX_BOXES), ("bound_radius", C.c_double),
        ("n_motors", C.c_int), ("motor_r", d3 * 4), ("thrust_unit", d3 * 4),
        ("thrust_coef", C.c_double * 4), ("torque_coef", C.c_double * 4), ("max_rpm", C.c_double * 4),
        ("motor_tau", C.c_double * 4), ("noise_ratio", C.c_double * 4),
        ("motor_map", (C.c_double * 4) * 4), ("drag_const", d3), ("drag_coef_pqr", C.c_double),
        ("pid", PidGains * 4), ("zpid", PidGains * 2), ("control_period", C.c_double),
        ("n_surf", C.c_int), ("surf", Surface * MAX_SURF), ("assist_ids", C.c_int * 6),
        ("assist_signs", C.c_double * 6),
        ("task", C.c_int), ("flight_mode", C.c_int),
        ("start_pos", d3), ("start_rpy", d3), ("start_vel", d3),
        ("dome", C.c_double), ("max_steps", C.c_int), ("env_step_ratio", C.c_int),
        ("settle_steps", C.c_int), ("sparse_reward", C.c_int), ("angle_repr", C.c_int),
        ("num_targets", C.c_int), ("goal_reach_distance", C.c_double), ("min_height", C.c_double),
        ("throttle_remap", C.c_double), ("collide_any", C.c_int),
        ("wp_dist_reward", C.c_double), ("wp_yaw_penalty", C.c_double),
        ("use_yaw_targets", C.c_int), ("goal_reach_angle", C.c_double),
        ("noise_mode", C.c_int), ("seed", C.c_uint64),
        ("n_links", C.c_int), ("link_mass", C.c_double * MAX_LINKS), ("link_r", d3 * MAX_LINKS), ("link_I", d3 * MAX_LINKS),
        ("fueltank_link", C.c_int), ("booster_link", C.c_int),
        ("total_fuel", C.c_double), ("max_fuel_rate", C.c_double), ("fuel_inertia", d3), ("min_thrust", C.c_double),
        ("max_thrust", C.c_double), ("booster_tau", C.c_double), ("booster_noise", C.c_double), ("reignitable", C.c_int),
        ("gimbal_tau", C.c_double), ("gimbal_range_rad", C.c_double), ("finlet_map", d3 * 4),
        ("starting_fuel_ratio", C.c_double),
        ("wind_fn", C.c_void_p),
    ]


WIND_FN = C.CFUNCTYPE(None, C.c_double, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double))


def set_wind(params: "Params", fn):
    """Attach a wind field `fn(time: float, position: (n, 3) ndarray) -> (n, 3) ndarray` (the signature of
    the reference's WindFieldClass.__call__, base_wind_field.py:46-55) to an oracle parameter block.
    Returns the ctypes callback object, which the caller must keep alive while the params are in use.
    The callback enters Python: step such lanes serially (n_threads irrelevant for OracleBatch of wind
    tests, which loops in Python)."""
    if fn is None:
        params.wind_fn = None
        return None

    def _cb(t, pos, n, out):
        p = np.ctypeslib.as_array(pos, shape=(n, 3)).copy()
        w = np.asarray(fn(float(t), p), dtype=np.float64)
        assert w.shape == (n, 3), w.shape
        np.ctypeslib.as_array(out, shape=(n, 3))[:] = w

    cb = WIND_FN(_cb)
    params.wind_fn = C.cast(cb, C.c_void_p)
    return cb


class Lane(C.Structure):
    _fields_ = [
        ("p", d3), ("q", C.c_double * 4), ("v", d3), ("w", d3),
        ("w_b", d3), ("rpy", d3), ("v_b", d3), ("surf_v", d3 * MAX_SURF), ("drag_v_b", d3),
        ("throttle", C.c_double * 4), ("actuation", C.c_double * MAX_SURF),
        ("pwm", C.c_double * 4), ("cmd", C.c_double * 8), ("setpoint", C.c_double * 8),
        ("fuel_ratio", C.c_double), ("ignition", C.c_int), ("gimbal", C.c_double * 2),
        ("pid_I", d3 * 4), ("pid_E", d3 * 4), ("zpid_I", C.c_double * 2), ("zpid_E", C.c_double * 2),
        ("mode", C.c_int), ("physics_steps", C.c_int), ("contact_now", C.c_int), ("contact_step", C.c_int),
        ("world_contact", C.c_int), ("peer_contact", C.c_int),
        ("step_count", C.c_int), ("terminated", C.c_int), ("truncated", C.c_int),
        ("info_oob", C.c_int), ("info_collision", C.c_int), ("info_complete", C.c_int),
        ("num_targets_reached", C.c_int),
        ("reward", C.c_double), ("action", C.c_double * 4), ("past_action", C.c_double * 4),
        ("targets", d3 * MAX_TARGETS), ("yaw_targets", C.c_double * MAX_TARGETS), ("yaw_error_scalar", C.c_double),
        ("n_targets_left", C.c_int),
        ("new_dist", C.c_double), ("old_dist", C.c_double),
        ("obs", C.c_double * 48),
        ("rng_ctr", C.c_uint32), ("reset_key", C.c_uint32), ("lane_id", C.c_uint64),
    ]


DF_MAX = 8
_dfm = (C.c_double * DF_MAX) * DF_MAX
_ifm = (C.c_int * DF_MAX) * DF_MAX


class Dogfight(C.Structure):
    """orc_dogfight (uav_oracle.h): env-level state of one MAFixedwingDogfightEnv world."""
    _fields_ = [
        ("A", C.c_int), ("team_size", C.c_int),
        ("damage_per_hit", C.c_double), ("lethal_distance", C.c_double), ("lethal_angle", C.c_double),
        ("aggressiveness", C.c_double), ("cooperativeness", C.c_double), ("sparse_reward", C.c_int),
        ("dome", C.c_double), ("max_steps", C.c_int), ("env_step_ratio", C.c_int), ("action_dim", C.c_int),
        ("step_count", C.c_int), ("alive", C.c_int * DF_MAX), ("health", C.c_double * DF_MAX),
        ("received_hits", C.c_int * DF_MAX), ("inactive", C.c_int * DF_MAX),
        ("cur_dist", _dfm), ("cur_ang", _dfm), ("prev_dist", _dfm), ("prev_ang", _dfm),
        ("cur_hit", _ifm), ("in_range", _ifm), ("chasing", _ifm),
        ("other_att", ((C.c_double * 12) * DF_MAX) * DF_MAX),
        ("acc_reward", C.c_double * DF_MAX), ("acc_term", C.c_int * DF_MAX), ("acc_trunc", C.c_int * DF_MAX),
        ("info_bits", C.c_int * DF_MAX),
        ("action", (C.c_double * 6) * DF_MAX), ("past_action", (C.c_double * 6) * DF_MAX),
        ("obs", (C.c_double * (25 + (DF_MAX - 1) * 14)) * DF_MAX),
        ("reward", C.c_double * DF_MAX), ("terminated", C.c_int * DF_MAX), ("truncated", C.c_int * DF_MAX),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "uav_oracle.c")
    hdr = os.path.join(_HERE, "uav_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(_LIB_PATH) for f in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        assert L.orc_sizeof_params() == C.sizeof(Params), (L.orc_sizeof_params(), C.sizeof(Params))
        assert L.orc_sizeof_lane() == C.sizeof(Lane), (L.orc_sizeof_lane(), C.sizeof(Lane))
        dp, PP, LP = C.POINTER(C.c_double), C.POINTER(Params), C.POINTER(Lane)
        L.orc_pid_step.argtypes = [dp, dp, dp, dp, C.c_double, C.c_int, dp, dp, dp, dp, dp]
        L.orc_quadx_mix.argtypes = [PP, dp, dp]
        L.orc_quadx_control.argtypes = [PP, LP]
        L.orc_surface_aero.argtypes = [C.POINTER(Surface), C.c_double, C.c_double, dp]
        L.orc_surface_force.argtypes = [C.POINTER(Surface), dp, C.c_double, dp, dp]
        L.orc_quat_from_euler.argtypes = [dp, dp]
        L.orc_euler_from_quat.argtypes = [dp, dp]
        L.orc_matrix_from_quat.argtypes = [dp, dp]
        L.orc_contact_plane.argtypes = [PP, dp, dp]
        L.orc_contact_plane.restype = C.c_int
        L.orc_rigid_tick.argtypes = [PP, dp, dp, dp, dp, dp, dp]
        L.orc_philox4x32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_philox4x32_r.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        L.orc_normal4.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, dp]
        L.orc_uniform4.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, dp]
        L.orc_update_state.argtypes = [PP, LP]
        L.orc_set_mode.argtypes = [PP, LP, C.c_int]
        L.orc_aviary_reset.argtypes = [PP, LP, C.c_uint64]
        L.orc_aviary_step.argtypes = [PP, LP, dp, C.c_uint32, C.c_uint32]
        L.orc_env_reset.argtypes = [PP, LP, C.c_uint64, dp, dp]
        PPP, LPP, dpp = C.POINTER(PP), C.POINTER(LP), C.POINTER(dp)
        L.orc_world_aviary_step.argtypes = [PPP, LPP, C.c_int, dpp, C.c_uint32, C.c_uint32]
        L.orc_world_env_reset.argtypes = [PPP, LPP, C.c_int, C.c_uint64, dpp]
        L.orc_world_env_step.argtypes = [PPP, LPP, C.c_int, dp, dpp]
        assert L.orc_sizeof_dogfight() == C.sizeof(Dogfight), (L.orc_sizeof_dogfight(), C.sizeof(Dogfight))
        DP = C.POINTER(Dogfight)
        L.orc_dogfight_spawn.argtypes = [C.c_int, C.c_double, C.c_double, dp, dp, dp, dp]
        L.orc_dogfight_reset.argtypes = [PPP, LPP, DP, C.c_uint64, dpp]
        L.orc_dogfight_step.argtypes = [PPP, LPP, DP, dp, dpp]
        L.orc_env_step.argtypes = [PP, LP, dp, dp]
        L.orc_obs_dim.argtypes = [PP]
        L.orc_env_obs.argtypes = [PP, LP, dp]
        L.orc_env_reset_batch.argtypes = [PP, LP, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_env_step_batch.argtypes = [PP, LP, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        for f in ("orc_params_quadx", "orc_params_fixedwing", "orc_params_primitive_drone", "orc_params_rocket", "orc_params_acrowing", "orc_task_hover", "orc_task_quadx_waypoints",
                  "orc_task_fixedwing_waypoints", "orc_task_ma_hover", "orc_finalize"):
            getattr(L, f).argtypes = [PP]
        _lib = L
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _vp(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def make_params(env: str, noise_mode: int = NOISE_OFF, seed: int = 0, **overrides) -> Params:
    """env in {'quadx', 'fixedwing', 'hover', 'quadx_waypoints', 'fixedwing_waypoints'}."""
    L = lib()
    P = Params()
    if env == "primitive_drone":
        L.orc_params_primitive_drone(C.byref(P))
    elif env == "rocket":
        L.orc_params_rocket(C.byref(P))
    elif env == "acrowing":
        L.orc_params_acrowing(C.byref(P))
    elif env in ("quadx", "hover", "quadx_waypoints", "ma_hover"):
        L.orc_params_quadx(C.byref(P))
    else:
        L.orc_params_fixedwing(C.byref(P))
    if env == "hover":
        L.orc_task_hover(C.byref(P))
    elif env == "quadx_waypoints":
        L.orc_task_quadx_waypoints(C.byref(P))
    elif env == "fixedwing_waypoints":
        L.orc_task_fixedwing_waypoints(C.byref(P))
    elif env == "ma_hover":
        L.orc_task_ma_hover(C.byref(P))
    P.noise_mode = noise_mode
    P.seed = seed
    for k, v in overrides.items():
        # (a ctypes Structure takes ANY attribute: a misspelt override would be dropped without a word)
        if not (hasattr(World, k[6:]) if k.startswith("world_") else hasattr(Params, k)):
            raise AttributeError(f"oracle.make_params: no such field: {k}")
        if k.startswith("world_"):
            setattr(P.world, k[6:], v)
        elif isinstance(v, (list, tuple, np.ndarray)):
            arr = getattr(P, k)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            setattr(P, k, v)
    L.orc_finalize(C.byref(P))
    return P


class OracleWorld:
    """A drones in ONE world (the PettingZoo envs): drone-drone contact detection and the world-global rotational-drag
    gate couple them; everything else is per drone. params: one block per drone (its own spawn pose)."""

    def __init__(self, params_list, lane_id0: int = 0):
        self.A = len(params_list)
        self.Ps = list(params_list)
        self.Ls = [Lane() for _ in range(self.A)]
        self.lane_id0 = lane_id0
        PP, LP = C.POINTER(Params), C.POINTER(Lane)
        self._pp = (PP * self.A)(*[C.pointer(p) for p in self.Ps])
        self._lp = (LP * self.A)(*[C.pointer(l) for l in self.Ls])
        self.obs_dim = lib().orc_obs_dim(C.byref(self.Ps[0]))

    def _rows(self, x):
        """[A, k] array of injected draws -> array of A row pointers (kept alive on self), or None."""
        if x is None:
            return None
        self._keep = [np.ascontiguousarray(r, dtype=np.float64) for r in x]
        dp = C.POINTER(C.c_double)
        return (dp * self.A)(*[r.ctypes.data_as(dp) for r in self._keep])

    def reset(self, xi_reset=None):
        lib().orc_world_env_reset(self._pp, self._lp, self.A, self.lane_id0, self._rows(xi_reset))
        return self.obs()

    def step(self, actions, xi=None):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.A, 4)
        lib().orc_world_env_step(self._pp, self._lp, self.A, _dp(a), self._rows(xi))
        return (self.obs(), np.array([l.reward for l in self.Ls]), np.array([bool(l.terminated) for l in self.Ls]),
                np.array([bool(l.truncated) for l in self.Ls]))

    def obs(self):
        return np.stack([np.frombuffer(l.obs, dtype=np.float64, count=self.obs_dim).copy() for l in self.Ls])


DOGFIGHT_DEFAULTS = dict(team_size=2, spawn_min_radius=10.0, spawn_max_radius=50.0, damage_per_hit=0.003, lethal_distance=20.0,
                         lethal_angle=0.07, aggressiveness=0.5, cooperativeness=0.5, sparse_reward=False, dome=800.0,
                         max_duration_seconds=60.0, agent_hz=30, assisted_flight=True)  # ma_fixedwing_dogfight_env.py:42-60


def dogfight_spawn(team_size, min_radius, max_radius, u):
    """_get_start_pos_orn (:176-213) + reset()'s 20 m/s forward velocity (:216-222) from 1 + 6 team_size uniforms."""
    A = 2 * team_size
    u = np.ascontiguousarray(u, dtype=np.float64)
    assert u.shape == (1 + 3 * A,)
    pos, rpy, vel = np.zeros((A, 3)), np.zeros((A, 3)), np.zeros((A, 3))
    lib().orc_dogfight_spawn(team_size, min_radius, max_radius, _dp(u), _dp(pos), _dp(rpy), _dp(vel))
    return pos, rpy, vel


class OracleDogfight:
    """One world of MAFixedwingDogfightEnv on the fp64 oracle: A = 2 team_size Acrowing aircraft (their own orc_params blocks,
    spawn pose / velocity per aircraft) stepped by orc_world_aviary_step, the env on top restated in orc_dogfight_*."""

    def __init__(self, start_pos, start_rpy, start_vel=None, noise_mode=NOISE_OFF, seed=0, lane_id0=0, **kw):
        cfg = dict(DOGFIGHT_DEFAULTS, **kw)
        self.A = 2 * cfg["team_size"]
        start_pos, start_rpy = np.asarray(start_pos, dtype=np.float64), np.asarray(start_rpy, dtype=np.float64)
        assert start_pos.shape == (self.A, 3) and start_rpy.shape == (self.A, 3)
        if start_vel is None:  # reset(): compute_rotation_forward(start_orn)[1] * 20
            cr, cp = np.cos(start_rpy), None
            start_vel = 20.0 * np.stack([np.cos(start_rpy[:, 2]) * np.cos(start_rpy[:, 1]), np.sin(start_rpy[:, 2]) * np.cos(start_rpy[:, 1]),
                                         -np.sin(start_rpy[:, 1])], axis=1)
        self.env_step_ratio = int(120 / cfg["agent_hz"])
        self.max_steps = int(cfg["agent_hz"] * cfg["max_duration_seconds"])
        # Aviary(world_scale=5.0, drone_type="fixedwing", drone_model="acrowing") (ma_fixedwing_base_env.py:193-210)
        self.Ps = [make_params("acrowing", noise_mode=noise_mode, seed=seed, start_pos=start_pos[i], start_rpy=start_rpy[i], start_vel=start_vel[i],
                               flight_mode=0, world_plane_half_xy=15.0 * 5.0, world_plane_half_z=5.0 * 5.0,
                               world_contact_response=1, **cfg.get("param_overrides", {})) for i in range(self.A)]
        self.Ls = [Lane() for _ in range(self.A)]
        self.lane_id0 = lane_id0
        PP, LP = C.POINTER(Params), C.POINTER(Lane)
        self._pp = (PP * self.A)(*[C.pointer(p) for p in self.Ps])
        self._lp = (LP * self.A)(*[C.pointer(l) for l in self.Ls])
        D = self.D = Dogfight()
        D.A, D.team_size = self.A, cfg["team_size"]
        D.damage_per_hit, D.lethal_distance, D.lethal_angle = cfg["damage_per_hit"], cfg["lethal_distance"], cfg["lethal_angle"]
        D.aggressiveness, D.cooperativeness, D.sparse_reward = cfg["aggressiveness"], cfg["cooperativeness"], int(cfg["sparse_reward"])
        D.dome, D.max_steps, D.env_step_ratio = cfg["dome"], self.max_steps, self.env_step_ratio
        self.action_dim = 4 if cfg["assisted_flight"] else 6
        D.action_dim = self.action_dim
        self.obs_dim = 19 + self.action_dim + (self.A - 1) * 14

    _rows = OracleWorld._rows

    def reset(self, xi_reset=None):
        lib().orc_dogfight_reset(self._pp, self._lp, C.byref(self.D), self.lane_id0, self._rows(xi_reset))
        return self.obs()

    def step(self, actions, xi=None):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.A, self.action_dim)
        lib().orc_dogfight_step(self._pp, self._lp, C.byref(self.D), _dp(a), self._rows(xi))
        D = self.D
        return (self.obs(), np.array(D.reward[:self.A]), np.array(D.terminated[:self.A], dtype=bool), np.array(D.truncated[:self.A], dtype=bool))

    def obs(self):
        return np.stack([np.frombuffer(self.D.obs[i], dtype=np.float64, count=self.obs_dim).copy() for i in range(self.A)])

    @property
    def alive(self):
        return np.array(self.D.alive[:self.A], dtype=bool)

    @property
    def health(self):
        return np.array(self.D.health[:self.A])


class OracleBatch:
    """N independent lanes stepped by the fp64 oracle (OpenMP over lanes)."""

    def __init__(self, params: Params, n: int, lane0: int = 0):
        self.P = params
        self.n = n
        self.lane0 = lane0
        self.lanes = (Lane * n)()
        self.obs_dim = lib().orc_obs_dim(C.byref(self.P))

    def reset(self, mask=None, xi_reset=None, u_targets=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        lib().orc_env_reset_batch(C.byref(self.P), self.lanes, self.n, self.lane0, _vp(m), _vp(xi_reset), _vp(u_targets))
        return self.obs()

    def obs(self):
        out = np.zeros((self.n, self.obs_dim))
        for i in range(self.n):
            out[i] = np.frombuffer(self.lanes[i].obs, dtype=np.float64, count=self.obs_dim)
        return out

    def step(self, actions, xi=None, xi_reset=None, u_targets=None, autoreset=0):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        assert a.shape == (self.n, 4)
        obs = np.zeros((self.n, self.obs_dim))
        fin = np.zeros((self.n, self.obs_dim))
        rew = np.zeros(self.n)
        term = np.zeros(self.n, dtype=np.uint8)
        trunc = np.zeros(self.n, dtype=np.uint8)
        lib().orc_env_step_batch(C.byref(self.P), self.lanes, self.n, _vp(a), _vp(xi), _vp(xi_reset), _vp(u_targets),
                                 autoreset, _vp(obs), _vp(rew), _vp(term), _vp(trunc), _vp(fin))
        return obs, rew, term.astype(bool), trunc.astype(bool), fin

    def field(self, name):
        """Gather a lane field into an [n, ...] float64/int array."""
        first = np.ctypeslib.as_array(getattr(self.lanes[0], name)) if hasattr(getattr(self.lanes[0], name), "_length_") else None
        if first is None:
            return np.array([getattr(self.lanes[i], name) for i in range(self.n)])
        return np.stack([np.ctypeslib.as_array(getattr(self.lanes[i], name)).copy() for i in range(self.n)])
