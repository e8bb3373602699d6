"""ctypes binding of libpyflyt_amd.so (include/pyflyt_amd.h). There is no fallback: if the HIP
extension is missing or cannot be loaded this module raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpyflyt_amd.so")

PF_MAX_BOXES, PF_MAX_SURF = 12, 5
QUADX, FIXEDWING, ROCKET = 0, 1, 2
TASK_NONE, TASK_HOVER, TASK_WAYPOINTS, TASK_MA_HOVER = 0, 1, 2, 3
NOISE_OFF, NOISE_INJECT, NOISE_PHILOX = 0, 1, 2
AUTORESET_OFF, AUTORESET_NEXT_STEP, AUTORESET_SAME_STEP = 0, 1, 2
F_TERMINATED, F_TRUNCATED, F_CONTACT, F_INFO_COLLISION, F_INFO_OOB, F_INFO_COMPLETE = 1, 2, 4, 8, 16, 32

f3 = C.c_float * 3
f4 = C.c_float * 4
f6 = C.c_float * 6


class PfPid(C.Structure):
    _fields_ = [("kp", f3), ("ki", f3), ("kd", f3), ("lim", f3)]


class PfBox(C.Structure):
    _fields_ = [("c", f3), ("h", f3), ("kind", C.c_int32), ("yaw", C.c_float)]


class PfSurface(C.Structure):
    _fields_ = [
        ("r", f3), ("lift", f3), ("drag", f3), ("torque", f3),
        ("Cl_alpha_3D", C.c_float), ("inv_Cl_alpha_3D", C.c_float), ("aero_tau_eta", C.c_float),
        ("flap_to_chord", C.c_float), ("inv_pi_aspect", C.c_float), ("exp_term", C.c_float),
        ("alpha_0_base", C.c_float), ("alpha_stall_P_base", C.c_float), ("alpha_stall_N_base", C.c_float),
        ("Cd_0", C.c_float), ("deflection_limit_rad", C.c_float), ("dt_over_tau", C.c_float),
        ("half_rho_area", C.c_float), ("chord", C.c_float),
    ]


class PfRocket(C.Structure):
    _fields_ = [
        ("dry_mass", C.c_float), ("dry_mr", f3), ("dry_S", f6), ("dry_I", f3), ("tank_r", f3),
        ("total_fuel", C.c_float), ("fuel_rate_ratio", C.c_float), ("fuel_inertia", f3),
        ("thrust_min_ratio", C.c_float), ("max_thrust", C.c_float), ("booster_dt_over_tau", C.c_float),
        ("booster_noise", C.c_float), ("reignitable", C.c_int32), ("booster_r", f3),
        ("gimbal_dt_over_tau", C.c_float), ("gimbal_range_rad", C.c_float), ("finlet_map", f3 * 4),
        ("starting_fuel_ratio", C.c_float),
    ]


class PfParams(C.Structure):
    _fields_ = [
        ("vehicle", C.c_int32), ("task", C.c_int32), ("flight_mode", C.c_int32), ("noise_mode", C.c_int32),
        ("autoreset", C.c_int32), ("angle_repr", C.c_int32), ("sparse_reward", C.c_int32),
        ("num_targets", C.c_int32), ("max_steps", C.c_int32), ("env_step_ratio", C.c_int32),
        ("settle_steps", C.c_int32), ("ticks_per_control", C.c_int32), ("use_gyro_term", C.c_int32),
        ("throttle_remap", C.c_int32), ("n_motors", C.c_int32), ("n_surf", C.c_int32), ("n_boxes", C.c_int32),
        ("has_com_offset", C.c_int32), ("seed", C.c_uint64),
        ("dt", C.c_float), ("gravity_z", C.c_float), ("max_coord_vel", C.c_float),
        ("plane_half_xy", C.c_float), ("plane_half_z", C.c_float),
        ("inv_mass", C.c_float), ("com", f3), ("I_own", f6), ("I_pa", f6), ("I_inv", f6),
        ("bound_radius", C.c_float), ("boxes", PfBox * PF_MAX_BOXES),
        ("motor_r", f3 * 4), ("thrust_unit", f3 * 4),
        ("motor_dt_over_tau", f4), ("motor_fmax", f4), ("motor_tmax", f4), ("motor_noise", f4),
        ("motor_map", f4 * 4), ("drag_const", f3), ("drag_coef_pqr", C.c_float),
        ("pid", PfPid * 4), ("zpid", PfPid * 2),
        ("control_period", C.c_float), ("inv_control_period", C.c_float),
        ("surf", PfSurface * PF_MAX_SURF), ("assist_ids", C.c_int32 * 6), ("assist_signs", f6),
        ("start_pos", f3), ("start_quat", f4), ("start_vel", f3),
        ("dome", C.c_float), ("goal_reach_distance", C.c_float), ("min_height", C.c_float),
        ("wp_dist_reward", C.c_float), ("wp_yaw_penalty", C.c_float),
        ("action_low", f4), ("action_high", f4),
        ("rocket", PfRocket),
    ]


class PfBuffers(C.Structure):
    _fields_ = [
        ("state", C.c_void_p), ("actions", C.c_void_p), ("obs", C.c_void_p), ("final_obs", C.c_void_p),
        ("reward", C.c_void_p), ("terminated", C.c_void_p), ("truncated", C.c_void_p),
        ("xi", C.c_void_p), ("xi_reset", C.c_void_p), ("u_targets", C.c_void_p),
        ("setpoints", C.c_void_p), ("out_state", C.c_void_p), ("out_aux", C.c_void_p),
        ("out_contact", C.c_void_p), ("start_pose", C.c_void_p),
        ("wind", C.c_void_p), ("out_link_pos", C.c_void_p), ("ctrl_ratio", C.c_void_p), ("modes", C.c_void_p), ("start_vel", C.c_void_p), ("armed", C.c_void_p),
    ]


EXPORTS = (
    "pf_abi_version", "pf_sizeof_params", "pf_sizeof_buffers", "pf_last_error", "pf_ctx_create", "pf_ctx_destroy", "pf_state_groups", "pf_obs_dim",
    "pf_n_lanes", "pf_env_reset", "pf_env_step", "pf_aviary_reset", "pf_aviary_set_mode", "pf_aviary_step",
    "pf_sample_actions", "pf_aviary_tick", "pf_wind_links",
)

_lib = None


class PyFlytAmdError(RuntimeError):
    pass


def lib():
    """Load the HIP extension; fail loudly when it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PyFlytAmdError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). pyflyt_amd has no CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise PyFlytAmdError(f"{LIB_PATH} does not export {name}")
    L.pf_abi_version.restype = C.c_int
    L.pf_last_error.restype = C.c_char_p
    L.pf_last_error.argtypes = [C.c_void_p]
    L.pf_ctx_create.argtypes = [C.POINTER(PfParams), C.c_int, C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
    L.pf_ctx_destroy.argtypes = [C.c_void_p]
    L.pf_ctx_destroy.restype = None
    for f in ("pf_state_groups", "pf_obs_dim", "pf_n_lanes"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.pf_env_reset.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_void_p, C.c_void_p]
    L.pf_env_step.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_void_p]
    L.pf_aviary_reset.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_void_p]
    L.pf_aviary_set_mode.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_int, C.c_void_p, C.c_void_p]
    L.pf_aviary_step.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_int, C.c_void_p]
    L.pf_sample_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.pf_aviary_tick.argtypes = [C.c_void_p, C.POINTER(PfBuffers), C.c_int, C.c_void_p]
    L.pf_wind_links.argtypes = [C.c_void_p]
    L.pf_sizeof_params.restype = C.c_size_t
    L.pf_sizeof_buffers.restype = C.c_size_t
    if L.pf_sizeof_params() != C.sizeof(PfParams) or L.pf_sizeof_buffers() != C.sizeof(PfBuffers):
        raise PyFlytAmdError("struct layout mismatch between pyflyt_amd/_lib.py and include/pyflyt_amd.h")
    if L.pf_abi_version() != 2:
        raise PyFlytAmdError("ABI version mismatch between pyflyt_amd/_lib.py and libpyflyt_amd.so")
    _lib = L
    return L


def check(rc: int, ctx=None):
    if rc != 0:
        msg = lib().pf_last_error(ctx)
        raise PyFlytAmdError(f"pyflyt_amd call failed (code {rc}): {msg.decode() if msg else '?'}")
