"""Vehicle / env constants -> pf_params.

The numbers below are copied (as numbers, with citations) from the reference's model files --
file:line relative to /root/reference/PyFlyt/ -- and turned, in float64 on the host, into the
derived constants the kernels consume (rounded to fp32 exactly once). Users can override any
entry through the `vehicle_options` / env kwargs, mirroring the reference's `drone_options`.
"""
from __future__ import annotations

import copy
import math
from typing import Any

import numpy as np

from . import _lib as L

# --------------------------------------------------------------------------- model tables
CF2X: dict[str, Any] = {
    # models/vehicles/cf2x/cf2x.urdf:13-14 (base link), :30-36 (collision box), :42,54,66,78 (prop COMs)
    "mass": 0.027,
    "inertia_diag": (1.4e-5, 1.4e-5, 2.17e-5),
    "collision_boxes": [((0.0, 0.0, 0.0), (0.09, 0.09, 0.02))],
    "motor_r": [(0.028, -0.028, 0.0), (-0.028, 0.028, 0.0), (0.028, 0.028, 0.0), (-0.028, -0.028, 0.0)],
    # models/vehicles/cf2x/cf2x.yaml:1-6
    "total_thrust": 2.0, "thrust_coef": 3.16e-10, "torque_coef": 7.94e-12, "noise_ratio": 0.02, "motor_tau": 0.01,
    # drones/quadx.py:94-101 (torque signs), :130-137 (motor map)
    "torque_signs": (-1.0, -1.0, 1.0, 1.0),
    "motor_map": [(-1.0, -1.0, -1.0, 1.0), (1.0, 1.0, -1.0, 1.0), (1.0, -1.0, 1.0, 1.0), (-1.0, 1.0, 1.0, 1.0)],
    # cf2x.yaml:8-11
    "drag_coef_xyz": 3.0, "drag_area_xyz": 4.0e-4, "drag_coef_pqr": 1.0e-4,
    # cf2x.yaml:13-54 (kp, ki, kd, lim)
    "pid": {
        "ang_vel": ([4.0e-2, 4.0e-2, 8.0e-2], [5.0e-7, 5.0e-7, 2.7e-4], [1.0e-4, 1.0e-4, 0.0], [1.0, 1.0, 1.0]),
        "ang_pos": ([2.0, 2.0, 2.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [3.0, 3.0, 3.0]),
        "lin_vel": ([0.8, 0.8], [0.3, 0.3], [0.5, 0.5], [0.4, 0.4]),
        "lin_pos": ([1.0, 1.0], [0.0, 0.0], [0.0, 0.0], [2.0, 2.0]),
        "z_pos": ([1.0], [0.0], [0.0], [1.0]),
        "z_vel": ([2.0], [0.5], [0.05], [1.0]),
    },
    "control_hz": 120,  # drones/quadx.py:27
}

# QuadX(drone_model="primitive_drone") (drones/quadx.py:29; examples/core/08_mixed_drones.py, the pole and
# ball-in-cup envs): same airframe code, another model folder.
PRIMITIVE_DRONE: dict[str, Any] = {
    # models/vehicles/primitive_drone/primitive_drone.urdf:27-30 (base link), :21-26 (collision box),
    # :42-47,69-74,96-101,123-128 (prop discs: cylinders r 0.12, length 0.01), :56,83,110,138 (prop joints)
    "mass": 1.0,
    "inertia_diag": (0.01, 0.01, 0.016),
    "collision_boxes": [((0.0, 0.0, 0.0), (0.2, 0.1, 0.05))],
    "collision_cylinders": [((0.16, -0.16, 0.0), 0.12, 0.01), ((-0.16, 0.16, 0.0), 0.12, 0.01),
                            ((0.16, 0.16, 0.0), 0.12, 0.01), ((-0.16, -0.16, 0.0), 0.12, 0.01)],
    "motor_r": [(0.16, -0.16, 0.0), (-0.16, 0.16, 0.0), (0.16, 0.16, 0.0), (-0.16, -0.16, 0.0)],
    # models/vehicles/primitive_drone/primitive_drone.yaml:1-6
    "total_thrust": 40.0, "thrust_coef": 3.0e-7, "torque_coef": 3.0e-7, "noise_ratio": 0.003, "motor_tau": 0.01,
    "torque_signs": (-1.0, -1.0, 1.0, 1.0),
    "motor_map": [(-1.0, -1.0, -1.0, 1.0), (1.0, 1.0, -1.0, 1.0), (1.0, -1.0, 1.0, 1.0), (-1.0, 1.0, 1.0, 1.0)],
    # primitive_drone.yaml:8-11
    "drag_coef_xyz": 2.0, "drag_area_xyz": 0.08, "drag_coef_pqr": 1.0e-4,
    # primitive_drone.yaml:13-54 (kp, ki, kd, lim)
    "pid": {
        "ang_vel": ([1.5e-2, 1.5e-2, 5.0e-3], [1.0e-5, 1.0e-5, 2.0e-6], [1.2e-5, 1.2e-5, 1.2e-6], [1.0, 1.0, 1.0]),
        "ang_pos": ([2.0, 2.0, 2.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [6.0, 6.0, 6.0]),
        "lin_vel": ([0.3, 0.3], [0.03, 0.03], [0.3, 0.3], [1.0, 1.0]),
        "lin_pos": ([1.0, 1.0], [0.0, 0.0], [0.0, 0.0], [5.0, 5.0]),
        "z_pos": ([1.0], [0.0], [0.0], [3.0]),
        "z_vel": ([3.0], [0.8], [0.2], [1.0]),
    },
    "control_hz": 120,
}
QUADX_MODELS = {"cf2x": CF2X, "primitive_drone": PRIMITIVE_DRONE}

_SURF_COMMON = dict(Cl_alpha_2D=6.283, flap_to_chord=0.3, eta=0.65, Cd_0=0.01, tau=0.05)
FIXEDWING: dict[str, Any] = {
    # models/vehicles/fixedwing/fixedwing.urdf: (mass, link origin) -- all link inertias are zero
    "links": [
        (0.3, (0.0, 0.0, 0.0)),      # base_link :16-20
        (0.1, (-1.1, 0.0, 0.0)),     # horizontal_tail :39-43,58
        (0.05, (-1.1, 0.0, 0.15)),   # vertical_tail :65-69,84
        (0.2, (-0.5, 0.95, 0.0)),    # ail_left :91-95,110
        (0.2, (-0.5, -0.95, 0.0)),   # ail_right :117-121,136
        (0.5, (-0.5, 0.0, 0.0)),     # main_wing :143-147,162
        (1.0, (-0.45, 0.0, 0.0)),    # fuselage :169-173,188
    ],
    # collision boxes :44-49,70-75,96-101,122-127,148-153,174-179 (centre, size)
    "collision_boxes": [
        ((-1.1, 0.0, 0.0), (0.3, 0.6, 0.05)), ((-1.1, 0.0, 0.15), (0.3, 0.05, 0.3)),
        ((-0.5, 0.95, 0.0), (0.31, 0.3, 0.06)), ((-0.5, -0.95, 0.0), (0.31, 0.3, 0.06)),
        ((-0.5, 0.0, 0.0), (0.3, 1.8, 0.05)), ((-0.45, 0.0, 0.0), (1.4, 0.2, 0.2)),
    ],
    # models/vehicles/fixedwing/fixedwing.yaml:1-6 ; drones/fixedwing.py:147-168
    "total_thrust": 18.0, "thrust_coef": 3.16e-10, "torque_coef": 7.94e-12, "noise_ratio": 0.02, "motor_tau": 0.01,
    "motor_r": (0.0, 0.0, 0.0), "thrust_unit": (1.0, 0.0, 0.0),
    # surfaces in the order of drones/fixedwing.py:80-138; params fixedwing.yaml:8-71
    "surfaces": [
        dict(name="left_aileron", r=(-0.5, 0.95, 0.0), lift=(0, 0, 1), chord=0.3, span=0.3, alpha_0_base=-2.0,
             alpha_stall_P_base=14.0, alpha_stall_N_base=-9.0, deflection_limit=30.0, **_SURF_COMMON),
        dict(name="right_aileron", r=(-0.5, -0.95, 0.0), lift=(0, 0, 1), chord=0.3, span=0.3, alpha_0_base=-2.0,
             alpha_stall_P_base=14.0, alpha_stall_N_base=-9.0, deflection_limit=30.0, **_SURF_COMMON),
        dict(name="horizontal_tail", r=(-1.1, 0.0, 0.0), lift=(0, 0, 1), chord=0.2, span=0.625, alpha_0_base=0.0,
             alpha_stall_P_base=9.0, alpha_stall_N_base=-9.0, deflection_limit=20.0, **_SURF_COMMON),
        dict(name="vertical_tail", r=(-1.1, 0.0, 0.15), lift=(0, 1, 0), chord=0.2, span=0.312, alpha_0_base=0.0,
             alpha_stall_P_base=9.0, alpha_stall_N_base=-9.0, deflection_limit=20.0, **_SURF_COMMON),
        dict(name="main_wing", r=(-0.5, 0.0, 0.0), lift=(0, 0, 1), chord=0.3, span=1.6, alpha_0_base=-2.0,
             alpha_stall_P_base=14.0, alpha_stall_N_base=-9.0, deflection_limit=0.0, **_SURF_COMMON),
    ],
    "assist_ids": (0, 0, 1, 2, 1, 3),                   # drones/fixedwing.py:143
    "assist_signs": (1.0, -1.0, 1.0, -1.0, -1.0, 1.0),  # drones/fixedwing.py:144
    "starting_velocity": (20.0, 0.0, 0.0),              # drones/fixedwing.py:35
    "control_hz": 120,                                  # drones/fixedwing.py:24
}

ROCKET: dict[str, Any] = {
    # models/vehicles/rocket/rocket.urdf: (mass, link COM, own diagonal inertia) in joint order; index 0 = base.
    # base :37-38 | fuel tank :58-59,69 | booster :78-79,95 | fins :104,121 :130,147 :156,173 :182,199 |
    # legs :208,225 :234,251 :260,277 (massless; they carry collision boxes)
    "links": [
        (91.0, (0.0, 0.0, 0.0), (372.6, 372.6, 1.55)),
        (410.9, (0.0, 0.0, 0.0), (1678.0, 1678.0, 7.01)),   # fuel tank: mass and inertia scale with the fuel left
        (47.0, (0.0, 0.0, -2.0), (192.43, 192.43, 0.81)),
        (0.05, (0.35, 0.0, 2.051), (0.0, 0.0, 0.0)), (0.05, (-0.35, 0.0, 2.051), (0.0, 0.0, 0.0)),
        (0.05, (0.0, 0.35, 2.051), (0.0, 0.0, 0.0)), (0.05, (0.0, -0.35, 2.051), (0.0, 0.0, 0.0)),
    ],
    "fueltank_link": 1, "booster_link": 2,
    # collision: base cylinder :43 (the fuel tank's :64 lies inside it), booster cylinder :84, fin boxes
    # :110,136,162,188, leg boxes :214,240,266 on links yawed by :225,251,277
    "collision_cylinders": [((0.0, 0.0, 0.0), 0.185, 4.77), ((0.0, 0.0, -2.0), 0.25, 0.5)],
    "collision_boxes": [((0.35, 0.0, 2.051), (0.3, 0.03, 0.3)), ((-0.35, 0.0, 2.051), (0.3, 0.03, 0.3)),
                        ((0.0, 0.35, 2.051), (0.03, 0.3, 0.3)), ((0.0, -0.35, 2.051), (0.03, 0.3, 0.3))],
    "collision_boxes_yawed": [((0.0, 0.35, -2.4), (0.05, 0.5, 0.05), 0.0), ((0.3031, -0.175, -2.4), (0.05, 0.5, 0.05), 4.188),
                              ((-0.3031, -0.175, -2.4), (0.05, 0.5, 0.05), -4.188)],
    # models/vehicles/rocket/rocket.yaml:7-19
    "total_fuel": 410.9, "max_fuel_rate": 1.451, "fuel_inertia": (1678.0, 1678.0, 7.01), "min_thrust": 2966.7,
    "max_thrust": 7607.0, "reignitable": True, "gimbal_range_degrees": 5.0, "booster_tau": 0.01, "gimbal_tau": 0.01,
    "noise_ratio": 0.01,
    # rocket.yaml:21-32; drones/rocket.py:114-142: "x fins" lift along y, "y fins" along x, all facing -z. The reference
    # binds the four LiftingSurface objects to link ids 0..3 = fuel tank, booster, fin_pos_x, fin_neg_x (surface_id=finlet_id),
    # not to the four fin links: that is where they sample velocity and apply force, and what is reproduced here.
    "finlet": dict(Cl_alpha_2D=6.283, chord=0.5, span=0.5, flap_to_chord=1.0, eta=0.65, alpha_0_base=0.0,
                   alpha_stall_P_base=20.0, alpha_stall_N_base=-20.0, Cd_0=0.01, deflection_limit=45.0, tau=0.05),
    "finlet_links": (1, 2, 3, 4),                   # indices into "links" (= link ids 0..3)
    "finlet_lift": ((0, 1, 0), (0, 1, 0), (1, 0, 0), (1, 0, 0)),
    "finlet_forward": (0, 0, -1),
    "finlet_map": ((0.0, 1.0, 1.0), (0.0, 1.0, -1.0), (1.0, 0.0, -1.0), (1.0, 0.0, 1.0)),  # rocket.py:152-159
    # rocket.yaml:34-40
    "drag_coef": (1.16, 1.16, 2.0), "drag_area": (1.7649, 1.7649, 0.1075),
    "starting_fuel_ratio": 0.05,                    # rocket.py:47
    "control_hz": 120,                              # rocket.py:36
}

# Fixedwing(drone_model="acrowing") (pz_envs/fixedwing_envs/ma_fixedwing_base_env.py:193-195): the aerobatic
# airframe; same class, another model folder.
ACROWING: dict[str, Any] = copy.deepcopy(FIXEDWING)
ACROWING.update({
    # models/vehicles/acrowing/acrowing.urdf: base :21, h-tail :44,61, v-tail :70,87, ailerons :96,113 :122,139,
    # main wing :148,165, fuselage :174,191 (motor and gunsight links are massless)
    "links": [
        (0.3, (0.0, 0.0, 0.0)), (0.1, (-1.1, 0.0, 0.0)), (0.05, (-1.1, 0.0, 0.25)), (0.2, (-0.35, 0.95, 0.0)),
        (0.2, (-0.35, -0.95, 0.0)), (0.5, (-0.35, 0.0, 0.0)), (1.0, (-0.45, 0.0, 0.0)),
    ],
    # collision boxes :50,76,102,128,154,180
    "collision_boxes": [
        ((-1.1, 0.0, 0.0), (0.3, 0.8, 0.05)), ((-1.1, 0.0, 0.25), (0.4, 0.05, 0.5)),
        ((-0.35, 0.95, 0.0), (0.31, 0.3, 0.06)), ((-0.35, -0.95, 0.0), (0.31, 0.3, 0.06)),
        ((-0.35, 0.0, 0.0), (0.3, 1.8, 0.05)), ((-0.45, 0.0, 0.0), (1.4, 0.2, 0.2)),
    ],
    "total_thrust": 30.0,  # acrowing.yaml:2
    # acrowing.yaml:8-71 in the surface order of drones/fixedwing.py:80-138
    "surfaces": [
        dict(_SURF_COMMON, name="left_aileron", r=(-0.35, 0.95, 0.0), lift=(0, 0, 1), chord=0.3, span=0.3, flap_to_chord=0.3,
             alpha_0_base=0.0, alpha_stall_P_base=16.0, alpha_stall_N_base=-12.0, deflection_limit=30.0),
        dict(_SURF_COMMON, name="right_aileron", r=(-0.35, -0.95, 0.0), lift=(0, 0, 1), chord=0.3, span=0.3, flap_to_chord=0.3,
             alpha_0_base=0.0, alpha_stall_P_base=16.0, alpha_stall_N_base=-12.0, deflection_limit=30.0),
        dict(_SURF_COMMON, name="horizontal_tail", r=(-1.1, 0.0, 0.0), lift=(0, 0, 1), chord=0.3, span=0.8, flap_to_chord=0.5,
             alpha_0_base=0.0, alpha_stall_P_base=11.0, alpha_stall_N_base=-11.0, deflection_limit=20.0),
        dict(_SURF_COMMON, name="vertical_tail", r=(-1.1, 0.0, 0.25), lift=(0, 1, 0), chord=0.4, span=0.4, flap_to_chord=0.4,
             alpha_0_base=0.0, alpha_stall_P_base=11.0, alpha_stall_N_base=-11.0, deflection_limit=35.0),
        dict(_SURF_COMMON, name="main_wing", r=(-0.35, 0.0, 0.0), lift=(0, 0, 1), chord=0.3, span=1.6, flap_to_chord=0.1,
             alpha_0_base=-2.0, alpha_stall_P_base=16.0, alpha_stall_N_base=-10.0, deflection_limit=15.0),
    ],
})
FIXEDWING_MODELS = {"fixedwing": FIXEDWING, "acrowing": ACROWING}

# MAFixedwingDogfightEnv's constructor defaults (pz_envs/fixedwing_envs/ma_fixedwing_dogfight_env.py:42-60); sample_spawn: draw
# the spawn circle at every reset (the reference's behaviour) or fly from the poses stored in the state's spawn groups
DOGFIGHT = {
    "team_size": 2, "spawn_min_radius": 10.0, "spawn_max_radius": 50.0, "damage_per_hit": 0.003, "lethal_distance": 20.0,
    "lethal_angle": 0.07, "aggressiveness": 0.5, "cooperativeness": 0.5, "sample_spawn": True,
    "assisted_flight": True,  # False: six-wide actions (see pf_params.df_action_dim for what the reference does with them)
    "freeze_wrecks": False,  # True: an aircraft stops where it hits the ground (no tumbling, no contact solve) -- not the reference's behaviour
}

WORLD: dict[str, Any] = {
    "physics_hz": 240,         # core/aviary.py:79
    "gravity_z": -9.81,        # core/aviary.py:226
    "world_scale": 1.0,        # core/aviary.py:80
    # [BULLET-FROM-MEMORY]: named so they can be corrected
    "use_gyro_term": True,     # btMultiBody::m_useGyroTerm
    "max_coord_vel": 100.0,    # btMultiBody::m_maxCoordinateVelocity
    "plane_half_xy": 15.0,     # pybullet_data plane.urdf collision box 30 x 30 x 10, centre z = -5
    "plane_half_z": 5.0,
    # contact response and contact report: a named-parameter model of what stepSimulation / getContactPoints do after collision
    # detection (vertex contacts, projected Gauss-Seidel at the velocity level, position-level penetration recovery; the model:
    # include/pyflyt_amd.h at pf_params.contact_response, the argument for each default: DESIGN.md section 3). Every entry is
    # [BULLET-FROM-MEMORY]; tests/golden/capture_pybullet.py prints the real getPhysicsEngineParameters() where PyBullet exists.
    # ON everywhere, as in the reference (stepSimulation, aviary.py:516, always solves its contacts). False is an explicit
    # opt-out: contact DETECTION only -- bodies pass through the floor; for the gym env tasks, which end the episode in the
    # Aviary step that reports the contact, it alters that terminal observation only.
    "contact_response": True,
    "contact_restitution": 0.0,          # Bullet's default restitution
    "contact_friction": 0.5,             # lateral friction 0.5 (body) x 1.0 (plane.urdf)
    "contact_erp": 0.2,                  # btContactSolverInfo::m_erp
    "contact_iters": 50,                 # PyBullet's numSolverIterations (Bullet's own default: 10)
    "contact_residual_threshold": 1e-7,  # PyBullet's solverResidualThreshold: sweeps end at a squared row-velocity change <= this
    "contact_margin": 0.0,               # fresh contact points exist from touching on (dBoxBox2: nothing while an axis separates)
    "contact_report_distance": 0.0,      # ... and a fresh pair is reported from touching on
    "contact_break_distance": 0.02,      # persisting points / reports: btPersistentManifold's contact breaking threshold
    "contact_manifold_points": 4,        # per collider box: the incident face's four vertices (8: every vertex)
    "contact_slop": 1e-5,                # PyBullet's m_linearSlop: what a resting body overlaps the floor by
}


def quat_from_euler(rpy):
    """pybullet getQuaternionFromEuler (x, y, z, w)."""
    phi, the, psi = (0.5 * float(a) for a in rpy)
    q = np.array([
        math.sin(phi) * math.cos(the) * math.cos(psi) - math.cos(phi) * math.sin(the) * math.sin(psi),
        math.cos(phi) * math.sin(the) * math.cos(psi) + math.sin(phi) * math.cos(the) * math.sin(psi),
        math.cos(phi) * math.cos(the) * math.sin(psi) - math.sin(phi) * math.sin(the) * math.cos(psi),
        math.cos(phi) * math.cos(the) * math.cos(psi) + math.sin(phi) * math.sin(the) * math.sin(psi),
    ])
    return q / np.linalg.norm(q)


def _sym6(M):
    return [M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2]]


def _fill(arr, vals):
    for i, v in enumerate(vals):
        arr[i] = v


def _set_body(P, links, own_inertia, boxes, cylinders=(), yawed_boxes=(), cylinders_first=False):
    """links: [(mass, r)], own_inertia: 3x3 sum of link inertias (base frame, about their own COMs).
    boxes: [(centre, size)]; cylinders: [(centre, radius, length)], axis = link z; yawed_boxes: [(centre, size, yaw)].
    The shapes are stored in the URDF's link order (the contact solver sweeps its contact vertices in that order, and a
    Gauss-Seidel sweep is order dependent): boxes then cylinders (primitive_drone.urdf: base box, four prop discs), or
    cylinders first (rocket.urdf: body and booster cylinders, then the fin boxes, then the legs)."""
    m = np.array([l[0] for l in links], dtype=np.float64)
    r = np.array([l[1] for l in links], dtype=np.float64)
    M = m.sum()
    com = (m[:, None] * r).sum(0) / M
    I_pa = np.zeros((3, 3))
    for mi, ri in zip(m, r):
        d = ri - com
        I_pa += mi * ((d @ d) * np.eye(3) - np.outer(d, d))
    I_inv = np.linalg.inv(own_inertia + I_pa)
    P.inv_mass = 1.0 / M
    _fill(P.com, com)
    _fill(P.I_own, _sym6(own_inertia))
    _fill(P.I_pa, _sym6(I_pa))
    _fill(P.I_inv, _sym6(I_inv))
    P.has_com_offset = int(np.abs(com).max() > 0.0)
    if len(boxes) + len(cylinders) + len(yawed_boxes) > L.PF_MAX_BOXES:
        raise ValueError(f"at most {L.PF_MAX_BOXES} collision shapes per vehicle")
    P.n_boxes = len(boxes) + len(cylinders) + len(yawed_boxes)
    rad = 0.0
    sb = [(c, 0.5 * np.array(size, dtype=np.float64), 0, 0.0) for c, size in boxes]
    sc = [(c, np.array([r, r, 0.5 * length], dtype=np.float64), 1, 0.0) for c, r, length in cylinders]
    shapes = (sc + sb if cylinders_first else sb + sc) + \
             [(c, 0.5 * np.array(size, dtype=np.float64), 0, float(yaw)) for c, size, yaw in yawed_boxes]
    for k, (c, h, kind, yaw) in enumerate(shapes):
        _fill(P.boxes[k].c, c)
        _fill(P.boxes[k].h, h)
        P.boxes[k].kind = kind
        P.boxes[k].yaw = yaw
        # (a yawed box's own bounding sphere: its half-diagonal, whatever the yaw)
        rad = max(rad, float(np.linalg.norm(c) + np.linalg.norm(h)) if yaw else float(np.linalg.norm(np.abs(np.array(c)) + h)))
    # nudged up so that fp32 rounding can never make the early-out stricter than the exact test
    P.bound_radius = rad * (1.0 + 1e-6)


def _set_pid(dst, kp, ki, kd, lim):
    _fill(dst.kp, kp); _fill(dst.ki, ki); _fill(dst.kd, kd); _fill(dst.lim, lim)


def build_params(
    vehicle: str,
    task: str = "none",
    *,
    flight_mode: int = 0,
    noise: str = "philox",
    autoreset: str = "next_step",
    seed: int = 0,
    angle_representation: str = "quaternion",
    sparse_reward: bool = False,
    flight_dome_size: float | None = None,
    max_duration_seconds: float | None = None,
    agent_hz: int | None = None,
    num_targets: int = 4,
    goal_reach_distance: float | None = None,
    use_yaw_targets: bool = False,
    goal_reach_angle: float = 0.1,
    agents_per_world: int = 0,
    dogfight: dict | None = None,
    start_pos=None,
    start_orn=None,
    vehicle_options: dict | None = None,
    world_options: dict | None = None,
) -> L.PfParams:
    """vehicle in {'quadx','fixedwing','rocket'}; task in {'none','hover','waypoints','ma_hover','dogfight'}.
    dogfight: overrides of DOGFIGHT (team_size, spawn radii, damage_per_hit, lethal_distance, lethal_angle, aggressiveness,
    cooperativeness, sample_spawn)."""
    W = dict(WORLD, **(world_options or {}))
    P = L.PfParams()
    P.noise_mode = {"off": L.NOISE_OFF, "inject": L.NOISE_INJECT, "philox": L.NOISE_PHILOX}[noise]
    P.autoreset = {"off": L.AUTORESET_OFF, "disabled": L.AUTORESET_OFF, "next_step": L.AUTORESET_NEXT_STEP,
                   "same_step": L.AUTORESET_SAME_STEP}[autoreset]
    P.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    if angle_representation not in ("euler", "quaternion"):
        raise ValueError(f"angle_representation must be either `euler` or `quaternion`, not {angle_representation}")
    P.angle_repr = 1 if angle_representation == "quaternion" else 0
    P.sparse_reward = int(bool(sparse_reward))
    P.flight_mode = int(flight_mode)
    dt = 1.0 / W["physics_hz"]
    P.dt = dt
    P.gravity_z = W["gravity_z"]
    P.max_coord_vel = W["max_coord_vel"]
    P.use_gyro_term = int(bool(W["use_gyro_term"]))
    P.plane_half_xy = W["plane_half_xy"] * W["world_scale"]
    P.plane_half_z = W["plane_half_z"] * W["world_scale"]
    # contact response against the ground slab (named-parameter model; Bullet's defaults)
    P.contact_response = int(bool(W["contact_response"]))
    P.contact_restitution, P.contact_friction, P.contact_erp = W["contact_restitution"], W["contact_friction"], W["contact_erp"]
    P.contact_iters = int(W["contact_iters"])
    P.contact_margin = W["contact_margin"]  # (lengths of the contact model itself: not scaled with the world, as in
    P.contact_slop = W["contact_slop"]      #  oracle/fake_bullet.py -- globalScaling scales the plane's geometry only)
    P.contact_report_distance = W["contact_report_distance"]
    P.contact_break_distance = W["contact_break_distance"]
    P.contact_residual_threshold = W["contact_residual_threshold"]
    P.contact_manifold_points = int(W["contact_manifold_points"])
    if P.contact_manifold_points not in (4, 8):
        raise ValueError("contact_manifold_points: 4 (the incident face of a box) or 8 (every vertex)")
    if min(P.contact_margin, P.contact_report_distance, P.contact_break_distance, P.contact_slop, P.contact_residual_threshold) < 0.0:
        raise ValueError("the contact model's distances and its residual threshold are non-negative")
    P.settle_steps = 10  # gym_envs/quadx_envs/quadx_base_env.py:209

    if vehicle == "quadx":
        vo = dict(vehicle_options or {})
        model = vo.pop("drone_model", "cf2x")  # drones/quadx.py:29
        if model not in QUADX_MODELS:
            raise ValueError(f"unknown QuadX drone_model {model!r}; available: {sorted(QUADX_MODELS)}")
        V = copy.deepcopy(QUADX_MODELS[model])
        V.update(vo)
        P.vehicle = L.QUADX
        _set_body(P, [(V["mass"], (0.0, 0.0, 0.0))], np.diag(V["inertia_diag"]).astype(np.float64), V["collision_boxes"],
                  V.get("collision_cylinders", ()))
        P.n_motors = 4
        max_rpm = math.sqrt(V["total_thrust"] / (4.0 * V["thrust_coef"]))  # drones/quadx.py:111-113
        for i in range(4):
            _fill(P.motor_r[i], V["motor_r"][i])
            _fill(P.thrust_unit[i], (0.0, 0.0, 1.0))
            P.motor_dt_over_tau[i] = dt / V["motor_tau"]
            P.motor_fmax[i] = V["thrust_coef"] * max_rpm**2
            P.motor_tmax[i] = V["torque_signs"][i] * V["torque_coef"] * max_rpm**2
            P.motor_noise[i] = V["noise_ratio"]
            _fill(P.motor_map[i], V["motor_map"][i])
        _fill(P.drag_const, [0.5 * 1.225 * V["drag_coef_xyz"] * V["drag_area_xyz"]] * 3)  # boring_bodies.py:63
        P.drag_coef_pqr = V["drag_coef_pqr"]
        for k, name in enumerate(("ang_vel", "ang_pos", "lin_vel", "lin_pos")):
            _set_pid(P.pid[k], *V["pid"][name])
        _set_pid(P.zpid[0], *V["pid"]["z_vel"])
        _set_pid(P.zpid[1], *V["pid"]["z_pos"])
        default_start, start_vel = (0.0, 0.0, 1.0), (0.0, 0.0, 0.0)  # quadx_base_env.py:23
        xyz, thr = math.pi, 0.8  # quadx_base_env.py:80-102
        if flight_mode == -1:
            low, high = (0.0, 0.0, 0.0, 0.0), (thr, thr, thr, thr)
        else:
            low, high = (-xyz, -xyz, -xyz, 0.0), (xyz, xyz, xyz, thr)
    elif vehicle == "fixedwing":
        vo = dict(vehicle_options or {})
        model = vo.pop("drone_model", "fixedwing")  # drones/fixedwing.py:25
        if model not in FIXEDWING_MODELS:
            raise ValueError(f"unknown Fixedwing drone_model {model!r}; available: {sorted(FIXEDWING_MODELS)}")
        V = copy.deepcopy(FIXEDWING_MODELS[model])
        V.update(vo)
        P.vehicle = L.FIXEDWING
        _set_body(P, V["links"], np.zeros((3, 3)), V["collision_boxes"])
        P.n_motors = 1
        max_rpm = math.sqrt(V["total_thrust"] / V["thrust_coef"])  # drones/fixedwing.py:152-154
        _fill(P.motor_r[0], V["motor_r"])
        _fill(P.thrust_unit[0], V["thrust_unit"])
        P.motor_dt_over_tau[0] = dt / V["motor_tau"]
        P.motor_fmax[0] = V["thrust_coef"] * max_rpm**2
        P.motor_tmax[0] = V["torque_coef"] * max_rpm**2
        P.motor_noise[0] = V["noise_ratio"]
        P.n_surf = len(V["surfaces"])
        assert P.n_surf == L.PF_MAX_SURF
        for i, s in enumerate(V["surfaces"]):  # abstractions/lifting_surfaces.py:228-239
            S = P.surf[i]
            lift = np.array(s["lift"], dtype=np.float64)
            drag = np.array((1.0, 0.0, 0.0))
            aspect = s["span"] / s["chord"]
            Cl3D = s["Cl_alpha_2D"] * (aspect / (aspect + ((2.0 * (aspect + 4.0)) / (aspect + 2.0))))
            theta_f = math.acos(2.0 * s["flap_to_chord"] - 1.0)
            aero_tau = 1.0 - ((theta_f - math.sin(theta_f)) / math.pi)
            _fill(S.r, s["r"]); _fill(S.lift, lift); _fill(S.drag, drag); _fill(S.torque, np.cross(lift, drag))
            S.Cl_alpha_3D = Cl3D
            S.inv_Cl_alpha_3D = 1.0 / Cl3D
            S.aero_tau_eta = aero_tau * s["eta"]
            S.flap_to_chord = s["flap_to_chord"]
            S.inv_pi_aspect = 1.0 / (math.pi * aspect)
            S.exp_term = 0.41 * (1.0 - math.exp(-17.0 / aspect))
            S.alpha_0_base = math.radians(s["alpha_0_base"])
            S.alpha_stall_P_base = math.radians(s["alpha_stall_P_base"])
            S.alpha_stall_N_base = math.radians(s["alpha_stall_N_base"])
            S.Cd_0 = s["Cd_0"]
            S.deflection_limit_rad = math.radians(s["deflection_limit"])
            S.dt_over_tau = dt / s["tau"]
            S.half_rho_area = 0.5 * 1.225 * s["chord"] * s["span"]
            S.chord = s["chord"]
        _fill(P.assist_ids, V["assist_ids"])
        _fill(P.assist_signs, V["assist_signs"])
        default_start, start_vel = (0.0, 0.0, 10.0), V["starting_velocity"]  # fixedwing_waypoints_env.py:63
        low, high = (-1.0,) * 4, (1.0,) * 4  # fixedwing_base_env.py:78-80
    elif vehicle == "rocket":
        if task != "none":
            raise ValueError("the Rocket is available at the Aviary level only (Rocket-Landing needs a resting contact)")
        V = copy.deepcopy(ROCKET)
        V.update(vehicle_options or {})
        P.vehicle = L.ROCKET
        links = V["links"]
        own = np.diag(np.sum([l[2] for l in links], axis=0)).astype(np.float64)
        _set_body(P, [(l[0], l[1]) for l in links], own, V["collision_boxes"], V["collision_cylinders"], V["collision_boxes_yawed"],
                  cylinders_first=True)
        P.n_motors = 1  # np_random.normal(*throttle.shape) with one booster: xi ~ N(1, 1) (boosters.py:236-240)
        K = P.rocket
        ft, bo = V["fueltank_link"], V["booster_link"]
        dry = [l for i, l in enumerate(links) if i != ft]
        m = np.array([l[0] for l in dry], dtype=np.float64); r = np.array([l[1] for l in dry], dtype=np.float64)
        K.dry_mass = m.sum()
        _fill(K.dry_mr, (m[:, None] * r).sum(0))
        S = sum(mi * ((ri @ ri) * np.eye(3) - np.outer(ri, ri)) for mi, ri in zip(m, r))
        _fill(K.dry_S, _sym6(S))
        _fill(K.dry_I, np.sum([l[2] for l in dry], axis=0))
        _fill(K.tank_r, links[ft][1]); _fill(K.booster_r, links[bo][1])
        K.total_fuel = V["total_fuel"]; K.fuel_rate_ratio = V["max_fuel_rate"] / V["total_fuel"]
        _fill(K.fuel_inertia, V["fuel_inertia"])
        K.thrust_min_ratio = V["min_thrust"] / V["max_thrust"]; K.max_thrust = V["max_thrust"]
        K.booster_dt_over_tau = dt / V["booster_tau"]; K.booster_noise = V["noise_ratio"]
        K.reignitable = int(bool(V["reignitable"]))
        K.gimbal_dt_over_tau = dt / V["gimbal_tau"]; K.gimbal_range_rad = math.radians(V["gimbal_range_degrees"])
        for i in range(4):
            _fill(K.finlet_map[i], V["finlet_map"][i])
        K.starting_fuel_ratio = float(V["starting_fuel_ratio"])
        P.n_surf = 4
        f = V["finlet"]
        for i in range(4):  # abstractions/lifting_surfaces.py:228-239
            Sf = P.surf[i]
            lift = np.array(V["finlet_lift"][i], dtype=np.float64); fwd = np.array(V["finlet_forward"], dtype=np.float64)
            aspect = f["span"] / f["chord"]
            Cl3D = f["Cl_alpha_2D"] * (aspect / (aspect + ((2.0 * (aspect + 4.0)) / (aspect + 2.0))))
            theta_f = math.acos(2.0 * f["flap_to_chord"] - 1.0)
            aero_tau = 1.0 - ((theta_f - math.sin(theta_f)) / math.pi)
            _fill(Sf.r, links[V["finlet_links"][i]][1]); _fill(Sf.lift, lift); _fill(Sf.drag, fwd); _fill(Sf.torque, np.cross(lift, fwd))
            Sf.Cl_alpha_3D = Cl3D; Sf.inv_Cl_alpha_3D = 1.0 / Cl3D; Sf.aero_tau_eta = aero_tau * f["eta"]
            Sf.flap_to_chord = f["flap_to_chord"]; Sf.inv_pi_aspect = 1.0 / (math.pi * aspect)
            Sf.exp_term = 0.41 * (1.0 - math.exp(-17.0 / aspect))
            Sf.alpha_0_base = math.radians(f["alpha_0_base"]); Sf.alpha_stall_P_base = math.radians(f["alpha_stall_P_base"])
            Sf.alpha_stall_N_base = math.radians(f["alpha_stall_N_base"]); Sf.Cd_0 = f["Cd_0"]
            Sf.deflection_limit_rad = math.radians(f["deflection_limit"]); Sf.dt_over_tau = dt / f["tau"]
            Sf.half_rho_area = 0.5 * 1.225 * f["chord"] * f["span"]; Sf.chord = f["chord"]
        _fill(P.drag_const, [0.5 * 1.225 * c * a for c, a in zip(V["drag_coef"], V["drag_area"])])  # boring_bodies.py:63
        default_start, start_vel = (0.0, 0.0, 1.0), (0.0, 0.0, 0.0)
        low, high = (-1.0,) * 4, (1.0,) * 4
    else:
        raise ValueError(f"unknown vehicle {vehicle!r}")

    control_hz = V["control_hz"]
    if W["physics_hz"] % control_hz != 0:  # base_drone.py:95-98
        raise ValueError(f"`physics_hz` ({W['physics_hz']}) must be multiple of `control_hz` ({control_hz}).")
    P.ticks_per_control = W["physics_hz"] // control_hz
    P.control_period = 1.0 / control_hz
    P.inv_control_period = float(control_hz)
    _fill(P.action_low, low)
    _fill(P.action_high, high)

    # ---- task constants (defaults of the reference's env constructors)
    if task == "none":
        P.task = L.TASK_NONE
        d_dome, d_dur, d_hz, d_reach = math.inf, 10.0, 30, 0.2
        P.env_step_ratio = 1
    elif task == "hover":  # quadx_hover_env.py:32-37
        P.task = L.TASK_HOVER
        d_dome, d_dur, d_hz, d_reach = 3.0, 10.0, 40, 0.2
    elif task == "ma_hover":  # pz_envs/quadx_envs/ma_quadx_hover_env.py:36-52
        P.task = L.TASK_MA_HOVER
        P.agents_per_world = int(agents_per_world)  # > 1: the agents of an env share one world (ma_quadx_base_env.py:206-241)
        d_dome, d_dur, d_hz, d_reach = 10.0, 30.0, 40, 0.2
    elif task == "dogfight":  # pz_envs/fixedwing_envs/ma_fixedwing_dogfight_env.py:42-60
        if vehicle != "fixedwing":
            raise ValueError("the dogfight task flies fixedwing aircraft (drone_model 'acrowing' in the reference)")
        DF = dict(DOGFIGHT, **(dogfight or {}))
        P.task = L.TASK_DOGFIGHT
        P.df_team_size = int(DF["team_size"])
        P.agents_per_world = 2 * P.df_team_size
        P.df_sample_spawn = int(bool(DF["sample_spawn"]))
        P.df_action_dim = 4 if DF["assisted_flight"] else 6
        P.df_freeze_wrecks = int(bool(DF["freeze_wrecks"]))
        if P.df_freeze_wrecks and "contact_response" not in (world_options or {}):
            P.contact_response = 0  # the aircraft stops at the first contact REPORT: there is nothing left for the contact solve to do
        P.df_spawn_min_radius, P.df_spawn_max_radius = float(DF["spawn_min_radius"]), float(DF["spawn_max_radius"])
        P.df_damage_per_hit, P.df_lethal_distance, P.df_lethal_angle = float(DF["damage_per_hit"]), float(DF["lethal_distance"]), float(DF["lethal_angle"])
        P.df_aggressiveness, P.df_cooperativeness = float(DF["aggressiveness"]), float(DF["cooperativeness"])
        P.throttle_remap = 1  # ma_fixedwing_base_env.py:300-301
        d_dome, d_dur, d_hz, d_reach = 800.0, 60.0, 30, 0.0
    elif task == "waypoints":
        P.task = L.TASK_WAYPOINTS
        if vehicle == "quadx":  # quadx_waypoints_env.py:38-47,87
            d_dome, d_dur, d_hz, d_reach = 5.0, 10.0, 30, 0.2
            P.min_height, P.wp_dist_reward, P.wp_yaw_penalty = 0.1, 0.1, 0.01
        else:  # fixedwing_waypoints_env.py:36-45,81
            d_dome, d_dur, d_hz, d_reach = 100.0, 120.0, 30, 2.0
            P.min_height, P.wp_dist_reward, P.wp_yaw_penalty = 0.5, 1.0, 0.0
            P.throttle_remap = 1
        P.num_targets = int(num_targets)
        if use_yaw_targets and vehicle != "quadx":
            raise ValueError("use_yaw_targets exists for QuadX-Waypoints only (fixedwing_waypoints_env.py:77)")
        P.use_yaw_targets = int(bool(use_yaw_targets))  # quadx_waypoints_env.py:40
        P.goal_reach_angle = min(float(goal_reach_angle), 3.0e38)  # :42
    else:
        raise ValueError(f"unknown task {task!r}")
    hz = d_hz if agent_hz is None else int(agent_hz)
    if 120 % hz != 0:  # quadx_base_env.py:47-52
        lowest = int(120 / (int(120 / hz) + 1))
        highest = int(120 / int(120 / hz))
        raise ValueError(f"`agent_hz` must be round denominator of 120, try {lowest} or {highest}.")
    dur = d_dur if max_duration_seconds is None else float(max_duration_seconds)
    P.max_steps = int(hz * dur)
    if task != "none":
        P.env_step_ratio = int(120 / hz)
    dome = d_dome if flight_dome_size is None else float(flight_dome_size)
    P.dome = min(dome, 3.0e38)
    P.goal_reach_distance = d_reach if goal_reach_distance is None else float(goal_reach_distance)
    _fill(P.start_pos, default_start if start_pos is None else start_pos)
    _fill(P.start_quat, quat_from_euler((0.0, 0.0, 0.0) if start_orn is None else start_orn))
    _fill(P.start_vel, start_vel)
    return P
