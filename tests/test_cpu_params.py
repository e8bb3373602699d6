"""CPU cross-check of the two independent transcriptions of the reference's model files: the device
parameter tables (pyflyt_amd/params.py -> pf_params, float32) against the oracle's (oracle/uav_oracle.c
-> orc_params, float64). The GPU parity tests imply this; here it runs without a GPU, every round."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from pyflyt_amd import build_params

CASES = [
    ("quadx", {}, "quadx"),
    ("quadx", dict(drone_model="primitive_drone"), "primitive_drone"),
    ("fixedwing", {}, "fixedwing"),
    ("fixedwing", dict(drone_model="acrowing"), "acrowing"),
    ("rocket", {}, "rocket"),
]


def sym6(M):
    M = np.array([list(r) for r in M])
    return np.array([M[0, 0], M[0, 1], M[0, 2], M[1, 1], M[1, 2], M[2, 2]])


@pytest.mark.parametrize("vehicle,opts,oname", CASES)
def test_device_params_match_oracle_params(vehicle, opts, oname):
    P = build_params(vehicle, "none", vehicle_options=opts)
    Q = O.make_params(oname)
    rt = dict(rtol=2e-6, atol=1e-9)
    assert P.dt == pytest.approx(Q.world.dt, rel=1e-6) and P.ticks_per_control == Q.world.ticks_per_control
    assert P.gravity_z == pytest.approx(Q.world.gravity_z, rel=1e-6) and P.max_coord_vel == pytest.approx(Q.world.max_coord_vel)
    assert bool(P.use_gyro_term) == bool(Q.world.use_gyro_term)
    # the contact model's named parameters: the same defaults on both sides
    for f in ("contact_restitution", "contact_friction", "contact_erp", "contact_margin", "contact_slop", "contact_report_distance",
              "contact_break_distance", "contact_residual_threshold"):
        assert float(getattr(P, f)) == pytest.approx(float(getattr(Q.world, f)), rel=1e-6, abs=1e-12), f
    assert (P.contact_response, P.contact_iters, P.contact_manifold_points) == (Q.world.contact_response, Q.world.contact_iters, Q.world.contact_manifold_points)
    assert 1.0 / P.inv_mass == pytest.approx(Q.mass, rel=2e-6)
    np.testing.assert_allclose(list(P.com), list(Q.com), **rt)
    np.testing.assert_allclose(list(P.I_own), sym6(Q.I_own), **rt)
    np.testing.assert_allclose(list(P.I_pa), sym6(Q.I_pa), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(list(P.I_inv), sym6(Q.I_inv), rtol=5e-6, atol=1e-7)
    assert P.n_boxes == Q.n_boxes
    key = lambda b: (b.kind, round(b.c[0], 4), round(b.c[1], 4), round(b.c[2], 4))  # noqa: E731 -- the verdict is an OR: order-free
    for a, b in zip(sorted(list(P.boxes)[: P.n_boxes], key=key), sorted(list(Q.boxes)[: Q.n_boxes], key=key)):
        np.testing.assert_allclose(list(a.c), list(b.c), **rt)
        np.testing.assert_allclose(list(a.h), list(b.h), **rt)
        assert a.kind == b.kind and a.yaw == pytest.approx(b.yaw, rel=1e-6)
    assert P.bound_radius >= 0.999 * max(np.linalg.norm(np.abs(list(Q.boxes[k].c)) + np.array(list(Q.boxes[k].h))) * (Q.boxes[k].yaw == 0)
                                         for k in range(Q.n_boxes))
    assert P.n_motors == Q.n_motors and P.n_surf == Q.n_surf
    assert P.control_period == pytest.approx(Q.control_period, rel=1e-6)
    np.testing.assert_allclose(list(P.drag_const), list(Q.drag_const), **rt)
    if vehicle != "rocket":
        for i in range(P.n_motors):
            np.testing.assert_allclose(list(P.motor_r[i]), list(Q.motor_r[i]), **rt)
            np.testing.assert_allclose(list(P.thrust_unit[i]), list(Q.thrust_unit[i]), **rt)
            assert P.motor_dt_over_tau[i] == pytest.approx(Q.world.dt / Q.motor_tau[i], rel=2e-6)
            assert P.motor_fmax[i] == pytest.approx(Q.thrust_coef[i] * Q.max_rpm[i] ** 2, rel=2e-6)
            assert P.motor_tmax[i] == pytest.approx(Q.torque_coef[i] * Q.max_rpm[i] ** 2, rel=2e-6)
            assert P.motor_noise[i] == pytest.approx(Q.noise_ratio[i], rel=2e-6)
    if vehicle == "quadx":
        assert P.drag_coef_pqr == pytest.approx(Q.drag_coef_pqr, rel=1e-6)
        np.testing.assert_allclose(np.array([list(r) for r in P.motor_map]), np.array([list(r) for r in Q.motor_map]))
        for k in range(4):
            for f in ("kp", "ki", "kd", "lim"):
                np.testing.assert_allclose(list(getattr(P.pid[k], f)), list(getattr(Q.pid[k], f)), **rt)
        for k in range(2):
            for f in ("kp", "ki", "kd", "lim"):
                assert getattr(P.zpid[k], f)[0] == pytest.approx(getattr(Q.zpid[k], f)[0], rel=2e-6)
    for i in range(P.n_surf):  # the host-side precompute of lifting_surfaces.py:228-239 against the oracle's
        S, T = P.surf[i], Q.surf[i]
        np.testing.assert_allclose(list(S.r), list(T.r), **rt)
        np.testing.assert_allclose(list(S.lift), list(T.lift_unit), **rt)
        np.testing.assert_allclose(list(S.drag), list(T.drag_unit), **rt)
        np.testing.assert_allclose(list(S.torque), list(T.torque_unit), atol=1e-7)
        assert S.Cl_alpha_3D == pytest.approx(T.Cl_alpha_3D, rel=2e-6)
        assert S.aero_tau_eta == pytest.approx(T.aero_tau * T.eta, rel=2e-6)
        assert S.flap_to_chord == pytest.approx(T.flap_to_chord, rel=1e-6)
        assert S.inv_pi_aspect == pytest.approx(1.0 / (math.pi * T.aspect), rel=2e-6)
        assert S.exp_term == pytest.approx(0.41 * (1.0 - math.exp(-17.0 / T.aspect)), rel=2e-6)
        for a, b in (("alpha_0_base", "alpha_0_base"), ("alpha_stall_P_base", "alpha_stall_P_base"), ("alpha_stall_N_base", "alpha_stall_N_base"), ("Cd_0", "Cd_0")):
            assert getattr(S, a) == pytest.approx(getattr(T, b), rel=2e-6, abs=1e-9)
        assert S.deflection_limit_rad == pytest.approx(math.radians(T.deflection_limit), rel=2e-6)
        assert S.dt_over_tau == pytest.approx(Q.world.dt / T.tau, rel=2e-6)
        assert S.half_rho_area == pytest.approx(T.half_rho * T.area, rel=2e-6) and S.chord == pytest.approx(T.chord, rel=1e-6)
    if vehicle == "fixedwing":
        assert list(P.assist_ids) == list(Q.assist_ids) and list(P.assist_signs) == list(Q.assist_signs)
        np.testing.assert_allclose(list(P.start_vel), list(Q.start_vel))
    if vehicle == "rocket":
        K = P.rocket
        m = np.array(list(Q.link_mass)[: Q.n_links]); r = np.array([list(x) for x in Q.link_r][: Q.n_links]); I = np.array([list(x) for x in Q.link_I][: Q.n_links])
        dry = [i for i in range(Q.n_links) if i != Q.fueltank_link]
        assert K.dry_mass == pytest.approx(m[dry].sum(), rel=2e-6)
        np.testing.assert_allclose(list(K.dry_mr), (m[dry, None] * r[dry]).sum(0), rtol=2e-6, atol=1e-6)
        S = sum(m[i] * ((r[i] @ r[i]) * np.eye(3) - np.outer(r[i], r[i])) for i in dry)
        np.testing.assert_allclose(list(K.dry_S), [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(list(K.dry_I), I[dry].sum(0), **rt)
        np.testing.assert_allclose(list(K.tank_r), r[Q.fueltank_link], **rt)
        np.testing.assert_allclose(list(K.booster_r), r[Q.booster_link], **rt)
        assert K.total_fuel == pytest.approx(Q.total_fuel, rel=1e-6) and K.fuel_rate_ratio == pytest.approx(Q.max_fuel_rate / Q.total_fuel, rel=2e-6)
        np.testing.assert_allclose(list(K.fuel_inertia), list(Q.fuel_inertia), **rt)
        assert K.thrust_min_ratio == pytest.approx(Q.min_thrust / Q.max_thrust, rel=2e-6) and K.max_thrust == pytest.approx(Q.max_thrust, rel=1e-6)
        assert K.booster_dt_over_tau == pytest.approx(Q.world.dt / Q.booster_tau, rel=2e-6) and K.booster_noise == pytest.approx(Q.booster_noise, rel=1e-6)
        assert K.gimbal_dt_over_tau == pytest.approx(Q.world.dt / Q.gimbal_tau, rel=2e-6) and K.gimbal_range_rad == pytest.approx(Q.gimbal_range_rad, rel=2e-6)
        assert bool(K.reignitable) == bool(Q.reignitable) and K.starting_fuel_ratio == pytest.approx(Q.starting_fuel_ratio, rel=1e-6)
        np.testing.assert_allclose(np.array([list(x) for x in K.finlet_map]), np.array([list(x) for x in Q.finlet_map]))


@pytest.mark.parametrize("task,oname", [("hover", "hover"), ("waypoints", "quadx_waypoints"), ("ma_hover", "ma_hover")])
def test_quadx_task_constants_match(task, oname):
    P = build_params("quadx", task, autoreset="off" if task == "ma_hover" else "next_step")
    Q = O.make_params(oname)
    assert (P.max_steps, P.env_step_ratio, P.settle_steps, P.num_targets if task == "waypoints" else 0) == \
           (Q.max_steps, Q.env_step_ratio, Q.settle_steps, Q.num_targets if task == "waypoints" else 0)
    assert P.dome == pytest.approx(Q.dome)
    if task == "waypoints":  # (fields the other tasks never read)
        assert P.goal_reach_distance == pytest.approx(Q.goal_reach_distance, rel=1e-6) and P.min_height == pytest.approx(Q.min_height, rel=1e-6)
        assert P.wp_dist_reward == pytest.approx(Q.wp_dist_reward, rel=1e-6) and P.wp_yaw_penalty == pytest.approx(Q.wp_yaw_penalty, rel=1e-6)
    if task != "ma_hover":  # (the MA env's spawns are per agent, in the state's side block)
        np.testing.assert_allclose(list(P.start_pos), list(Q.start_pos))


def test_fixedwing_task_constants_match():
    P = build_params("fixedwing", "waypoints")
    Q = O.make_params("fixedwing_waypoints")
    assert (P.max_steps, P.env_step_ratio, P.settle_steps, P.num_targets) == (Q.max_steps, Q.env_step_ratio, Q.settle_steps, Q.num_targets)
    assert P.dome == pytest.approx(Q.dome) and P.goal_reach_distance == pytest.approx(Q.goal_reach_distance, rel=1e-6)
    assert P.min_height == pytest.approx(Q.min_height, rel=1e-6) and bool(P.throttle_remap) == bool(Q.throttle_remap)
    np.testing.assert_allclose(list(P.start_pos), list(Q.start_pos))
    np.testing.assert_allclose(list(P.start_vel), list(Q.start_vel))


def test_dogfight_task_constants():
    """PF_TASK_DOGFIGHT parameter block against MAFixedwingDogfightEnv's constructor defaults (ma_fixedwing_dogfight_env.py:42-60)
    and the base env's Aviary arguments (ma_fixedwing_base_env.py:193-210)."""
    from pyflyt_amd import _lib as L
    from pyflyt_amd.params import build_params

    P = build_params("fixedwing", "dogfight", autoreset="off", angle_representation="euler", vehicle_options=dict(drone_model="acrowing"),
                     world_options=dict(world_scale=5.0))
    assert P.task == L.TASK_DOGFIGHT and P.df_team_size == 2 and P.agents_per_world == 4 and P.df_sample_spawn == 1 and P.df_freeze_wrecks == 0
    assert (P.df_spawn_min_radius, P.df_spawn_max_radius) == (10.0, 50.0)
    assert abs(P.df_damage_per_hit - 0.003) < 1e-9 and P.df_lethal_distance == 20.0 and abs(P.df_lethal_angle - 0.07) < 1e-9
    assert P.df_aggressiveness == 0.5 and P.df_cooperativeness == 0.5
    assert P.dome == 800.0 and P.max_steps == 1800 and P.env_step_ratio == 4 and P.throttle_remap == 1 and P.flight_mode == 0
    assert P.plane_half_xy == 75.0 and P.plane_half_z == 25.0 and P.contact_response == 1  # world_scale 5; wrecks come to rest on the floor
    assert abs(P.motor_fmax[0] - 30.0) < 1e-4  # acrowing.yaml:2
    Pf = build_params("fixedwing", "dogfight", autoreset="off", angle_representation="euler", vehicle_options=dict(drone_model="acrowing"),
                      dogfight=dict(team_size=3, freeze_wrecks=True))
    assert Pf.agents_per_world == 6 and Pf.df_freeze_wrecks == 1 and Pf.contact_response == 0
    import pytest
    with pytest.raises(ValueError):
        build_params("quadx", "dogfight")
