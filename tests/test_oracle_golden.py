"""The fp64 oracle (oracle/uav_oracle.c) against the golden vectors captured from the reference's
own Python (tests/golden/gen_goldens.py). CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O

TOL = 1e-10


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


# ------------------------------------------------------------------ components
@pytest.mark.parametrize("name,n", [("ang_vel", 3), ("lin_vel", 2), ("z_vel", 1)])
def test_pid(golden_dir, name, n):
    g = load(golden_dir, "pid")
    kp, ki, kd, lim = (np.ascontiguousarray(x) for x in g[f"{name}_gains"])
    I, E = np.zeros(3), np.zeros(3)
    for s, sp, ref in zip(g[f"{name}_state"], g[f"{name}_setpoint"], g[f"{name}_out"]):
        out = np.zeros(3)
        s, sp = np.ascontiguousarray(s), np.ascontiguousarray(sp)
        O.lib().orc_pid_step(dp(kp), dp(ki), dp(kd), dp(lim), 1.0 / 120.0, n, dp(I), dp(E), dp(s), dp(sp), dp(out))
        np.testing.assert_allclose(out[:n], ref, rtol=0, atol=1e-15)


def test_aero_coefficients_and_forces(golden_dir):
    g = load(golden_dir, "aero")
    P = O.make_params("fixedwing")
    for s in range(5):
        S = P.surf[s]
        np.testing.assert_allclose(
            [S.area, S.aspect, S.Cl_alpha_3D, S.theta_f, S.aero_tau, S.half_rho], g["consts"][s], rtol=1e-15)
        for a, act in enumerate(g["actuations"]):
            for k, al in enumerate(g["alphas"]):
                out = np.zeros(3)
                O.lib().orc_surface_aero(C.byref(S), float(al), float(act), dp(out))
                np.testing.assert_allclose(out, g["coeffs"][s, a, k], rtol=1e-13, atol=1e-15)
        for k, v in enumerate(g["vels"]):
            F, T = np.zeros(3), np.zeros(3)
            v = np.ascontiguousarray(v)
            O.lib().orc_surface_force(C.byref(S), dp(v), float(g["force_actuation"]), dp(F), dp(T))
            np.testing.assert_allclose(F, g["forces"][s, k], rtol=1e-12, atol=1e-13)
            np.testing.assert_allclose(T, g["torques"][s, k], rtol=1e-12, atol=1e-13)


def test_quadx_mixer_motors_drag(golden_dir):
    g = load(golden_dir, "quadx_components")
    P = O.make_params("quadx")
    np.testing.assert_allclose(np.array(P.max_rpm[:]), g["max_rpm"], rtol=1e-15)
    np.testing.assert_allclose(np.array(P.drag_const[:]), g["drag_consts"][0], rtol=1e-15)
    for cmd, ref in zip(g["mix_cmd"], g["mix_pwm"]):
        pwm = np.zeros(4)
        cmd = np.ascontiguousarray(cmd)
        O.lib().orc_quadx_mix(C.byref(P), dp(cmd), dp(pwm))
        np.testing.assert_allclose(pwm, ref, rtol=0, atol=1e-15)
    for v, f in zip(g["drag_v"], g["drag_f"]):
        out = np.zeros(3)
        v = np.ascontiguousarray(v)
        O.lib().orc_body_drag(C.byref(P), dp(v), dp(out))
        np.testing.assert_allclose(out, f, rtol=1e-15)
    # thrust/torque: orc_motors_update with dt/tau bypassed -> use pwm == throttle, xi == 0
    thrust_t = ((C.c_double * 3) * 4)
    for thr, th_ref, tq_ref in zip(g["throttle"], g["thrust"], g["torque"]):
        t = np.ascontiguousarray(thr.copy())
        th, tq = thrust_t(), thrust_t()
        O.lib().orc_motors_update.argtypes = [C.POINTER(O.Params), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                              C.c_double, thrust_t, thrust_t]
        O.lib().orc_motors_update(C.byref(P), dp(t), dp(t.copy()), 0.0, th, tq)
        np.testing.assert_allclose(t, thr, rtol=1e-15)  # pwm == throttle leaves throttle unchanged
        np.testing.assert_allclose(np.array([list(r) for r in th]), th_ref, rtol=1e-14, atol=1e-18)
        np.testing.assert_allclose(np.array([list(r) for r in tq]), tq_ref, rtol=1e-14, atol=1e-20)


# ------------------------------------------------------------------ Aviary level
def lane_state(L, fixedwing):
    st = np.array([list(L.w_b), list(L.rpy), list(L.v_b), list(L.p)])
    aux = np.array(list(L.actuation) + [L.throttle[0]]) if fixedwing else np.array(list(L.throttle))
    return st, aux


AVIARY = [f"aviary_quadx_mode{m}" for m in ["m1", 0, 1, 2, 3, 4, 5, 6, 7, "7_nonoise"]] + \
         ["aviary_fixedwing_mode0", "aviary_fixedwing_modem1"] + \
         [f"aviary_primitive_mode{m}" for m in (0, 6, 7)] + \
         ["aviary_acrowing_mode0", "aviary_acrowing_modem1"] + \
         ["aviary_quadx_land", "aviary_primitive_land"] + \
         ["aviary_quadx_mode6_hz60", "aviary_quadx_mode0_hz60", "aviary_fixedwing_mode0_hz60"]  # drone_options=dict(control_hz=60): four ticks per Aviary step
# (aviary_*_land: motors off, from first touch to rest: the contact response)


def model_of(name):
    if "acrowing" in name:
        return "acrowing"
    return "fixedwing" if "fixedwing" in name else ("primitive_drone" if "primitive" in name else "quadx")


@pytest.mark.parametrize("name", AVIARY)
def test_aviary_trajectory(golden_dir, name):
    g = load(golden_dir, name)
    fw = "fixedwing" in name or "acrowing" in name
    noise = bool(g["noise"])
    rate = {}
    if "control_hz" in g.files:  # base_drone.py:95-107: physics ticks per control update, the controllers' period
        hz = int(g["control_hz"])
        rate = dict(world_ticks_per_control=240 // hz, control_period=1.0 / hz)
    P = O.make_params(model_of(name), noise_mode=O.NOISE_INJECT if noise else O.NOISE_OFF,
                      start_pos=g["start_pos"], start_rpy=g["start_orn"], **rate)
    L = O.Lane()
    lib = O.lib()
    lib.orc_aviary_reset(C.byref(P), C.byref(L), 0)
    lib.orc_set_mode(C.byref(P), C.byref(L), int(g["mode"]))
    st, aux = lane_state(L, fw)
    np.testing.assert_allclose(st, g["init_state"], atol=1e-14)
    np.testing.assert_allclose(np.array(list(L.setpoint))[: len(g["init_setpoint"])], g["init_setpoint"], atol=1e-14)
    worst = 0.0
    for k in range(len(g["states"])):
        sp = g["setpoints"][k]
        for i, x in enumerate(sp):
            L.setpoint[i] = x
        xi = np.ascontiguousarray(np.nan_to_num(g["xi"][k]))
        lib.orc_aviary_step(C.byref(P), C.byref(L), dp(xi), 0, 0)
        st, aux = lane_state(L, fw)
        err = max(np.abs(st - g["states"][k]).max(), np.abs(aux - g["aux"][k]).max())
        worst = max(worst, err)
        assert bool(L.contact_step) == bool(g["contact"][k])
    # trajectories that touch the floor go through the contact solver's Gauss-Seidel sweeps, where the two independent
    # formulations (COM-based 3x3 here, 6x6 spatial inertia at the base origin in fake_bullet) round differently
    assert worst < (1e-7 if g["contact"].any() else TOL), worst


def wind_from_coef(c):
    """The analytic wind field of the wind goldens (coefficients stored in the fixture)."""
    def wind(time, position):
        w = np.zeros_like(position)
        w[:, 0] = c[0] + c[1] * np.sin(c[2] * time) + c[3] * position[:, 1]
        w[:, 1] = c[4] + c[5] * position[:, 2]
        w[:, 2] = c[6] * np.cos(c[7] * time) + c[8] * position[:, 0]
        return w

    return wind


@pytest.mark.parametrize("name", ["aviary_quadx_wind_register", "aviary_quadx_wind_ctor",
                                  "aviary_fixedwing_wind_ctor", "aviary_fixedwing_wind_register", "aviary_rocket_wind_ctor"])
def test_aviary_wind_trajectory(golden_dir, name):
    """Wind hook (aviary.py:266-285,324-333): sampled in update_state at the link positions with the
    Aviary's (lagging) elapsed time, subtracted from the link velocities that feed the body drag
    (boring_bodies.py:93-96) / the lifting surfaces (lifting_surfaces.py:88-93). wind_kind 1: the
    field is registered after construction, so the reset-time velocities are wind-free; 2: given to
    the constructor, sampled in reset()."""
    g = load(golden_dir, name)
    fw = "fixedwing" in name
    rocket = "rocket" in name
    extra = dict(starting_fuel_ratio=0.3) if rocket else {}
    P = O.make_params("rocket" if rocket else ("fixedwing" if fw else "quadx"), noise_mode=O.NOISE_INJECT,
                      start_pos=g["start_pos"], start_rpy=g["start_orn"], **extra)
    L = O.Lane()
    lib = O.lib()
    fn = wind_from_coef(g["wind_coef"])
    if rocket:
        def lane_state(L, _fw):  # noqa: F811 -- rocket aux layout (rocket.py:320-326)
            st = np.array([list(L.w_b), list(L.rpy), list(L.v_b), list(L.p)])
            return st, np.array(list(L.actuation)[:4] + [float(L.ignition), L.fuel_ratio, L.throttle[0]] + list(L.gimbal))
    else:
        lane_state = globals()["lane_state"]
    keep = None
    if int(g["wind_kind"]) == 2:
        keep = O.set_wind(P, fn)
    lib.orc_aviary_reset(C.byref(P), C.byref(L), 0)
    if int(g["wind_kind"]) == 1:
        keep = O.set_wind(P, fn)
    lib.orc_set_mode(C.byref(P), C.byref(L), int(g["mode"]))
    st, aux = lane_state(L, fw)
    np.testing.assert_allclose(st, g["init_state"], atol=1e-14)
    worst = 0.0
    for k in range(len(g["states"])):
        for i, x in enumerate(g["setpoints"][k]):
            L.setpoint[i] = x
        xi = np.ascontiguousarray(np.nan_to_num(g["xi"][k]))
        lib.orc_aviary_step(C.byref(P), C.byref(L), dp(xi), 0, 0)
        st, aux = lane_state(L, fw)
        worst = max(worst, np.abs(st - g["states"][k]).max(), np.abs(aux - g["aux"][k]).max())
    assert keep is not None
    assert worst < TOL, worst
    # and the wind matters: the same run without it must differ visibly
    P2 = O.make_params("rocket" if rocket else ("fixedwing" if fw else "quadx"), noise_mode=O.NOISE_INJECT, start_pos=g["start_pos"],
                       start_rpy=g["start_orn"], **extra)
    L2 = O.Lane()
    lib.orc_aviary_reset(C.byref(P2), C.byref(L2), 0)
    lib.orc_set_mode(C.byref(P2), C.byref(L2), int(g["mode"]))
    for k in range(len(g["states"])):
        for i, x in enumerate(g["setpoints"][k]):
            L2.setpoint[i] = x
        lib.orc_aviary_step(C.byref(P2), C.byref(L2), dp(np.ascontiguousarray(np.nan_to_num(g["xi"][k]))), 0, 0)
    assert np.abs(lane_state(L2, fw)[0] - g["states"][-1]).max() > 1e-3


@pytest.mark.parametrize("name", ["aviary_quadx_drop", "aviary_fixedwing_drop", "aviary_primitive_drop"])
def test_aviary_drop_contact(golden_dir, name):
    """Fall onto the floor: the contact flag must rise on the same Aviary step as in the reference-on-fake-Bullet run,
    and the landing itself -- impulses, friction, penetration recovery, coming to rest -- must match: the contact
    response is implemented twice, independently (oracle: COM-based; fake_bullet: spatial inertia at the base origin)."""
    g = load(golden_dir, name)
    fw = "fixedwing" in name
    noise = bool(g["noise"])
    P = O.make_params(model_of(name), noise_mode=O.NOISE_INJECT if noise else O.NOISE_OFF,
                      start_pos=g["start_pos"], start_rpy=g["start_orn"])
    L = O.Lane()
    lib = O.lib()
    lib.orc_aviary_reset(C.byref(P), C.byref(L), 0)
    lib.orc_set_mode(C.byref(P), C.byref(L), 0)
    assert g["contact"].any() and not g["contact"][0]
    for k in range(len(g["states"])):
        for i, x in enumerate(g["setpoints"][k]):
            L.setpoint[i] = x
        xi = np.ascontiguousarray(np.nan_to_num(g["xi"][k]))
        lib.orc_aviary_step(C.byref(P), C.byref(L), dp(xi), 0, 0)
        assert bool(L.contact_step) == bool(g["contact"][k]), k
        st, aux = lane_state(L, fw)
        np.testing.assert_allclose(st, g["states"][k], atol=TOL)


@pytest.mark.parametrize("name,fuel", [("aviary_rocket_default_fuel", 0.05), ("aviary_rocket_fuel60", 0.6), ("aviary_rocket_drop", 0.0),
                                       ("aviary_rocket_land", 0.0)])
def test_rocket_trajectory(golden_dir, name, fuel):
    """Rocket (drones/rocket.py + boosters.py + gimbals.py): grid fins, gimballed booster with fuel burn
    (the composite mass / centre of mass / inertia are rebuilt every tick, as changeDynamics does in
    the reference), per-axis body drag; the drop run ends with a leg / the booster on the floor."""
    g = load(golden_dir, name)
    noise = bool(g["noise"])
    P = O.make_params("rocket", noise_mode=O.NOISE_INJECT if noise else O.NOISE_OFF, start_pos=g["start_pos"],
                      start_rpy=g["start_orn"], starting_fuel_ratio=fuel)
    L = O.Lane()
    lib = O.lib()
    lib.orc_aviary_reset(C.byref(P), C.byref(L), 0)
    lib.orc_set_mode(C.byref(P), C.byref(L), 0)

    def state():
        st = np.array([list(L.w_b), list(L.rpy), list(L.v_b), list(L.p)])
        aux = np.array(list(L.actuation)[:4] + [float(L.ignition), L.fuel_ratio, L.throttle[0]] + list(L.gimbal))
        return st, aux

    st, aux = state()
    np.testing.assert_allclose(st, g["init_state"], atol=1e-14)
    np.testing.assert_allclose(aux, g["init_aux"], atol=1e-14)
    worst = 0.0
    first_contact = int(np.argmax(g["contact"])) if g["contact"].any() else len(g["states"])
    for k in range(len(g["states"])):
        for i, x in enumerate(g["setpoints"][k]):
            L.setpoint[i] = x
        xi = np.ascontiguousarray(np.nan_to_num(g["xi"][k]))
        lib.orc_aviary_step(C.byref(P), C.byref(L), dp(xi), 0, 0)
        assert bool(L.contact_step) == bool(g["contact"][k]), k
        st, aux = state()
        scale = np.maximum(1.0, np.abs(g["states"][k]))
        worst = max(worst, (np.abs(st - g["states"][k]) / scale).max(), np.abs(aux - g["aux"][k]).max())
    assert worst < (1e-7 if g["contact"].any() else TOL), worst  # (contact solver: see test_aviary_trajectory)
    if "drop" in name:
        assert first_contact < len(g["states"]) and not g["contact"][0]


# ------------------------------------------------------------------ env level
ENVS = [
    ("env_hover_random", "hover", {}),
    ("env_hover_gentle_trunc", "hover", {"max_steps": 20}),
    ("env_hover_euler_sparse", "hover", {"angle_repr": 0, "sparse_reward": 1}),
    ("env_hover_crash", "hover", {}),
    ("env_hover_crash_detect_only", "hover", {"world_contact_response": 0}),
    ("env_quadx_waypoints_random", "quadx_waypoints", {}),
    ("env_quadx_waypoints_reach", "quadx_waypoints", {"goal_reach_distance": 2.5}),
    ("env_fixedwing_waypoints_random", "fixedwing_waypoints", {}),
    ("env_fixedwing_waypoints_gentle", "fixedwing_waypoints", {"goal_reach_distance": 40.0}),
    ("env_quadx_waypoints_yaw_random", "quadx_waypoints", {"use_yaw_targets": 1}),
    ("env_quadx_waypoints_yaw_reach", "quadx_waypoints", {"use_yaw_targets": 1, "goal_reach_distance": 2.5, "goal_reach_angle": 1.2}),
    # every other flight mode (quadx.py:233-373,437-479), 1.5 s episodes (max_steps = 40 Hz x 1.5 s)
    *[(f"env_hover_mode{'m1' if m == -1 else m}", "hover", {"flight_mode": m, "max_steps": 60}) for m in (-1, 1, 2, 3, 4, 5, 6, 7)],
    ("env_quadx_waypoints_mode7", "quadx_waypoints", {"flight_mode": 7, "goal_reach_distance": 0.4}),
    # constructor options away from their defaults (gen_goldens.py: gen_envs_options): agent_hz -> env_step_ratio = 120 / agent_hz and
    # max_steps = seconds x agent_hz, flight_dome_size -> dome, num_targets, sparse_reward, Euler observations, goal_reach_distance
    ("env_hover_opts", "hover", {"env_step_ratio": 4, "max_steps": 45, "dome": 2.0}),
    ("env_quadx_waypoints_opts", "quadx_waypoints", {"num_targets": 2, "sparse_reward": 1, "dome": 4.0, "env_step_ratio": 2, "max_steps": 240,
                                                     "goal_reach_distance": 1.5, "angle_repr": 0}),
    ("env_fixedwing_waypoints_opts", "fixedwing_waypoints", {"num_targets": 3, "sparse_reward": 1, "dome": 60.0, "env_step_ratio": 3, "max_steps": 800,
                                                             "goal_reach_distance": 30.0, "angle_repr": 0}),
]


@pytest.mark.parametrize("name,env,over", ENVS)
def test_env_trajectory(golden_dir, name, env, over):
    g = load(golden_dir, name)
    P = O.make_params(env, noise_mode=O.NOISE_INJECT, **over)
    lib = O.lib()
    D = lib.orc_obs_dim(C.byref(P))
    assert D == g["obs"].shape[1]
    L = O.Lane()
    resets = set(int(k) for k in g["reset_before"])
    ri = 0

    def do_reset():
        nonlocal ri
        xr = np.ascontiguousarray(g["reset_xi"][ri])
        u = np.ascontiguousarray(g["reset_u"][ri])
        lib.orc_env_reset(C.byref(P), C.byref(L), 0, dp(xr), dp(u))
        obs = np.frombuffer(L.obs, dtype=np.float64, count=D)
        np.testing.assert_allclose(obs, g["reset_obs"][ri], atol=TOL)
        ri += 1

    do_reset()
    seen = dict(term=0, trunc=0, col=0, oob=0, complete=0, reached=0)
    for k in range(len(g["action"])):
        if k in resets:
            do_reset()
        a = np.ascontiguousarray(g["action"][k])
        xi = np.ascontiguousarray(np.nan_to_num(g["xi"][k]))
        lib.orc_env_step(C.byref(P), C.byref(L), dp(a), dp(xi))
        obs = np.frombuffer(L.obs, dtype=np.float64, count=D)
        np.testing.assert_allclose(obs, g["obs"][k], atol=TOL, err_msg=f"step {k}")
        assert abs(L.reward - g["reward"][k]) < 1e-9, (k, L.reward, g["reward"][k])
        assert bool(L.terminated) == bool(g["term"][k]), k
        assert bool(L.truncated) == bool(g["trunc"][k]), k
        assert bool(L.info_oob) == bool(g["info_oob"][k])
        assert bool(L.info_collision) == bool(g["info_col"][k])
        assert bool(L.info_complete) == bool(g["info_complete"][k])
        assert int(L.num_targets_reached) == int(g["info_ntr"][k])
        seen["term"] += int(g["term"][k]); seen["trunc"] += int(g["trunc"][k])
        seen["col"] += int(g["info_col"][k]); seen["oob"] += int(g["info_oob"][k])
        seen["complete"] += int(g["info_complete"][k]); seen["reached"] = max(seen["reached"], int(g["info_ntr"][k]))
    assert ri == len(g["reset_obs"])
    print(name, seen)


def test_ma_quadx_hover_trajectory(golden_dir):
    """pz_envs MAQuadXHoverEnv through its dict API (4 agents as 4 independent lanes)."""
    g = load(golden_dir, "env_ma_quadx_hover")
    lib = O.lib()
    n = g["start_pos"].shape[0]
    Ps = [O.make_params("ma_hover", noise_mode=O.NOISE_INJECT, start_pos=g["start_pos"][i], start_rpy=g["start_orn"][i],
                        dome=float(g["dome"]), max_steps=int(g["max_steps"])) for i in range(n)]
    Ls = [O.Lane() for _ in range(n)]
    D = lib.orc_obs_dim(C.byref(Ps[0]))
    assert D == 24
    resets = set(int(k) for k in g["reset_before"])
    ri = 0

    def do_reset():
        nonlocal ri
        for i in range(n):
            xr = np.ascontiguousarray(g["reset_xi"][ri][:, i])
            lib.orc_env_reset(C.byref(Ps[i]), C.byref(Ls[i]), i, dp(xr), None)
            obs = np.frombuffer(Ls[i].obs, dtype=np.float64, count=D)
            np.testing.assert_allclose(obs, g["reset_obs"][ri][i], atol=TOL)
        ri += 1

    do_reset()
    seen_term = 0
    for k in range(len(g["action"])):
        if k in resets:
            do_reset()
        for i in range(n):
            a = np.ascontiguousarray(g["action"][k][i])
            xi = np.ascontiguousarray(g["xi"][k][:, i])
            lib.orc_env_step(C.byref(Ps[i]), C.byref(Ls[i]), dp(a), dp(xi))
            if g["alive"][k][i]:
                obs = np.frombuffer(Ls[i].obs, dtype=np.float64, count=D)
                np.testing.assert_allclose(obs, g["obs"][k][i], atol=TOL, err_msg=f"step {k} agent {i}")
                assert abs(Ls[i].reward - g["reward"][k][i]) < 1e-9
                assert bool(Ls[i].terminated) == bool(g["term"][k][i]) and bool(Ls[i].truncated) == bool(g["trunc"][k][i])
                seen_term += int(g["term"][k][i])
    assert seen_term >= 4 and ri == len(g["reset_obs"])


@pytest.mark.parametrize("name", ["env_ma_quadx_hover_shared", "env_ma_quadx_hover_stack"])
def test_ma_quadx_hover_shared_world_trajectory(golden_dir, name):
    """The PettingZoo env with everything its SHARED world adds: two agents fly into each other -- the hit
    enters contact_array[drone.Id] and ends both episodes (ma_quadx_hover_env.py:181) -- and a dead drone on the floor
    switches off the rotational drag of every drone in the world (quadx.py:509); the drones push each other (the pair stage of
    the contact response: `stack` is a culled drone falling onto a live one, ma_quadx_base_env.py:365-369). Recorded from the
    reference's env on fake_bullet (6x6 spatial-inertia formulation), replayed through the oracle's world-level step (COM-based
    3x3 formulation): every alive agent's observation, and the pose of EVERY drone -- culled ones included -- after every step."""
    g = load(golden_dir, name)
    A = g["start_pos"].shape[0]
    Ps = [O.make_params("ma_hover", noise_mode=O.NOISE_INJECT, start_pos=g["start_pos"][i], start_rpy=g["start_orn"][i], dome=float(g["dome"]),
                        max_steps=int(g["max_steps"]), world_contact_response=1) for i in range(A)]
    W = O.OracleWorld(Ps)
    resets = set(int(k) for k in g["reset_before"])
    ri = 0

    def do_reset():
        nonlocal ri
        obs = W.reset(xi_reset=g["reset_xi"][ri].T)
        np.testing.assert_allclose(obs, g["reset_obs"][ri], atol=TOL)
        ri += 1

    do_reset()
    hits = 0
    for k in range(len(g["action"])):
        if k in resets:
            do_reset()
        obs, rew, term, trunc = W.step(g["action"][k], xi=g["xi"][k].T)
        for i in range(A):
            if g["alive"][k][i]:
                np.testing.assert_allclose(obs[i], g["obs"][k][i], atol=1e-8, err_msg=f"step {k} agent {i}")
                assert abs(rew[i] - g["reward"][k][i]) < 1e-8
                assert bool(term[i]) == bool(g["term"][k][i]) and bool(trunc[i]) == bool(g["trunc"][k][i]), (k, i)
        assert bool(any(L.contact_now for L in W.Ls)) == bool(g["world_contact"][k]), k
        pos = np.array([[L.p[0], L.p[1], L.p[2]] for L in W.Ls])
        rpy = np.array([[L.rpy[0], L.rpy[1], L.rpy[2]] for L in W.Ls])
        np.testing.assert_allclose(pos, g["all_pos"][k], atol=1e-8, err_msg=f"step {k}: positions of all drones")
        np.testing.assert_allclose(rpy, g["all_rpy"][k], atol=1e-7, err_msg=f"step {k}: attitudes of all drones")
        hits += int(g["drone_contact"][k].any())
    assert hits > 0 and ri == len(g["reset_obs"])
    if name != "env_ma_quadx_hover_shared":
        return
    # the same actions with every agent alone in its own world: no hit, different outcome
    Ws = [O.OracleWorld([p]) for p in Ps]
    for i, w in enumerate(Ws):
        w.reset(xi_reset=g["reset_xi"][0].T[i:i + 1])
    k_hit = int(np.argmax(g["drone_contact"].any(1)))
    for k in range(k_hit + 1):
        outs = [w.step(g["action"][k][i:i + 1], xi=g["xi"][k].T[i:i + 1]) for i, w in enumerate(Ws)]
    assert not outs[0][2][0] and not outs[1][2][0]  # alone, agents 0 and 1 do not terminate at the step of the hit


@pytest.mark.parametrize("name", ["env_dogfight_default", "env_dogfight_engage", "env_dogfight_crash", "env_dogfight_team1_sparse", "env_dogfight_team3",
                                  "env_dogfight_unassisted", "env_dogfight_midair"])
def test_dogfight_trajectory(golden_dir, name):
    """MAFixedwingDogfightEnv (ma_fixedwing_dogfight_env.py) recorded from the reference's env on fake_bullet, replayed through
    orc_dogfight_*: observation (self + the others in the own body frame, inactive aircraft dropped, zero padded), the
    accumulate-then-pop rewards (fp32 in the reference: 1e-6 relative), health / hits, the collision / out-of-bounds / element-wise
    team-win overrides, culling, aircraft that fly on with zero commands after they were culled."""
    g = load(golden_dir, name)
    W = O.OracleDogfight(g["start_pos"], g["start_orn"], noise_mode=O.NOISE_INJECT, team_size=int(g["team_size"]),
                         damage_per_hit=float(g["damage_per_hit"]), lethal_distance=float(g["lethal_distance"]), lethal_angle=float(g["lethal_angle"]),
                         aggressiveness=float(g["aggressiveness"]), cooperativeness=float(g["cooperativeness"]), sparse_reward=bool(g["sparse_reward"]),
                         dome=float(g["dome"]), max_duration_seconds=int(g["max_steps"]) / 30.0, assisted_flight=int(g["action_dim"]) == 4)
    np.testing.assert_allclose(W.reset(xi_reset=g["reset_xi"].T), g["reset_obs"], atol=TOL)
    bits = np.zeros(W.A, dtype=int)
    for k in range(len(g["action"])):
        alive = g["alive"][k]
        assert (W.alive == alive).all(), k
        obs, rew, term, trunc = W.step(g["action"][k], xi=g["xi"][k].T)
        for i in range(W.A):
            if alive[i]:
                np.testing.assert_allclose(obs[i], g["obs"][k][i], atol=1e-8, err_msg=f"step {k} agent {i}")
                assert abs(rew[i] - g["reward"][k][i]) <= 1e-6 * max(1.0, abs(g["reward"][k][i])), (k, i, rew[i], g["reward"][k][i])
                assert bool(term[i]) == bool(g["term"][k][i]) and bool(trunc[i]) == bool(g["trunc"][k][i]), (k, i)
                assert (int(W.D.info_bits[i]) & int(g["info_bits"][k][i])) == int(g["info_bits"][k][i]), (k, i)
                bits[i] |= int(g["info_bits"][k][i])
        np.testing.assert_allclose(W.health, g["health"][k], atol=1e-6)
        assert (np.array(W.D.received_hits[:W.A]) == g["received_hits"][k]).all()
    if name == "env_dogfight_engage":
        assert g["received_hits"][-1].sum() > 50 and (bits & 1).any() and (bits & 8).any()  # hits, deaths, team wins
    if name == "env_dogfight_midair":
        # two aircraft met in mid-air: both out in the same step with the collision bit, and what the survivors see of the wrecks
        # afterwards is the contact response BETWEEN the aircraft -- the control recording without it parts from this one by metres
        assert (bits[[0, 2]] & 2).all() and not (bits[[1, 3]] & 2).any()
        g0 = load(golden_dir, "env_dogfight_midair_nopair")
        k0 = int(np.argmax(g["term"][:, 0]))
        assert np.abs(np.nan_to_num(g["obs"][:k0]) - np.nan_to_num(g0["obs"][:k0])).max() < 1e-9  # the same flight until the step of the hit
        assert np.abs(np.nan_to_num(g["obs"][k0 + 5:, [1, 3]]) - np.nan_to_num(g0["obs"][k0 + 5:, [1, 3]])).max() > 1.0
    if name == "env_dogfight_crash":
        assert (bits & 4).any() and g["trunc"].any()  # out of bounds, truncation
        assert any(W.D.inactive[:W.A])  # a dead aircraft at rest on the ground has dropped out of the observations


def test_dogfight_spawn_against_reference_output(golden_dir):
    """orc_dogfight_spawn against what the reference's own `_get_start_pos_orn(seed=3)` returned when the default fixture was
    recorded (its start_pos / start_orn): the uniforms are re-drawn from np.random.RandomState(3) in the reference's draw order."""
    g = load(golden_dir, "env_dogfight_default")
    team, lo, hi = int(g["team_size"]), 10.0, 50.0
    rs = np.random.RandomState(seed=3)
    u0 = rs.uniform(0.0, 2 * np.pi) / (2 * np.pi)
    ur = (rs.uniform(low=lo, high=hi, size=(2 * team,)) - lo) / (hi - lo)
    uh = (rs.uniform(low=lo, high=hi, size=(2 * team,)) - lo) / (hi - lo)
    uy = rs.random(2 * team)
    pos, rpy, vel = O.dogfight_spawn(team, lo, hi, np.concatenate([[u0], ur, uh, uy]))
    np.testing.assert_allclose(pos, g["start_pos"], atol=1e-10)
    np.testing.assert_allclose(rpy, g["start_orn"], atol=1e-10)


def test_dogfight_spawn_restatement():
    """orc_dogfight_spawn against the reference's _get_start_pos_orn arithmetic (ma_fixedwing_dogfight_env.py:176-213), fed the
    draws of the same np.random.RandomState."""
    team, lo, hi, seed = 2, 10.0, 50.0, 11
    rs = np.random.RandomState(seed=seed)
    u0 = rs.uniform(0.0, 2 * np.pi) / (2 * np.pi)
    ur = (rs.uniform(low=lo, high=hi, size=(2 * team,)) - lo) / (hi - lo)
    uh = (rs.uniform(low=lo, high=hi, size=(2 * team,)) - lo) / (hi - lo)
    uy = rs.random(2 * team)
    pos, rpy, vel = O.dogfight_spawn(team, lo, hi, np.concatenate([[u0], ur, uh, uy]))
    rs = np.random.RandomState(seed=seed)  # the reference's lines, verbatim arithmetic
    start_radian = np.pi / team * np.arange(team * 2) + rs.uniform(0.0, 2 * np.pi)
    start_radius = rs.uniform(low=lo, high=hi, size=(team * 2,))
    start_height = rs.uniform(low=lo, high=hi, size=(team * 2,))
    yaw = start_radian + rs.random(2 * team) * np.pi / 8.0
    np.testing.assert_allclose(pos, np.stack([start_radius * np.cos(start_radian), start_radius * np.sin(start_radian), start_height], axis=1), atol=1e-12)
    np.testing.assert_allclose(rpy[:, 2], yaw, atol=1e-12)
    np.testing.assert_allclose(vel, 20.0 * np.stack([np.cos(yaw), np.sin(yaw), 0 * yaw], axis=1), atol=1e-12)
