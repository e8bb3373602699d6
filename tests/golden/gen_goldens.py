"""gen_goldens.py -- generates the committed golden vectors under tests/golden/*.npz.

RUN ONLY IN THE BUILD CONTAINER (needs /root/reference):  python tests/golden/gen_goldens.py

It imports the reference's own Python (PyFlyt @ /root/reference) under the module stubs of
ref_stubs.py and drives the *real* reference classes -- PID, Motors, BoringBodies, LiftingSurface,
QuadX, Fixedwing, Aviary, QuadXHoverEnv, QuadXWaypointsEnv, FixedwingWaypointsEnv -- recording
inputs (actions, every RNG draw) and outputs (states, observations, rewards, flags).

What these vectors pin: all PyFlyt-side arithmetic and env semantics. What they do NOT pin:
PyBullet itself -- `pybullet` is replaced by oracle/fake_bullet.py (our restatement), so the
Bullet boundary stays "parity unpinned" (SURVEY.md section 8(c)).

The .npz files hold data only (inputs + expected outputs); no reference source travels.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

gym = ref_stubs.install()

from PyFlyt.core import Aviary  # noqa: E402
from PyFlyt.core.abstractions.pid import PID  # noqa: E402
from PyFlyt.gym_envs.fixedwing_envs.fixedwing_waypoints_env import FixedwingWaypointsEnv  # noqa: E402
from PyFlyt.gym_envs.quadx_envs.quadx_hover_env import QuadXHoverEnv  # noqa: E402
from PyFlyt.gym_envs.quadx_envs.quadx_waypoints_env import QuadXWaypointsEnv  # noqa: E402


OUT_DIR = HERE  # (--out DIR: write somewhere else -- tests/test_golden_regen.py regenerates into a scratch directory and compares)


def save(name, **arrays):
    path = os.path.join(OUT_DIR, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path}: " + ", ".join(f"{k}{np.asarray(v).shape}" for k, v in arrays.items()))


# --------------------------------------------------------------------------- components
def gen_pid():
    rng = np.random.default_rng(11)
    out = {}
    gains = {
        "ang_vel": ([4.0e-2, 4.0e-2, 8.0e-2], [5.0e-7, 5.0e-7, 2.7e-4], [1.0e-4, 1.0e-4, 0.0], [1.0, 1.0, 1.0]),
        "lin_vel": ([0.8, 0.8], [0.3, 0.3], [0.5, 0.5], [0.4, 0.4]),
        "z_vel": ([2.0], [0.5], [0.05], [1.0]),
    }
    for name, (kp, ki, kd, lim) in gains.items():
        n = len(kp)
        pid = PID(np.array(kp), np.array(ki), np.array(kd), np.array(lim), 1.0 / 120.0)
        states = rng.normal(0, 1.5, size=(60, n))
        sps = rng.normal(0, 1.5, size=(60, n))
        outs = np.stack([pid.step(s, sp) for s, sp in zip(states, sps)])
        out[f"{name}_gains"] = np.array([kp, ki, kd, lim])
        out[f"{name}_state"] = states
        out[f"{name}_setpoint"] = sps
        out[f"{name}_out"] = outs
    save("pid", **out)


def gen_aero_and_motors():
    """Real LiftingSurface / Motors / QuadX objects (constructed by the reference from its YAML)."""
    env = Aviary(start_pos=np.array([[0.0, 0.0, 10.0]]), start_orn=np.zeros((1, 3)), drone_type="fixedwing",
                 np_random=ref_stubs.RecordingRNG(np.random.default_rng(0)))
    fw = env.drones[0]
    alphas = np.concatenate([np.linspace(-np.pi, np.pi, 73), np.deg2rad([-9.0, -8.99, 13.99, 14.0, 14.01, 9.0, 89.9, -89.9])])
    acts = np.array([-1.0, -0.35, 0.0, 0.5, 1.0])
    coeffs = np.zeros((5, len(acts), len(alphas), 3))
    rng = np.random.default_rng(5)
    vels = np.concatenate([rng.normal(0, 12.0, size=(40, 3)), np.array([[20.0, 0, -1.0], [20.0, 0.5, 3.0], [-5.0, 1.0, 2.0]])])
    forces = np.zeros((5, len(vels), 3))
    torques = np.zeros((5, len(vels), 3))
    consts = np.zeros((5, 6))
    for s, surf in enumerate(fw.lifting_surfaces.surfaces):
        consts[s] = [surf.area, surf.aspect, surf.Cl_alpha_3D, surf.theta_f, surf.aero_tau, surf.half_rho]
        for a, act in enumerate(acts):
            for k, al in enumerate(alphas):
                coeffs[s, a, k] = surf._jitted_compute_aero_data(
                    al, surf.aspect, surf.flap_to_chord, surf.aero_tau, act, surf.deflection_limit, surf.eta,
                    surf.Cl_alpha_3D, surf.alpha_stall_P_base, surf.alpha_0_base, surf.alpha_stall_N_base, surf.Cd_0)
        for k, v in enumerate(vels):
            act = 0.3
            alpha, V = surf._compute_aoa_freestream(v, surf.lift_unit, surf.drag_unit)
            Cl, Cd, CM = surf._jitted_compute_aero_data(
                alpha, surf.aspect, surf.flap_to_chord, surf.aero_tau, act, surf.deflection_limit, surf.eta,
                surf.Cl_alpha_3D, surf.alpha_stall_P_base, surf.alpha_0_base, surf.alpha_stall_N_base, surf.Cd_0)
            forces[s, k], torques[s, k] = surf._jitted_compute_force_torque(
                alpha, V, Cl, Cd, CM, surf.half_rho, surf.area, surf.chord, surf.lift_unit, surf.drag_unit, surf.torque_unit)
    save("aero", alphas=alphas, actuations=acts, coeffs=coeffs, vels=vels, forces=forces, torques=torques,
         force_actuation=0.3, consts=consts)

    # motors + mixer + drag on a real QuadX
    env = Aviary(start_pos=np.array([[0.0, 0.0, 1.0]]), start_orn=np.zeros((1, 3)), drone_type="quadx",
                 np_random=ref_stubs.RecordingRNG(np.random.default_rng(0)))
    q = env.drones[0]
    thr = np.linspace(-1.0, 1.0, 21)[:, None] * np.array([1.0, 0.9, 0.8, 0.7])
    thrust = np.zeros((len(thr), 4, 3))
    torque = np.zeros((len(thr), 4, 3))
    for i, t in enumerate(thr):
        thrust[i], torque[i] = q.motors._jitted_compute_thrust_torque(
            None, t, q.motors.max_rpm, q.motors.thrust_unit, q.motors.thrust_coef, q.motors.torque_coef)
    # mixer/saturation through update_control in mode 0 with zero rate error -> cmd = [PID(...), T]
    rng = np.random.default_rng(9)
    cmds = np.concatenate([rng.uniform(-1.2, 1.2, size=(80, 4)), np.zeros((1, 4)), np.ones((1, 4)) * 0.3])
    cmds[:, 3] = np.abs(cmds[:, 3])
    pwms = np.zeros_like(cmds)
    for i, c in enumerate(cmds):
        # drive the reference's own mixing/saturation branch (quadx.py:482-493): with zero state,
        # unit kp, zero ki/kd and wide limits the rate PID passes the setpoint through unchanged
        q.set_mode(0)
        q.state = np.zeros((4, 3))
        q.PIDs[0].kp = np.ones(3)
        q.PIDs[0].ki = np.zeros(3)
        q.PIDs[0].kd = np.zeros(3)
        q.PIDs[0].limits = np.ones(3) * 10.0
        q.setpoint = np.array([c[0], c[1], c[2], c[3]])
        q.update_control(0)
        pwms[i] = q.pwm
    drag_v = rng.normal(0, 3.0, size=(20, 3))
    drag_f = -np.sign(drag_v) * q.body.drag_consts[0] * drag_v**2
    save("quadx_components", throttle=thr, thrust=thrust, torque=torque, max_rpm=q.motors.max_rpm,
         mix_cmd=np.concatenate([cmds[:, :3], np.clip(cmds[:, 3:], 0, 1)], axis=1), mix_pwm=pwms,
         drag_v=drag_v, drag_f=drag_f, drag_consts=q.body.drag_consts)


# --------------------------------------------------------------------------- Aviary level
# wind field used for the wind goldens: time- and position-dependent, coefficients travel in the .npz
WIND_COEF = np.array([1.5, 0.5, 2.0, 0.1, -0.8, 0.2, 0.3, 1.3, -0.05])


def wind_from_coef(c):
    def wind(time, position):
        w = np.zeros_like(position)
        w[:, 0] = c[0] + c[1] * np.sin(c[2] * time) + c[3] * position[:, 1]
        w[:, 1] = c[4] + c[5] * position[:, 2]
        w[:, 2] = c[6] * np.cos(c[7] * time) + c[8] * position[:, 0]
        return w

    return wind


def run_aviary(drone_type, mode, n_steps, seed, start_pos, start_orn, noise=True, drone_options=None, wind=None, hold_setpoint=None):
    """wind: None | "register" (Aviary.register_wind_field_function after construction, as
    tests/test_core.py:266-296 does: the first tick still sees wind-free velocities) | "ctor"
    (wind_type=<class>, as tests/test_core.py:299-340: the wind is sampled in reset())."""
    rng_env = ref_stubs.RecordingRNG(np.random.default_rng(seed))
    if not noise:
        rng_env.normal = lambda *a, **k: 0.0
    extra = {}
    if wind == "ctor":
        class GoldenWind:
            def __init__(self, np_random=None):
                self.fn = wind_from_coef(WIND_COEF)

            def __call__(self, time, position):
                return self.fn(time, position)

        extra = dict(wind_type=GoldenWind)
    env = Aviary(start_pos=np.array([start_pos]), start_orn=np.array([start_orn]), drone_type=drone_type,
                 np_random=rng_env, drone_options=drone_options or {}, **extra)
    if wind == "register":
        env.register_wind_field_function(wind_from_coef(WIND_COEF))
    env.set_mode(mode)
    rng = np.random.default_rng(seed + 1000)
    sp_dim = 7 if drone_type == "rocket" else (4 if not (drone_type == "fixedwing" and mode == -1) else 6)
    states, auxs, sps, xis, contacts = [], [], [], [], []
    init_state, init_aux, init_sp = env.state(0).copy(), env.aux_state(0).copy(), env.drones[0].setpoint.copy()
    sp = np.array(env.drones[0].setpoint, dtype=np.float64).copy()
    if hold_setpoint is not None:
        sp = np.array(hold_setpoint, dtype=np.float64)
        env.set_setpoint(0, sp.copy())
    for k in range(n_steps):
        if k % 25 == 10 and hold_setpoint is None:
            if drone_type == "quadx":
                if mode == -1:
                    sp = rng.uniform(0.1, 0.6, size=4)
                elif mode == 0:
                    sp = np.array([*rng.uniform(-1.0, 1.0, size=3), rng.uniform(0.2, 0.6)])
                elif mode == 1:
                    sp = np.array([*rng.uniform(-0.4, 0.4, size=3), rng.uniform(-0.5, 0.5)])
                elif mode == 2:
                    sp = np.array([*rng.uniform(-0.5, 0.5, size=3), rng.uniform(0.5, 2.0)])
                elif mode == 3:
                    sp = np.array([*rng.uniform(-0.3, 0.3, size=3), rng.uniform(0.5, 2.0)])
                elif mode == 4:
                    sp = np.array([*rng.uniform(-1.0, 1.0, size=2), rng.uniform(-0.5, 0.5), rng.uniform(0.5, 2.0)])
                elif mode in (5, 6):
                    sp = np.array([*rng.uniform(-1.0, 1.0, size=2), rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5)])
                elif mode == 7:
                    sp = np.array([*rng.uniform(-2.0, 2.0, size=2), rng.uniform(-1.0, 1.0), rng.uniform(0.5, 2.5)])
            elif drone_type == "rocket":  # rocket.py:230-236: fins x, y, yaw | ignition | throttle | gimbal 1, 2
                sp = np.concatenate([rng.uniform(-0.6, 0.6, size=3), [float(rng.random() < 0.8)], rng.uniform(0.0, 1.0, size=1),
                                     rng.uniform(-1.0, 1.0, size=2)])
            else:
                sp = rng.uniform(-1.0, 1.0, size=sp_dim)
                sp[-1] = rng.uniform(0.0, 1.0)
            env.set_setpoint(0, sp.copy())
        env.step()
        states.append(env.state(0).copy())
        auxs.append(env.aux_state(0).copy())
        sps.append(np.array(env.drones[0].setpoint, dtype=np.float64).copy())
        x = rng_env.drain("normal") if noise else np.zeros(0)
        xis.append(np.concatenate([x, np.full(env.updates_per_step - len(x), np.nan)]))  # (one draw per physics tick of the step: aviary.py:510)
        contacts.append(bool(np.any(env.contact_array)))
    hz = int(round(1.0 / env.drones[0].control_period))
    rate = {} if hz == 120 else dict(control_hz=np.array(hz))  # (recorded where it is not the default: the older fixtures stay byte for byte)
    return dict(**rate, states=np.array(states), aux=np.array(auxs), setpoints=np.array(sps), xi=np.array(xis),
                contact=np.array(contacts), init_state=init_state, init_aux=init_aux, init_setpoint=init_sp,
                mode=mode, start_pos=np.array(start_pos), start_orn=np.array(start_orn), noise=noise,
                wind_kind=np.array({None: 0, "register": 1, "ctor": 2}[wind]), wind_coef=WIND_COEF)


def gen_aviary():
    for mode in range(-1, 8):
        d = run_aviary("quadx", mode, 200, seed=100 + mode, start_pos=[0.3, -0.2, 1.5],
                       start_orn=[0.05, -0.08, 0.6], noise=True)
        save(f"aviary_quadx_mode{mode}".replace("-1", "m1"), **d)
    d = run_aviary("quadx", 7, 120, seed=7, start_pos=[0.0, 0.0, 1.0], start_orn=[0, 0, 0], noise=False)
    save("aviary_quadx_mode7_nonoise", **d)
    # a drop onto the floor: contact reporting (no contact response is restated -> keep only up to contact)
    d = run_aviary("quadx", 0, 150, seed=3, start_pos=[0.0, 0.0, 0.15], start_orn=[0.3, 0.2, 0.0], noise=False)
    save("aviary_quadx_drop", **d)
    for mode in (0, -1):
        d = run_aviary("fixedwing", mode, 200, seed=200 + mode, start_pos=[0.0, 0.0, 10.0], start_orn=[0.02, 0.05, -0.3], noise=True)
        save(f"aviary_fixedwing_mode{mode}".replace("-1", "m1"), **d)


def gen_control_rate():
    """drone_options=dict(control_hz=60) (base_drone.py:95-107, aviary.py:288-290,510-531): the Aviary step is FOUR physics ticks, the
    controller runs on the first of them with T = 1/60 in its integral and derivative terms, the motor commands are held for the other
    three. (The cascaded modes' gains are tuned for 120 Hz: at 60 Hz the rate loop swings between the motor limits and the closed loop
    amplifies round-off by a decade every few steps -- numpy against C, 1e-11 after forty steps and 1e-3 after a hundred -- so the
    cascaded recording is thirty steps long; mode 0, the rate loop alone, holds 1e-10 over its 120.)"""
    save("aviary_quadx_mode6_hz60", **run_aviary("quadx", 6, 30, seed=31, start_pos=[0.0, 0.0, 1.5], start_orn=[0.0, 0.0, 0.3], drone_options=dict(control_hz=60)))
    save("aviary_quadx_mode0_hz60", **run_aviary("quadx", 0, 120, seed=32, start_pos=[0.0, 0.0, 2.5], start_orn=[0.05, -0.05, 0.0], drone_options=dict(control_hz=60)))
    save("aviary_fixedwing_mode0_hz60", **run_aviary("fixedwing", 0, 150, seed=33, start_pos=[0.0, 0.0, 10.0], start_orn=[0.0, 0.0, 0.0], drone_options=dict(control_hz=60)))


def gen_primitive():
    # QuadX(drone_model="primitive_drone") (quadx.py:29): another parameter set + cylinder prop colliders
    opts = dict(drone_model="primitive_drone")
    for mode in (0, 6, 7):
        d = run_aviary("quadx", mode, 150, seed=300 + mode, start_pos=[0.3, -0.2, 1.5], start_orn=[0.05, -0.08, 0.6],
                       noise=True, drone_options=opts)
        save(f"aviary_primitive_mode{mode}", **d)
    # a tilted drop: a prop disc (cylinder) reaches the floor before the base box does
    d = run_aviary("quadx", 0, 120, seed=5, start_pos=[0.0, 0.0, 0.30], start_orn=[0.5, 0.2, 0.0], noise=False, drone_options=opts)
    save("aviary_primitive_drop", **d)


def gen_acrowing():
    # Fixedwing(drone_model="acrowing"): the aerobatic airframe of the dogfight env (ma_fixedwing_base_env.py:193-195)
    for mode in (0, -1):
        d = run_aviary("fixedwing", mode, 200, seed=600 + mode, start_pos=[0.0, 0.0, 30.0], start_orn=[0.02, 0.05, -0.3],
                       noise=True, drone_options=dict(drone_model="acrowing"))
        save(f"aviary_acrowing_mode{mode}".replace("-1", "m1"), **d)


def gen_rocket():
    # Rocket (drones/rocket.py, abstractions/boosters.py, gimbals.py): booster with fuel burn (variable mass
    # and inertia through changeDynamics), 2-axis thrust gimbal, four grid fins, per-axis body drag
    d = run_aviary("rocket", 0, 200, seed=500, start_pos=[0.0, 0.0, 80.0], start_orn=[0.05, 0.02, 0.3], noise=True)
    save("aviary_rocket_default_fuel", **d)
    # plenty of fuel, a burn long enough to move the composite centre of mass noticeably
    d = run_aviary("rocket", 0, 300, seed=501, start_pos=[1.0, -2.0, 200.0], start_orn=[-0.1, 0.15, -1.0], noise=True,
                   drone_options=dict(starting_fuel_ratio=0.6))
    save("aviary_rocket_fuel60", **d)
    # free fall with the engine off from low altitude, tilted: a leg / the booster reaches the floor
    d = run_aviary("rocket", 0, 150, seed=502, start_pos=[0.0, 0.0, 4.0], start_orn=[0.4, 0.1, 0.0], noise=False,
                   drone_options=dict(starting_fuel_ratio=0.0))
    save("aviary_rocket_drop", **d)


def gen_landing():
    # landings with the motors off: the contact RESPONSE (impulses, friction, penetration recovery) from first touch to rest
    d = run_aviary("quadx", -1, 250, seed=31, start_pos=[0.1, -0.1, 0.2], start_orn=[0.2, -0.1, 0.5], noise=False, hold_setpoint=[0, 0, 0, 0])
    save("aviary_quadx_land", **d)
    d = run_aviary("quadx", -1, 300, seed=32, start_pos=[0.0, 0.0, 0.4], start_orn=[0.6, 0.3, 0.0], noise=False,
                   drone_options=dict(drone_model="primitive_drone"), hold_setpoint=[0, 0, 0, 0])
    save("aviary_primitive_land", **d)
    # the Rocket settling on its three legs (rocket.urdf:208-277), engine off, tank empty
    d = run_aviary("rocket", 0, 450, seed=33, start_pos=[0.0, 0.0, 2.8], start_orn=[0.03, -0.02, 0.4], noise=False,
                   drone_options=dict(starting_fuel_ratio=0.0), hold_setpoint=[0, 0, 0, 0, 0, 0, 0])
    save("aviary_rocket_land", **d)


def gen_wind():
    # wind hook (aviary.py:266-285,324-333; boring_bodies.py:93-96; lifting_surfaces.py:88-93)
    d = run_aviary("quadx", 6, 150, seed=41, start_pos=[0.3, -0.2, 1.5], start_orn=[0.05, -0.08, 0.6], noise=True, wind="register")
    save("aviary_quadx_wind_register", **d)
    d = run_aviary("quadx", 0, 100, seed=42, start_pos=[0.0, 0.0, 2.0], start_orn=[0.0, 0.0, 0.0], noise=True, wind="ctor")
    save("aviary_quadx_wind_ctor", **d)
    d = run_aviary("fixedwing", 0, 150, seed=43, start_pos=[0.0, 0.0, 10.0], start_orn=[0.02, 0.05, -0.3], noise=True, wind="ctor")
    save("aviary_fixedwing_wind_ctor", **d)
    d = run_aviary("fixedwing", 0, 100, seed=44, start_pos=[0.0, 0.0, 10.0], start_orn=[0.0, 0.0, 0.0], noise=True, wind="register")
    save("aviary_fixedwing_wind_register", **d)
    d = run_aviary("rocket", 0, 120, seed=45, start_pos=[1.0, -2.0, 150.0], start_orn=[-0.1, 0.15, -1.0], noise=True, wind="ctor",
                   drone_options=dict(starting_fuel_ratio=0.3))
    save("aviary_rocket_wind_ctor", **d)


# --------------------------------------------------------------------------- env level
def flat_obs(env, obs, num_targets):
    if isinstance(obs, dict):
        t = np.zeros((num_targets, 4 if getattr(env.waypoints, "use_yaw_targets", False) else 3))
        n = obs["target_deltas"].shape[0]
        t[:n] = obs["target_deltas"]
        return np.concatenate([obs["attitude"], t.reshape(-1)])
    return np.array(obs, dtype=np.float64)


def run_env(make, n_steps, seed, action_fn, num_targets=0, ticks=6, contact_response=True):
    """contact_response: every env task runs what stepSimulation does (aviary.py:516), contact solve included -- fake_bullet's
    own default, left alone here. `False` records the explicit opt-out (`world_options=dict(contact_response=False)`:
    contact DETECTION only, bodies pass through the floor), which only the detect-only fixture uses."""
    from oracle import fake_bullet

    assert fake_bullet.BulletClient.DEFAULT_CONTACT_RESPONSE is True
    if contact_response:
        return _run_env(make, n_steps, seed, action_fn, num_targets, ticks)
    fake_bullet.BulletClient.DEFAULT_CONTACT_RESPONSE = False
    try:
        return _run_env(make, n_steps, seed, action_fn, num_targets, ticks)
    finally:
        fake_bullet.BulletClient.DEFAULT_CONTACT_RESPONSE = True


def _run_env(make, n_steps, seed, action_fn, num_targets=0, ticks=6):
    env = make()
    obs, info = env.reset(seed=seed)
    rng_env = env.np_random
    rec = dict(action=[], obs=[], reward=[], term=[], trunc=[], xi=[], reset_before=[], reset_obs=[], reset_xi=[],
               reset_u=[], info_oob=[], info_col=[], info_complete=[], info_ntr=[])

    def log_reset(o):
        rec["reset_obs"].append(flat_obs(env, o, num_targets))
        u = rng_env.drain("uniform")
        x = rng_env.drain("normal")
        rec["reset_u"].append(u if len(u) else np.zeros(3 * max(num_targets, 1)))
        rec["reset_xi"].append(x)

    log_reset(obs)
    rng = np.random.default_rng(seed + 77)
    need_reset = False
    ep_seed = seed
    for k in range(n_steps):
        if need_reset:
            ep_seed += 1
            obs, info = env.reset(seed=ep_seed)
            rng_env = env.np_random
            log_reset(obs)
            need_reset = False
            rec["reset_before"].append(k)
        a = action_fn(env, rng, k)
        if ref_stubs.RecordingRNG.F32:  # (float32-exact inputs: see RecordingRNG)
            a = np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)
        obs, r, te, tr, info = env.step(a)
        x = rng_env.drain("normal")
        rec["action"].append(a)
        rec["obs"].append(flat_obs(env, obs, num_targets))
        rec["reward"].append(r)
        rec["term"].append(te)
        rec["trunc"].append(tr)
        rec["xi"].append(np.concatenate([x, np.full(ticks - len(x), np.nan)]))
        rec["info_oob"].append(info["out_of_bounds"])
        rec["info_col"].append(info["collision"])
        rec["info_complete"].append(info["env_complete"])
        rec["info_ntr"].append(info.get("num_targets_reached", 0))
        need_reset = bool(te or tr)
    out = {k: np.array(v) for k, v in rec.items()}
    out["seed"] = seed
    return out


def uniform_action(env, rng, k):
    return rng.uniform(env.action_space.low, env.action_space.high)


def gentle_quad_action(env, rng, k):
    return np.array([*rng.uniform(-0.3, 0.3, size=3), rng.uniform(0.33, 0.40)])


def gentle_fw_action(env, rng, k):
    return np.array([*rng.uniform(-0.3, 0.3, size=3), rng.uniform(-0.2, 0.8)])


def lowthrust_quad_action(env, rng, k):
    return np.array([*rng.uniform(-0.5, 0.5, size=3), rng.uniform(0.0, 0.25)])


def gen_envs_crash():
    # (the terminal observations carry the impact impulses of stepSimulation's contact solve)
    save("env_hover_crash", **run_env(lambda: QuadXHoverEnv(), 150, 8, lowthrust_quad_action, ticks=6))
    # the same episodes under the explicit opt-out world_options=dict(contact_response=False): detection only
    save("env_hover_crash_detect_only", **run_env(lambda: QuadXHoverEnv(), 150, 8, lowthrust_quad_action, ticks=6, contact_response=False))
    # the fixedwing cannot reach the 30 m x 30 m floor box from its default start (z=10, 20 m/s), so
    # floor contact for it is pinned at Aviary level, from a low start
    d = run_aviary("fixedwing", 0, 60, seed=12, start_pos=[0.0, 0.0, 0.8], start_orn=[0.2, 0.25, 0.0], noise=True)
    save("aviary_fixedwing_drop", **d)


def gen_envs():
    save("env_hover_random", **run_env(lambda: QuadXHoverEnv(), 500, 0, uniform_action, ticks=6))
    save("env_hover_gentle_trunc", **run_env(lambda: QuadXHoverEnv(max_duration_seconds=0.5), 120, 1, gentle_quad_action, ticks=6))
    save("env_hover_euler_sparse", **run_env(lambda: QuadXHoverEnv(angle_representation="euler", sparse_reward=True), 150, 2, uniform_action, ticks=6))
    save("env_quadx_waypoints_random", **run_env(lambda: QuadXWaypointsEnv(), 400, 3, uniform_action, num_targets=4, ticks=8))
    save("env_quadx_waypoints_reach", **run_env(lambda: QuadXWaypointsEnv(goal_reach_distance=2.5), 300, 4, gentle_quad_action, num_targets=4, ticks=8))
    save("env_fixedwing_waypoints_random", **run_env(lambda: FixedwingWaypointsEnv(), 500, 5, uniform_action, num_targets=4, ticks=8))
    save("env_fixedwing_waypoints_gentle", **run_env(lambda: FixedwingWaypointsEnv(goal_reach_distance=40.0), 400, 6, gentle_fw_action, num_targets=4, ticks=8))


def gen_envs_options():
    """constructor options away from their defaults (quadx_base_env.py:30-147, quadx_waypoints_env.py:28-100, fixedwing_waypoints_env.py:28-100):
    agent_hz (env steps of 4 and of 2 Aviary steps instead of 3 / 4), flight_dome_size, max_duration_seconds, num_targets, sparse_reward,
    Euler-angle observations, goal_reach_distance -- each of them a branch or a loop bound in the env step"""
    save("env_hover_opts", **run_env(lambda: QuadXHoverEnv(agent_hz=30, flight_dome_size=2.0, max_duration_seconds=1.5), 200, 21, uniform_action, ticks=8))
    save("env_quadx_waypoints_opts", **run_env(lambda: QuadXWaypointsEnv(num_targets=2, sparse_reward=True, flight_dome_size=4.0, agent_hz=60, goal_reach_distance=1.5,
                                                                        angle_representation="euler", max_duration_seconds=4.0), 300, 22, gentle_quad_action, num_targets=2, ticks=4))
    save("env_fixedwing_waypoints_opts", **run_env(lambda: FixedwingWaypointsEnv(num_targets=3, sparse_reward=True, angle_representation="euler", flight_dome_size=60.0, agent_hz=40,
                                                                                goal_reach_distance=30.0, max_duration_seconds=20.0), 400, 23, gentle_fw_action, num_targets=3, ticks=6))


def mode_action(mode):
    """Random setpoints that exercise a flight mode's outer loops without leaving its sensible range (the action box itself,
    quadx_base_env.py:80-102, is the same [-pi, pi]^3 x [0, 0.8] for every mode but -1)."""
    def f(env, rng, k):
        if mode == -1:
            return rng.uniform(0.0, 0.8, size=4)  # motor commands
        if mode in (1, 3):   # angles (+ climb rate / height)
            return np.array([*rng.uniform(-0.5, 0.5, size=3), rng.uniform(0.0, 0.8) if mode == 1 else rng.uniform(0.5, 1.5)])
        if mode == 2:        # rates + height
            return np.array([*rng.uniform(-1.0, 1.0, size=3), rng.uniform(0.5, 1.5)])
        if mode in (4, 5, 6):  # velocities (+ yaw rate) + height / climb rate
            return np.array([*rng.uniform(-1.5, 1.5, size=2), rng.uniform(-1.0, 1.0), rng.uniform(0.5, 1.5) if mode == 4 else rng.uniform(-0.5, 0.8)])
        return np.array([*rng.uniform(-2.5, 2.5, size=2), rng.uniform(-1.0, 1.0), rng.uniform(0.3, 2.5)])  # 7: position, yaw, height
    return f


def gen_envs_modes():
    """QuadXHoverEnv / QuadXWaypointsEnv under every flight mode other than 0 (quadx.py:233-373,437-479): the env-level fixtures of
    the cascaded-PID kernels -- set_mode's default setpoint and the z PIDs inside the reset's settle steps included."""
    # (round 6: these fixtures' actions and draws are float32-exact -- ref_stubs.RecordingRNG.F32)
    ref_stubs.RecordingRNG.F32 = True
    for m in (-1, 1, 2, 3, 4, 5, 6, 7):
        tag = "m1" if m == -1 else str(m)
        # (1.5 s episodes: every fixture goes through several resets, i.e. through the settle steps under its mode's controller)
        save(f"env_hover_mode{tag}", **run_env(lambda: QuadXHoverEnv(flight_mode=m, max_duration_seconds=1.5), 260 if m in (6, 7, -1) else 140, 30 + m, mode_action(m), ticks=6))

    def chase(env, rng, k):  # position mode flown at the next waypoint: targets are reached
        t = env.waypoints.targets[0]
        return np.array([t[0], t[1], 0.0, t[2]]) + rng.uniform(-0.05, 0.05, size=4)

    save("env_quadx_waypoints_mode7", **run_env(lambda: QuadXWaypointsEnv(flight_mode=7, goal_reach_distance=0.4), 360, 47, chase, num_targets=4, ticks=8))
    ref_stubs.RecordingRNG.F32 = False


def gen_envs_yaw():
    # use_yaw_targets=True (quadx_waypoints_env.py:40, waypoint_handler.py:85-89,144-156,167-179): four more uniforms at
    # reset (after the position draws), (remaining, 4) target deltas, reach = distance AND yaw error under goal_reach_angle
    save("env_quadx_waypoints_yaw_random", **run_env(lambda: QuadXWaypointsEnv(use_yaw_targets=True), 300, 21, uniform_action, num_targets=4, ticks=8))
    # wide distance gate, moderate angle gate, gentle flight: targets are reached, and some distance-only passes are refused
    save("env_quadx_waypoints_yaw_reach", **run_env(lambda: QuadXWaypointsEnv(use_yaw_targets=True, goal_reach_distance=2.5, goal_reach_angle=1.2),
                                                    400, 22, gentle_quad_action, num_targets=4, ticks=8))


def gen_ma_hover():
    """pz_envs/quadx_envs/ma_quadx_hover_env.py driven through its PettingZoo dict API: 4 agents in one
    (fake-Bullet) world -- drone-drone contact is not restated, so agents are independent lanes. The
    shared np_random draws one motor-noise scalar per drone per tick, in drone order."""
    from PyFlyt.pz_envs.quadx_envs.ma_quadx_hover_env import MAQuadXHoverEnv

    made = []
    orig = np.random.default_rng

    def recording_default_rng(seed=None):
        r = ref_stubs.RecordingRNG(orig(seed))
        made.append(r)
        return r

    from oracle import fake_bullet

    np.random.default_rng = recording_default_rng
    try:
        env = MAQuadXHoverEnv(flight_dome_size=2.5, max_duration_seconds=1.0)  # small dome/duration: both exits occur
        rng = orig(123)
        rec = dict(action=[], obs=[], reward=[], term=[], trunc=[], xi=[], alive=[], reset_before=[], reset_obs=[], reset_xi=[])

        def do_reset(seed):
            obs, infos = env.reset(seed=seed)
            r = made[-1]
            rec["reset_obs"].append(np.stack([obs[a] for a in env.possible_agents]))
            rec["reset_xi"].append(r.drain("normal").reshape(-1, 4))  # [tick][drone]
            return r

        r = do_reset(1000)
        n_ag = len(env.possible_agents)
        for k in range(90):
            if len(env.agents) == 0:
                rec["reset_before"].append(k)
                r = do_reset(1000 + k)
            alive = [a in env.agents for a in env.possible_agents]
            acts = {a: np.array([*rng.uniform(-1.0, 1.0, size=3), rng.uniform(0.2, 0.7)]) for a in env.agents}
            obs, rew, term, trunc, infos = env.step(acts)
            A = np.zeros((n_ag, 4)); O = np.full((n_ag, 24), np.nan); R = np.full(n_ag, np.nan)
            T = np.zeros(n_ag, bool); U = np.zeros(n_ag, bool)
            for i, a in enumerate(env.possible_agents):
                if a in acts:
                    A[i] = acts[a]; O[i] = obs[a]; R[i] = rew[a]; T[i] = term[a]; U[i] = trunc[a]
            rec["action"].append(A); rec["obs"].append(O); rec["reward"].append(R); rec["term"].append(T); rec["trunc"].append(U)
            rec["alive"].append(alive)
            rec["xi"].append(r.drain("normal").reshape(-1, 4))
        save("env_ma_quadx_hover", start_pos=env.start_pos, start_orn=env.start_orn, dome=2.5, max_steps=env.max_steps,
             **{k: np.array(v) for k, v in rec.items()})
    finally:
        np.random.default_rng = orig
        fake_bullet.BulletClient.DEFAULT_CONTACT_RESPONSE = True


def _run_ma_shared(name, start_pos, base_actions, steps, dome, duration, rng_seed):
    """MAQuadXHoverEnv on ONE fake-Bullet world (every agent's drone in it, as in the reference), recorded through its PettingZoo
    dict API; `all_pos` / `all_rpy`: the pose of every drone after every step, culled ones included (they stay in the world
    with zero commands, ma_quadx_base_env.py:326-332)."""
    from oracle import fake_bullet
    from PyFlyt.pz_envs.quadx_envs.ma_quadx_hover_env import MAQuadXHoverEnv

    made = []
    orig = np.random.default_rng

    def recording_default_rng(seed=None):
        r = ref_stubs.RecordingRNG(orig(seed))
        made.append(r)
        return r

    np.random.default_rng = recording_default_rng
    assert fake_bullet.BulletClient.DEFAULT_CONTACT_RESPONSE is True
    try:
        env = MAQuadXHoverEnv(start_pos=start_pos, start_orn=np.zeros((len(start_pos), 3)), flight_dome_size=dome, max_duration_seconds=duration)
        rng = orig(rng_seed)
        rec = dict(action=[], obs=[], reward=[], term=[], trunc=[], xi=[], alive=[], reset_before=[], reset_obs=[], reset_xi=[],
                   world_contact=[], drone_contact=[], all_pos=[], all_rpy=[])

        def do_reset(seed):
            obs, infos = env.reset(seed=seed)
            r = made[-1]
            rec["reset_obs"].append(np.stack([obs[a] for a in env.possible_agents]))
            rec["reset_xi"].append(r.drain("normal").reshape(-1, len(start_pos)))
            return r

        r = do_reset(1000)
        n_ag = len(env.possible_agents)
        for k in range(steps):
            if len(env.agents) == 0:
                rec["reset_before"].append(k)
                r = do_reset(1000 + k)
            alive = [a in env.agents for a in env.possible_agents]
            acts = {}
            for a in env.agents:
                i = env.agent_name_mapping[a]
                acts[a] = np.array(base_actions[i]) + np.array([*rng.uniform(-0.05, 0.05, size=3), rng.uniform(-0.01, 0.01)])
            obs, rew, term, trunc, infos = env.step(acts)
            A = np.zeros((n_ag, 4)); O = np.full((n_ag, 24), np.nan); R = np.full(n_ag, np.nan)
            T = np.zeros(n_ag, bool); U = np.zeros(n_ag, bool)
            for i, a in enumerate(env.possible_agents):
                if a in acts:
                    A[i] = acts[a]; O[i] = obs[a]; R[i] = rew[a]; T[i] = term[a]; U[i] = trunc[a]
            rec["action"].append(A); rec["obs"].append(O); rec["reward"].append(R); rec["term"].append(T); rec["trunc"].append(U)
            rec["alive"].append(alive)
            rec["xi"].append(r.drain("normal").reshape(-1, n_ag))
            ca = env.aviary.contact_array
            ids = [d.Id for d in env.aviary.drones]
            rec["world_contact"].append(bool(np.any(ca)))
            rec["drone_contact"].append([bool(np.any(ca[i][ids])) for i in ids])
            st = [env.aviary.state(i) for i in range(n_ag)]
            rec["all_pos"].append([s_[3] for s_ in st]); rec["all_rpy"].append([s_[1] for s_ in st])
        save(name, start_pos=env.start_pos, start_orn=env.start_orn, dome=dome, max_steps=env.max_steps,
             **{k: np.array(v) for k, v in rec.items()})
        return rec
    finally:
        np.random.default_rng = orig


def gen_ma_hover_shared():
    """The PettingZoo env with what makes its world SHARED visible: agents spawned 10 cm apart in height so that the pairs
    collide while they drift (drone-drone hits enter contact_array[drone.Id], ma_quadx_hover_env.py:181) and push each other
    (the contact response between the drones), a dead drone comes to rest on the floor -- and from then on switches off the
    rotational drag of every drone in the world (quadx.py:509 looks at the contact points of the whole world)."""
    # agents 0 and 1 steer towards each other (roll-rate commands of opposite sign), the others hover / sink
    _run_ma_shared("env_ma_quadx_hover_shared", np.array([[-0.1, 0.0, 1.0], [0.1, 0.0, 1.01], [0.0, 1.0, 1.0], [0.0, -1.0, 0.5]]),
                   {0: [0.0, 0.6, 0.0, 0.36], 1: [0.0, -0.6, 0.0, 0.36], 2: [0.0, 0.0, 0.3, 0.37], 3: [0.0, 0.0, 0.0, 0.05]}, 140, 3.0, 2.0, 321)


def gen_ma_hover_stack():
    """A culled drone lands on a live one (ma_quadx_base_env.py:365-369): agent 1 spawns outside the flight dome, directly
    above agent 0 -- out of bounds in its first step, culled, zero commands from then on -- and falls onto agent 0, which
    hovers; the hit ends agent 0's episode too, and the two come down together, one on top of the other, while agents 2 and 3
    fly on (their rotational drag switches off with the first contact point in the world)."""
    # (a small dome: agent 1 is out of it 25 cm above agent 0 and meets it at about 2 m/s -- under a centimetre a tick against a
    #  collision box 2 cm thick. Contact points exist from touching on (contact_margin 0), so a drop from a metre up, 2 cm a tick,
    #  would put a vertex right through the other box: which face it leaves by is then decided by the last digits)
    rec = _run_ma_shared("env_ma_quadx_hover_stack", np.array([[0.0, 0.0, 1.0], [0.02, 0.01, 1.25], [0.6, 0.0, 0.9], [-0.5, -0.5, 0.8]]),
                         {0: [0.0, 0.0, 0.0, 0.364], 1: [0.0, 0.0, 0.0, 0.364], 2: [0.0, 0.0, 0.2, 0.366], 3: [0.0, 0.0, -0.2, 0.365]}, 100, 1.2, 3.0, 654)
    t = np.array(rec["term"]); dc = np.array(rec["drone_contact"])
    print("stack: agent 1 out at step", int(np.argmax(t[:, 1])), "agent 0 hit at step", int(np.argmax(t[:, 0])), "drone-drone steps", int(dc.any(axis=1).sum()),
          "final z", np.array(rec["all_pos"])[-1][:, 2])


def gen_dogfight():
    """MAFixedwingDogfightEnv (pz_envs/fixedwing_envs/ma_fixedwing_dogfight_env.py, ma_fixedwing_base_env.py): two teams of
    Acrowing aircraft in ONE world (world_scale 5), spawned on a circle flying outwards at 20 m/s, observing each other in
    their own body frames, scoring hits inside a cone of fire. Two recordings of the reference's env on fake_bullet:
      * `env_dogfight_default`: the default parameters, random actions (no one gets a shot in: observation layout, the
        closing / boundary rewards, the accumulate-then-pop reward protocol, truncation);
      * `env_dogfight_engage`: a wide cone of fire, long range and heavy damage with spawn poses that put every aircraft in
        an opponent's sights (hits, health, deaths, the element-wise team-win override, culling of finished agents);
      * `env_dogfight_crash`: a small flight dome and one aircraft diving from low altitude: the collision and out-of-bounds
        overrides (-1000, health 0), aircraft that keep flying with zero commands after they were culled, and the `inactive`
        filter of the observation (a dead aircraft at rest on the ground disappears from the others' observations).
    Start poses of the last two are injected through `_get_start_pos_orn`."""
    from oracle import fake_bullet
    from PyFlyt.pz_envs.fixedwing_envs.ma_fixedwing_dogfight_env import MAFixedwingDogfightEnv

    made = []
    orig = np.random.default_rng

    def recording_default_rng(seed=None):
        r = ref_stubs.RecordingRNG(orig(seed))
        made.append(r)
        return r

    np.random.default_rng = recording_default_rng
    fake_bullet.BulletClient.DEFAULT_CONTACT_RESPONSE = True
    try:
        def run(name, n_steps, policy, seed, spawn=None, **kw):
            env = MAFixedwingDogfightEnv(**kw)
            if spawn is not None:
                env._get_start_pos_orn = lambda seed_: (spawn[0].copy(), spawn[1].copy())
            A = env.num_possible_agents
            D = env.observation_space(None).shape[0]
            n0 = len(made)
            obs, infos = env.reset(seed=seed)
            rngs = made[n0:]  # the Aviary's Generator, shared by its drones (aviary.py:258-262)
            rec = dict(action=[], obs=[], reward=[], term=[], trunc=[], alive=[], health=[], received_hits=[], xi=[], info_bits=[])

            def drain():
                # motor noise: one N(0,1) per aircraft per physics tick (motors.py:188, one motor), drawn drone by drone
                # inside each tick (aviary.py:505-508) -> [ticks, A]
                cols = [c for c in (r.drain("normal") for r in rngs) if c.size]
                assert len(cols) == 1, [c.size for c in cols]
                return cols[0].reshape(-1, A)

            reset_xi = drain()
            reset_obs = np.stack([obs[a] for a in env.possible_agents])
            prng = orig(seed + 1000)
            for k in range(n_steps):
                if len(env.agents) == 0:
                    break
                alive = np.array([a in env.agents for a in env.possible_agents])
                acts = {a: policy(k, env.agent_name_mapping[a], prng) for a in env.agents}
                obs, rew, term, trunc, infos = env.step(acts)
                Aa = np.zeros((A, env.action_space(None).shape[0])); Oo = np.full((A, D), np.nan); R = np.full(A, np.nan)
                T = np.zeros(A, bool); U = np.zeros(A, bool); B = np.zeros(A, np.int32)
                for i, a in enumerate(env.possible_agents):
                    if a in acts:
                        Aa[i] = acts[a]; Oo[i] = obs[a]; R[i] = rew[a]; T[i] = term[a]; U[i] = trunc[a]
                        inf = infos[a]
                        B[i] = (1 * bool(inf.get("dead"))) | (2 * bool(inf.get("collision"))) | (4 * bool(inf.get("out_of_bounds"))) | (8 * bool(inf.get("team_win")))
                rec["action"].append(Aa); rec["obs"].append(Oo); rec["reward"].append(R); rec["term"].append(T); rec["trunc"].append(U)
                rec["alive"].append(alive); rec["health"].append(env.healths.copy()); rec["received_hits"].append(env.received_hits.copy())
                rec["info_bits"].append(B)
                rec["xi"].append(drain())
            save(name, start_pos=env.start_pos, start_orn=env.start_orn, reset_obs=reset_obs, reset_xi=reset_xi,
                 team_size=env.team_size, max_steps=env.max_steps, env_step_ratio=env.env_step_ratio, dome=env.flight_dome_size,
                 damage_per_hit=env.damage_per_hit, lethal_distance=env.lethal_distance, lethal_angle=env.lethal_angle,
                 aggressiveness=env.aggressiveness, cooperativeness=env.cooperativeness, sparse_reward=env.sparse_reward,
                 action_dim=env.action_space(None).shape[0],
                 **{k: np.array(v) for k, v in rec.items()})
            return rec

        run("env_dogfight_default", 80, lambda k, i, g: g.uniform(-1.0, 1.0, size=4), seed=3, max_duration_seconds=2.5)
        # the other team sizes of the reference's own test matrix (tests/test_pz_envs.py: 1, 2, 3) and the sparse reward
        run("env_dogfight_team1_sparse", 40, lambda k, i, g: g.uniform(-0.3, 0.3, size=4) + np.array([0, 0, 0, 0.4]), seed=11, team_size=1,
            sparse_reward=True, max_duration_seconds=1.0, lethal_distance=80.0, lethal_angle_radians=0.8)
        # assisted_flight=False: six-wide actions; the Aviary stays in mode 0 and reads the first four, the thrust remap lands on the sixth
        run("env_dogfight_unassisted", 40, lambda k, i, g: g.uniform(-0.3, 0.3, size=6) + np.array([0, 0, 0, 0.6, 0.2, -0.4]), seed=13, assisted_flight=False,
            max_duration_seconds=1.0, lethal_distance=80.0, lethal_angle_radians=0.8)
        run("env_dogfight_team3", 40, lambda k, i, g: g.uniform(-0.3, 0.3, size=4) + np.array([0, 0, 0, 0.4]), seed=12, team_size=3,
            max_duration_seconds=1.0, lethal_distance=80.0, lethal_angle_radians=0.8, damage_per_hit=0.01)

        # four aircraft 40 m up: 0 chases 2 from behind, 3 chases 1; a gentle random wobble on top of level flight
        pos = np.array([[0.0, 0.0, 40.0], [60.0, 35.0, 42.0], [25.0, 1.0, 40.5], [35.0, 34.0, 41.5]])
        orn = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.02], [0.0, 0.0, 0.03]])

        def chase(k, i, g):
            a = np.array([0.0, 0.0, 0.0, 0.6]) + g.uniform(-0.05, 0.05, size=4)
            if i == 1 and k > 25:
                a[1] = -1.0  # full pitch down: into the ground
            return a

        rec = run("env_dogfight_engage", 200, chase, seed=5, spawn=(pos, orn), damage_per_hit=0.02, lethal_distance=40.0,
                  lethal_angle_radians=0.25, max_duration_seconds=20.0)
        print("engage: hits", np.array(rec["received_hits"])[-1], "health", np.array(rec["health"])[-1], "bits", np.bitwise_or.reduce(np.array(rec["info_bits"]), axis=0))

        pos = np.array([[10.0, 0.0, 30.0], [0.0, 20.0, 3.0], [-30.0, 0.0, 35.0], [0.0, -370.0, 30.0]])
        orn = np.array([[0.0, 0.0, 0.0], [0.0, 0.6, np.pi / 2], [0.0, 0.0, 3.0], [0.0, 0.0, -np.pi / 2]])  # (yaw 3.0, not pi: the Euler branch cut)

        def crash(k, i, g):
            a = np.array([0.0, 0.02, 0.0, 0.5]) + g.uniform(-0.1, 0.1, size=4)
            if i == 1:
                a[1], a[3] = -1.0, -1.0  # nose down, throttle closed
            return a

        rec = run("env_dogfight_crash", 260, crash, seed=7, spawn=(pos, orn), flight_dome_size=400.0, max_duration_seconds=8.0)
        print("crash: steps", len(rec["action"]), "health", np.array(rec["health"])[-1], "bits", np.bitwise_or.reduce(np.array(rec["info_bits"]), axis=0),
              "others rows seen", sorted(set(int((~np.isnan(o) & (o != 0)).sum()) for o in np.array(rec["obs"]).reshape(-1, 65))))
        # a mid-air collision (ma_fixedwing_dogfight_env.py:672-676: both aircraft are out, -1000, health 0 -- and fly on as wrecks
        # that the other two keep observing): 0 and 2 head-on at the same height, 30 cm apart sideways, the other two far away.
        # Recorded twice: with the contact response between the aircraft (what stepSimulation does) and, as a control, without it
        # (`obs_nopair`): where the two recordings part is what the pair stage is worth.
        pos = np.array([[-20.0, 0.0, 50.0], [0.0, 120.0, 60.0], [19.8, -2.52, 50.0], [0.0, -120.0, 60.0]])
        orn = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 1.5], [0.0, 0.0, 3.0], [0.0, 0.0, -1.5]])

        def level(k, i, g):
            return np.array([0.0, 0.0, 0.0, 0.6]) + (g.uniform(-0.03, 0.03, size=4) if i in (1, 3) else 0.0)

        rec = run("env_dogfight_midair", 60, level, seed=9, spawn=(pos, orn), flight_dome_size=400.0, max_duration_seconds=6.0)
        term = np.array(rec["term"])
        print("midair: collision steps", [int(np.argmax(term[:, i])) if term[:, i].any() else None for i in range(4)], "bits", np.bitwise_or.reduce(np.array(rec["info_bits"]), axis=0))
        fake_bullet.BulletClient.DEFAULT_PAIR_RESPONSE = False
        try:
            rec0 = run("env_dogfight_midair_nopair", 60, level, seed=9, spawn=(pos, orn), flight_dome_size=400.0, max_duration_seconds=6.0)
        finally:
            fake_bullet.BulletClient.DEFAULT_PAIR_RESPONSE = True
        d = np.nan_to_num(np.array(rec["obs"])[:, [1, 3]]) - np.nan_to_num(np.array(rec0["obs"])[:, [1, 3]])
        print("midair: the survivors' observations with / without the pair stage part by", float(np.abs(d).max()))
    finally:
        np.random.default_rng = orig
        fake_bullet.BulletClient.DEFAULT_CONTACT_RESPONSE = True


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--out":
        OUT_DIR = sys.argv[2]
        del sys.argv[1:3]
    if len(sys.argv) > 1:  # regenerate selected groups only: python gen_goldens.py wind
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_pid()
    gen_aero_and_motors()
    gen_aviary()
    gen_envs()
    gen_envs_crash()
    gen_envs_options()
    gen_envs_yaw()
    gen_envs_modes()
    gen_landing()
    gen_ma_hover()
    gen_ma_hover_shared()
    gen_ma_hover_stack()
    gen_dogfight()
    gen_wind()
    gen_control_rate()
    gen_primitive()
    gen_rocket()
    gen_acrowing()
