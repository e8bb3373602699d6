"""Analytic known-answer tests of the DEVICE code's Bullet restatement (rows 10-12 of SURVEY.md 8(a)) through
the C ABI: the same closed forms (tests/kat.py) the fp64 oracle is held to in tests/test_oracle_kat.py.
pf_body_tick = applyExternalForce / applyExternalTorque + stepSimulation on the generic Body::tick;
the specialised hot kernel's own integrator (QuadHot::tick) is reached through pf_env_step with the motors and
drag parameters zeroed. fp32 tolerances are stated per test."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

import kat  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def body_engine(n, z=10.0, world=None, quat=None):
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    P = build_params("quadx", "none", noise="off", autoreset="off", world_options=world)
    eng = BatchEngine(P, n, device=DEV)
    pose = torch.zeros(n, 7, device=DEV)
    pose[:, 2] = z
    pose[:, 6] = 1.0
    if quat is not None:
        pose[:, 3:7] = torch.as_tensor(quat, dtype=torch.float32, device=DEV)
    eng.aviary_reset(pose.contiguous())
    return eng


def set_w(eng, w):
    """world-frame angular velocity -> state groups g2.w, g3.xy."""
    w = torch.as_tensor(w, dtype=torch.float32, device=DEV)
    eng.state[2, :, 3] = w[:, 0]
    eng.state[3, :, 0] = w[:, 1]
    eng.state[3, :, 1] = w[:, 2]


def base(eng):
    s = eng.state
    p, q = s[0, :, :3].double().cpu().numpy(), s[1].double().cpu().numpy()
    v = s[2, :, :3].double().cpu().numpy()
    w = torch.stack([s[2, :, 3], s[3, :, 0], s[3, :, 1]], dim=1).double().cpu().numpy()
    return p, q, v, w


def test_free_fall_closed_form():
    n = 64
    eng = body_engine(n, z=50.0)
    wr = torch.zeros(n, 6, device=DEV)
    for k in range(1, 41):
        eng.body_tick(wr, 10)
        p, q, v, w = base(eng)
        z, vz = kat.free_fall_z(50.0, 10 * k)
        assert np.abs(p[:, 2] - z).max() < 1e-4 * 50.0 and np.abs(v[:, 2] - vz).max() < 1e-5 * max(1.0, abs(vz)), k
    assert not p[:, :2].any() and not w.any() and (q == np.array([0, 0, 0, 1.0])).all()
    # the disarmed drone of Aviary.set_armed runs the same tick (aviary.py:423-438): gravity only
    from pyflyt_amd.core import Aviary

    env = Aviary(np.array([[0.0, 0.0, 50.0]] * 4), np.zeros((4, 3)), "quadx", motor_noise=False)
    env.set_armed(False)
    env.step(n_steps=100)
    z, _ = kat.free_fall_z(50.0, 200)
    assert abs(float(env.engine.state[0, 0, 2]) - z) < 5e-3
    env.disconnect()


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_constant_principal_axis_torque(axis):
    n = 64
    eng = body_engine(n, world=dict(gravity_z=0.0))
    wr = torch.zeros(n, 6, device=DEV)
    wr[:, 3 + axis] = 2.0e-5
    for k in range(1, 11):
        eng.body_tick(wr, 10)
        p, q, v, w = base(eng)
        wn, th = kat.const_torque_principal(2.0e-5, axis, 10 * k)
        assert np.abs(w[:, axis] - wn).max() < 1e-5 * max(1.0, abs(wn))
        assert np.abs(np.delete(w, axis, axis=1)).max() < 1e-6
        assert np.abs(q[:, axis] - np.sin(th / 2)).max() < 2e-5 and np.abs(q[:, 3] - np.cos(th / 2)).max() < 2e-5
        assert np.abs(np.linalg.norm(q, axis=1) - 1.0).max() < 2e-7
    assert np.abs(p - np.array([0, 0, 10.0])).max() == 0.0


@pytest.mark.parametrize("gyro", [True, False])
def test_torque_free_principal_spin_is_stationary(gyro):
    n = 64
    for axis in range(3):
        eng = body_engine(n, world=dict(gravity_z=0.0, use_gyro_term=gyro))
        w0 = np.zeros((n, 3), dtype=np.float32)
        w0[:, axis] = 7.0
        set_w(eng, w0)
        eng.body_tick(torch.zeros(n, 6, device=DEV), 500)
        p, q, v, w = base(eng)
        assert np.abs(w - w0).max() < 1e-5
        th = 7.0 * kat.DT * 500
        assert np.abs(np.abs(q[:, 3]) - abs(np.cos(th / 2))).max() < 1e-4
        assert np.abs(np.linalg.norm(q, axis=1) - 1.0).max() < 2e-7


def test_gyroscopic_term_switch():
    """Off the principal axes: with use_gyro_term the body axis precesses while R I w_b stays put; without it
    w_world is exactly constant."""
    n = 64
    w0 = np.tile(np.array([[3.0, -2.0, 5.0]], dtype=np.float32), (n, 1))
    eng = body_engine(n, world=dict(gravity_z=0.0, use_gyro_term=True))
    set_w(eng, w0)
    eng.body_tick(torch.zeros(n, 6, device=DEV), 240)
    p, q, v, w = base(eng)
    x, y, z, s = q[0]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * s), 2 * (x * z + y * s)],
                  [2 * (x * y + z * s), 1 - 2 * (x * x + z * z), 2 * (y * z - x * s)],
                  [2 * (x * z - y * s), 2 * (y * z + x * s), 1 - 2 * (x * x + y * y)]])
    L0 = kat.I_DIAG * w0[0]
    L1 = R @ (kat.I_DIAG * (R.T @ w[0]))
    assert np.linalg.norm(L1 - L0) / np.linalg.norm(L0) < 2e-2 and np.abs(w[0] - w0[0]).max() > 0.1
    eng = body_engine(n, world=dict(gravity_z=0.0, use_gyro_term=False))
    set_w(eng, w0)
    eng.body_tick(torch.zeros(n, 6, device=DEV), 240)
    assert np.array_equal(base(eng)[3], w0.astype(np.float64))


def test_velocity_clamp():
    n = 64
    eng = body_engine(n, world=dict(gravity_z=0.0))
    wr = torch.tensor([[1e3, -1e3, 2e3, 1.0, -1.0, 0.5]], device=DEV).repeat(n, 1).contiguous()
    eng.body_tick(wr, 5)
    p0, q, v, w = base(eng)
    assert (np.abs(v) == kat.VMAX).all() and (np.abs(w) == kat.VMAX).all()
    eng.body_tick(wr, 1)
    p1, q, v, w = base(eng)
    np.testing.assert_allclose(p1 - p0, kat.DT * v, atol=2e-6)
    assert np.abs(np.linalg.norm(q, axis=1) - 1.0).max() < 2e-7


def test_euler_quat_round_trip_and_gimbal_branch():
    """getQuaternionFromEuler on the host (params.quat_from_euler), getEulerFromQuaternion on the device
    (Aviary.state row 1 after reset)."""
    from pyflyt_amd.core import Aviary

    rng = np.random.default_rng(0)
    n = 256
    rpy = rng.uniform([-3.1, -1.5, -3.1], [3.1, 1.5, 3.1], size=(n, 3))
    sign = np.where(np.arange(64) % 2 == 0, 1.0, -1.0)
    gim = np.stack([rng.uniform(-1.5, 1.5, 64), sign * np.pi / 2, rng.uniform(-1.5, 1.5, 64)], axis=1)
    allr = np.concatenate([rpy, gim])
    pos = np.tile(np.array([[0.0, 0.0, 5.0]]), (len(allr), 1))
    env = Aviary(pos, allr, "quadx", motor_noise=False)
    got = env.all_states[:, 1].double().cpu().numpy()
    np.testing.assert_allclose(got[:n], rpy, atol=5e-6)
    g = got[n:]
    assert (g[:, 0] == 0.0).all() and np.abs(g[:, 1] - sign * np.pi / 2).max() < 1e-6
    want = kat.gimbal_yaw(gim[:, 0], gim[:, 2], sign)
    d = (g[:, 2] - want + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() < 2e-3  # fp32 quaternion of a pole attitude: sqrt(eps)-conditioned
    env.disconnect()


def test_denormalised_spawn_quaternion():
    """btMatrix3x3::setRotation divides by |q|^2. The generic kernels keep that factor, so a scaled spawn
    quaternion (only reachable through raw per-lane poses; the reference always passes unit quaternions from
    getQuaternionFromEuler) gives the same body-frame velocities; the specialised hot kernel, which assumes a
    unit quaternion in derive(), refuses such a parameter block and leaves it to the generic one."""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    n = 64
    rng = np.random.default_rng(2)
    qs = rng.normal(size=(n, 4))
    qu = qs / np.linalg.norm(qs, axis=1, keepdims=True)
    outs = []
    for scale in (1.0, 1.37):
        P = build_params("quadx", "none", noise="off", autoreset="off")
        eng = BatchEngine(P, n, device=DEV)
        eng.start_vel = torch.tensor(np.tile([[1.0, -2.0, 3.0]], (n, 1)), dtype=torch.float32, device=DEV)
        pose = np.concatenate([np.tile([[0.0, 0.0, 5.0]], (n, 1)), qu * scale], axis=1)
        eng.aviary_reset(torch.tensor(pose, dtype=torch.float32, device=DEV).contiguous())
        outs.append(eng.out_state.clone())
    assert torch.allclose(outs[0], outs[1], atol=3e-6)
    # the hot kernel's envelope excludes a non-unit spawn quaternion
    Ph = build_params("quadx", "hover")
    Ph.start_quat[2], Ph.start_quat[3] = 0.3, 1.2
    e = BatchEngine(Ph, 64, device=DEV)
    obs = e.env_reset().clone()
    Pu = build_params("quadx", "hover", start_orn=(0.0, 0.0, 2.0 * np.arctan2(0.3, 1.2)))
    e2 = BatchEngine(Pu, 64, device=DEV)
    obs2 = e2.env_reset()
    assert torch.allclose(obs, obs2, atol=2e-6)


def test_motor_lag_closed_form():
    from pyflyt_amd.core import Aviary

    env = Aviary(np.array([[0.0, 0.0, 5.0]] * 64), np.zeros((64, 3)), "quadx", motor_noise=False)
    env.set_mode(-1)
    pwm = np.array([0.3, 0.5, 0.7, 0.9])
    env.set_all_setpoints(np.tile(pwm, (64, 1)))
    for s in range(1, 21):
        env.step()
        thr = env.all_aux_states.double().cpu().numpy()
        assert np.abs(thr - kat.motor_lag(pwm, 2 * s)).max() < 3e-7 * 20
    env.disconnect()


def test_hover_equilibrium_throttle():
    from pyflyt_amd.core import Aviary

    env = Aviary(np.array([[0.0, 0.0, 5.0]] * 64), np.zeros((64, 3)), "quadx", motor_noise=False)
    env.set_mode(-1)
    env.set_all_setpoints(np.full((64, 4), np.sqrt(kat.HOVER_THROTTLE_SQ)))
    env.step(n_steps=40)
    vz0 = env.engine.state[2, :, 2].double().cpu().numpy().copy()
    env.step(n_steps=60)
    vz1 = env.engine.state[2, :, 2].double().cpu().numpy()
    drag_acc = 7.35e-4 * vz0 * vz0 / kat.MASS
    # thrust balances gravity to fp32 rounding of 4 * 0.5 N * thr^2 / m vs 9.81 (~1e-6 m/s^2 -> 5e-7 m/s over 120 ticks)
    assert np.abs((vz1 - vz0) - drag_acc * 120 * kat.DT).max() < 2e-5
    s = env.engine.state
    # (the four equal thrusts cancel every torque exactly in exact arithmetic; in fp32 the arm sum r_y f rounds to ~1e-7 relative)
    assert float(s[2, :, 3].abs().max() + s[3, :, :2].abs().max()) < 1e-5 and float(s[0, :, :2].abs().max()) < 1e-5
    env.disconnect()


def test_hot_kernel_integrator_free_fall_and_clamp():
    """QuadHot::tick (the specialised env kernel's own copy of the integrator) with the motors, drag and rate
    PID taken out of the parameter block: an env step is then 6 ticks of pure free fall, and the closed form
    must come out of the OBSERVATION the env returns. A huge gravity exercises the +-100 clamp."""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    def engine(g):
        P = build_params("quadx", "hover", noise="off", autoreset="off", flight_dome_size=1e4, start_pos=(0.0, 0.0, 2000.0),
                         world_options=dict(gravity_z=g))
        for i in range(4):
            P.motor_fmax[i] = 0.0
            P.motor_tmax[i] = 0.0
        for k in range(3):
            P.drag_const[k] = 0.0
        P.drag_coef_pqr = 0.0
        e = BatchEngine(P, 64, device=DEV)
        assert e.lib.pf_ctx_is_specialised(e._ctx) == 1
        return e

    e = engine(-kat.G)
    e.env_reset()
    a = torch.zeros(64, 4, device=DEV)
    for k in range(1, 51):
        obs, *_ = e.env_step(a)
    z, vz = kat.free_fall_z(2000.0, 20 + 6 * 50)  # 20 settle ticks + 50 env steps x 6 ticks
    o = obs.double().cpu().numpy()
    assert np.abs(o[:, 12] - z).max() < 1e-4 * 2000.0 and np.abs(o[:, 9] - vz).max() < 1e-5 * abs(vz)
    assert (o[:, 3:7] == np.array([0, 0, 0, 1.0])).all() and not o[:, :3].any()
    e = engine(-2000.0)
    e.env_reset()
    for k in range(20):
        obs, *_ = e.env_step(a)
    assert (obs[:, 9] == -kat.VMAX).all()  # body-frame vz of a level drone == world vz, clamped


# ------------------------------------------------------------------ contact response (the same known answers as the oracle's)
def test_contact_drop_comes_to_rest_at_half_height():
    """tests/test_oracle_kat.py::test_contact_drop_comes_to_rest_at_half_height on the device: level and tilted quads
    dropped with the motors off end ON the floor, z = half the collision box's height (0.01) minus the allowed overlap
    (contact_slop, 1e-5), at rest, never deeper than one tick of travel into the slab (contact points exist from touching on:
    contact_margin 0), with the contact reported on every step once they rest."""
    from pyflyt_amd.core import Aviary

    n = 128
    rng = np.random.default_rng(3)
    pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(0.15, 0.4, size=(n, 1))], axis=1)
    orn = np.concatenate([rng.uniform(-0.6, 0.6, size=(n, 2)), rng.uniform(-3, 3, size=(n, 1))], axis=1)
    orn[:16] = 0.0  # level drops
    env = Aviary(pos, orn, "quadx", motor_noise=False)
    env.set_mode(-1)
    env.set_all_setpoints(np.zeros((n, 4)))
    zmin = np.full(n, np.inf)
    touched = np.zeros(n, dtype=bool)
    for k in range(360):
        env.step()
        z = env.engine.state[0, :, 2].cpu().numpy()
        touched |= env.contact_array.cpu().numpy()
        zmin = np.minimum(zmin, z)
    st = env.all_states.double().cpu().numpy()
    assert touched.all()
    rest_z = 0.01 - kat.CONTACT_SLOP
    assert np.abs(st[:, 3, 2] - rest_z).max() < 2e-5, np.abs(st[:, 3, 2] - rest_z).max()  # rests at half-height minus the slop
    # no tunnelling: the box centre never sinks below the floor (impact speeds here stay under 3 m/s = 12 mm a tick, and a
    # tilted box's centre is that much higher anyway)
    assert zmin.min() > -0.003, zmin.min()
    assert env.contact_array.all()                                                         # and the resting contact is reported
    assert np.abs(st[:, 1, :2]).max() < 1e-3 and np.abs(st[:, 0]).max() < 1e-2 and np.abs(st[:, 2]).max() < 1e-3  # flat, at rest
    # the level drops did not move sideways (beyond what the sweeps' residual bound, 3.2e-4 m/s, leaves behind at the impact)
    assert np.abs(st[:16, 3, :2] - pos[:16, :2]).max() < 3e-4
    env.disconnect()


def test_rocket_settles_on_its_legs():
    from pyflyt_amd.core import Aviary

    n = 64
    rng = np.random.default_rng(4)
    # (gentle: dropped half a metre with a 3 degree tilt the 9.5 m body topples -- in the oracle as well)
    pos = np.concatenate([rng.uniform(-2, 2, size=(n, 2)), rng.uniform(2.45, 2.55, size=(n, 1))], axis=1)
    orn = np.concatenate([rng.uniform(-0.01, 0.01, size=(n, 2)), rng.uniform(-3, 3, size=(n, 1))], axis=1)
    env = Aviary(pos, orn, "rocket", motor_noise=False, drone_options=dict(starting_fuel_ratio=0.0))
    env.set_mode(0)
    env.step(n_steps=1500)
    st = env.all_states.double().cpu().numpy()
    assert np.abs(st[:, 3, 2] - (2.425 - kat.CONTACT_SLOP)).max() < 1e-3 and np.abs(st[:, 1, :2]).max() < 2e-3  # standing on the legs (rocket.urdf:208-277), upright
    assert np.abs(st[:, 2]).max() < 2e-2 and np.abs(st[:, 0]).max() < 2e-2
    env.disconnect()


def test_contact_friction_stops_a_slide():
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    n = 64
    out = {}
    for mu in (0.5, 0.0):
        P = build_params("quadx", "none", noise="off", autoreset="off", world_options=dict(contact_friction=mu))
        eng = BatchEngine(P, n, device=DEV)
        eng.start_vel = torch.tensor(np.tile([[1.0, 0.0, 0.0]], (n, 1)), dtype=torch.float32, device=DEV)
        pose = torch.zeros(n, 7, device=DEV)
        pose[:, 2] = 0.01 - kat.CONTACT_SLOP
        pose[:, 6] = 1.0
        eng.aviary_reset(pose.contiguous())
        sp = torch.zeros(n, 4, device=DEV)
        eng.aviary_set_mode(-1, sp)
        sp.zero_()
        eng.aviary_step(sp, 120)
        out[mu] = (eng.state[0, :, 0].double().cpu().numpy(), eng.state[2, :, 0].double().cpu().numpy())
    x, vx = out[0.5]
    assert np.abs(x - 1.0 / (2 * 0.5 * kat.G)).max() < 0.1 * 0.102 and np.abs(vx).max() < 1e-3  # stops after v^2 / (2 mu g)
    assert np.abs(out[0.0][1] - (1.0 - 7.35e-4 / kat.MASS)).max() < 2e-3  # frictionless: only the body drag slows it


def test_dogfight_known_answers_gpu():
    """The dogfight bookkeeping's closed-form answers (tests/kat.py: dogfight_tail_chase_expectations, see the oracle's twin in
    test_oracle_kat.py) on the device: 1 v 1 tail chase, one hit per update, health, the kill and the team win in the same update."""
    import numpy as np
    import torch
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine
    from kat import dogfight_tail_chase_expectations

    dmg = 0.05
    P = build_params("fixedwing", "dogfight", noise="off", autoreset="off", angle_representation="euler", sparse_reward=True, max_duration_seconds=10.0,
                     vehicle_options=dict(drone_model="acrowing"), world_options=dict(world_scale=5.0),
                     dogfight=dict(team_size=1, sample_spawn=False, damage_per_hit=dmg, lethal_distance=40.0, lethal_angle=0.2))
    E = 32
    eng = BatchEngine(P, 2 * E, device="cuda:0")
    sp = torch.zeros(2, 2 * E, 4, device="cuda:0")
    sp[0, 0::2, :3] = torch.tensor([0.0, 0.0, 60.0], device="cuda:0")   # the hunter
    sp[0, 1::2, :3] = torch.tensor([25.0, 0.0, 60.0], device="cuda:0")  # its quarry, 25 m dead ahead on the same heading
    eng.state[13:15] = sp
    obs = eng.env_reset().clone()
    side = eng.state[6]
    assert (side[0::2, 2].view(torch.int32) == 0).all() and (side[1::2, 2].view(torch.int32) == 1).all()
    assert torch.allclose(obs[0::2, 23 + 9:23 + 12].norm(dim=1), torch.full((E,), 25.0, device="cuda:0"), atol=1e-4)
    assert (obs[0::2, 23 + 9] > 24.9).all() and (obs[1::2, 23 + 9] < -24.9).all() and (obs[:, 23 + 13] == 0).all()
    act = torch.tensor([0.0, 0.0, 0.0, 0.2], device="cuda:0").repeat(2 * E, 1)
    for k, (hits, r_hunter, r_quarry, done) in enumerate(dogfight_tail_chase_expectations(dmg, aggressiveness=0.5, cooperativeness=0.5)):
        o, r, t, u = eng.env_step(act)
        side = eng.state[6]
        assert (side[1::2, 2].view(torch.int32) == hits).all() and (side[0::2, 2].view(torch.int32) == 0).all(), k
        assert torch.allclose(r[0::2], torch.full((E,), float(r_hunter), device="cuda:0"), atol=1e-3), (k, r[0].item(), r_hunter)
        assert torch.allclose(r[1::2], torch.full((E,), float(r_quarry), device="cuda:0"), atol=1e-3), (k, r[1].item(), r_quarry)
        assert bool(t.all()) == done and bool(t.any()) == done and not bool(u.any())
        assert torch.allclose(side[1::2, 0], torch.full((E,), max(0.0, 1 - dmg * hits), device="cuda:0"), atol=1e-5)
        if done:
            bits = side[:, 3].view(torch.int32)
            assert ((bits[0::2] & 128) != 0).all() and ((bits[1::2] & 16) != 0).all()  # team_win / dead
            break
    else:
        raise AssertionError("the quarry never died")
