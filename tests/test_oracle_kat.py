"""Analytic known-answer tests of the fp64 oracle's Bullet restatement (rows 10-12 of SURVEY.md 8(a)):
closed forms from tests/kat.py, no second restatement in the loop. CPU only."""
import ctypes as C

import numpy as np
import pytest

import kat
from oracle import oracle as O


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def tick(P, p, q, v, w, F=(0, 0, 0), tau=(0, 0, 0), n=1):
    F, tau = np.array(F, dtype=np.float64), np.array(tau, dtype=np.float64)
    for _ in range(n):
        O.lib().orc_rigid_tick(C.byref(P), dp(p), dp(q), dp(v), dp(w), dp(F), dp(tau))


def fresh(z=10.0):
    return np.array([0.0, 0.0, z]), np.array([0.0, 0.0, 0.0, 1.0]), np.zeros(3), np.zeros(3)


def test_free_fall_closed_form():
    P = O.make_params("quadx")
    p, q, v, w = fresh(50.0)
    for n in range(1, 401):
        tick(P, p, q, v, w)
        z, vz = kat.free_fall_z(50.0, n)
        assert abs(p[2] - z) < 1e-11 and abs(v[2] - vz) < 1e-12, n
    assert p[0] == 0.0 and p[1] == 0.0 and np.array_equal(q, [0, 0, 0, 1]) and not w.any()


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_constant_principal_axis_torque(axis):
    """w_n = n dt tau / I and the rotation angle dt^2 tau / I n (n + 1) / 2 about the same axis."""
    P = O.make_params("quadx", world_gravity_z=0.0)
    p, q, v, w = fresh()
    tau = np.zeros(3)
    tau[axis] = 2.0e-5
    for n in range(1, 101):
        tick(P, p, q, v, w, tau=tau)
        wn, th = kat.const_torque_principal(2.0e-5, axis, n)
        assert abs(w[axis] - wn) < 1e-10 * max(1.0, abs(wn)), n
        assert abs(np.delete(w, axis)).max() < 1e-13
        # q = (axis sin(th/2), cos(th/2))
        assert abs(q[axis] - np.sin(th / 2)) < 1e-9 and abs(q[3] - np.cos(th / 2)) < 1e-9, (n, th)
        assert abs(np.linalg.norm(q) - 1.0) < 1e-14
    assert not p[:2].any() and p[2] == 10.0  # no force, no gravity: the body stays put


@pytest.mark.parametrize("gyro", [1, 0])
def test_torque_free_principal_spin_is_stationary(gyro):
    """Torque-free spin about a principal axis: w x I w = 0, so w is constant with the gyroscopic term on and off."""
    P = O.make_params("quadx", world_gravity_z=0.0, world_use_gyro_term=gyro)
    for axis in range(3):
        p, q, v, w = fresh()
        w[axis] = 7.0
        tick(P, p, q, v, w, n=500)
        e = np.zeros(3)
        e[axis] = 7.0
        assert np.abs(w - e).max() < 1e-12
        th = 7.0 * kat.DT * 500
        assert abs(abs(q[3]) - abs(np.cos(th / 2))) < 1e-9


def test_gyroscopic_term_conserves_angular_momentum_direction():
    """A spin off the principal axes: with the term on, the world-frame angular momentum R I w_b keeps its
    direction and magnitude to first order in dt (explicit Euler drift only); with it off, w_world is constant."""
    on = O.make_params("quadx", world_gravity_z=0.0, world_use_gyro_term=1)
    off = O.make_params("quadx", world_gravity_z=0.0, world_use_gyro_term=0)
    w0 = np.array([3.0, -2.0, 5.0])
    p, q, v, w = fresh()
    w[:] = w0
    L0 = kat.I_DIAG * w0  # R = 1 at the start
    for _ in range(240):
        tick(on, p, q, v, w)
    R = np.zeros((3, 3))
    O.lib().orc_matrix_from_quat(dp(q), dp(R))
    L1 = R @ (kat.I_DIAG * (R.T @ w))
    assert np.linalg.norm(L1 - L0) / np.linalg.norm(L0) < 2e-2
    assert np.abs(w - w0).max() > 0.1  # the body axis precesses: w itself does change
    p, q, v, w = fresh()
    w[:] = w0
    tick(off, p, q, v, w, n=240)
    assert np.array_equal(w, w0)


def test_velocity_clamp():
    """+-100 per base-velocity coordinate (btMultiBody::m_maxCoordinateVelocity, applyDeltaVee)."""
    P = O.make_params("quadx", world_gravity_z=0.0)
    p, q, v, w = fresh()
    tick(P, p, q, v, w, F=(1e3, -1e3, 2e3), tau=(1.0, -1.0, 0.5), n=5)
    assert np.array_equal(np.abs(v), [kat.VMAX] * 3) and np.array_equal(np.abs(w), [kat.VMAX] * 3)
    # positions move with the clamped velocity
    p0 = p.copy()
    tick(P, p, q, v, w, F=(1e3, -1e3, 2e3), tau=(1.0, -1.0, 0.5))
    np.testing.assert_allclose(p - p0, kat.DT * v, atol=1e-12)
    assert abs(np.linalg.norm(q) - 1.0) < 1e-14


def test_euler_quat_round_trip_and_gimbal_branch():
    rng = np.random.default_rng(0)
    lib = O.lib()
    for _ in range(200):  # away from the branch: exact round trip
        rpy = rng.uniform([-3.1, -1.5, -3.1], [3.1, 1.5, 3.1])
        q, back = np.zeros(4), np.zeros(3)
        lib.orc_quat_from_euler(dp(rpy), dp(q))
        assert abs(np.linalg.norm(q) - 1.0) < 1e-15
        lib.orc_euler_from_quat(dp(q), dp(back))
        np.testing.assert_allclose(back, rpy, atol=1e-12)
    for sign in (1.0, -1.0):  # |sarg| >= 0.99999: roll = 0, pitch = +-pi/2, yaw = yaw -+ roll
        for _ in range(50):
            roll, yaw = rng.uniform(-1.5, 1.5, size=2)
            for pitch in (sign * np.pi / 2, sign * (np.pi / 2 - 1e-3)):  # sin(pi/2 - 1e-3) = 0.9999995 >= 0.99999
                rpy = np.array([roll, pitch, yaw])
                q, back = np.zeros(4), np.zeros(3)
                lib.orc_quat_from_euler(dp(rpy), dp(q))
                lib.orc_euler_from_quat(dp(q), dp(back))
                assert back[0] == 0.0 and back[1] == sign * np.pi / 2
                assert abs(back[2] - kat.gimbal_yaw(roll, yaw, sign)) < 2.5e-3  # (1e-3 off the pole: yaw mixes in O(1e-3))
        # just outside the branch the regular formulas apply
        rpy = np.array([0.3, sign * (np.pi / 2 - 1e-2), -0.4])
        q, back = np.zeros(4), np.zeros(3)
        lib.orc_quat_from_euler(dp(rpy), dp(q))
        lib.orc_euler_from_quat(dp(q), dp(back))
        np.testing.assert_allclose(back, rpy, atol=1e-9)


def test_matrix_from_denormalised_quaternion():
    """btMatrix3x3::setRotation divides by |q|^2: a scaled quaternion gives the same rotation."""
    rng = np.random.default_rng(1)
    q = rng.normal(size=4)
    R1, R2 = np.zeros((3, 3)), np.zeros((3, 3))
    O.lib().orc_matrix_from_quat(dp(q / np.linalg.norm(q)), dp(R1))
    O.lib().orc_matrix_from_quat(dp(q * 1.37), dp(R2))
    np.testing.assert_allclose(R1, R2, atol=1e-14)
    np.testing.assert_allclose(R1 @ R1.T, np.eye(3), atol=1e-14)


def _lane(P, mode, pos=(0, 0, 5.0)):
    L = O.Lane()
    O.lib().orc_aviary_reset(C.byref(P), C.byref(L), 0)
    O.lib().orc_set_mode(C.byref(P), C.byref(L), mode)
    return L


def test_motor_lag_closed_form():
    """Mode -1 (raw pwm), noise off: throttle_n = p (1 - (1 - dt/tau)^n) tick by tick."""
    P = O.make_params("quadx", noise_mode=O.NOISE_OFF, start_pos=[0, 0, 5.0])
    L = _lane(P, -1)
    pwm = [0.3, 0.5, 0.7, 0.9]
    for i, x in enumerate(pwm):
        L.setpoint[i] = x
    for s in range(1, 21):
        O.lib().orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
        np.testing.assert_allclose(list(L.throttle), kat.motor_lag(np.array(pwm), 2 * s), rtol=1e-13)


def test_hover_equilibrium_throttle():
    """throttle^2 = m g / total_thrust = 0.132435: thrust balances gravity (noise off, level)."""
    assert abs(kat.HOVER_THROTTLE_SQ - 0.132435) < 1e-12
    P = O.make_params("quadx", noise_mode=O.NOISE_OFF, start_pos=[0, 0, 5.0])
    L = _lane(P, -1)
    for i in range(4):
        L.setpoint[i] = np.sqrt(kat.HOVER_THROTTLE_SQ)
    for _ in range(40):  # the first-order lag converges: (1 - 0.41667)^80 ~ 2e-19
        O.lib().orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
    vz = L.v[2]
    for _ in range(60):
        O.lib().orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
    # only the body drag 7.35e-4 v^2 / m acts on the residual sink rate picked up during the spin-up
    drag_acc = 7.35e-4 * vz * vz / kat.MASS
    assert abs((L.v[2] - vz) - drag_acc * 120 * kat.DT) < 0.05 * drag_acc * 120 * kat.DT + 1e-12
    assert abs(L.w[0]) + abs(L.w[1]) + abs(L.w[2]) < 1e-15 and abs(L.p[0]) + abs(L.p[1]) < 1e-15


# ------------------------------------------------------------------ contact response (named-parameter model)
def _drop(model, pos, rpy, steps, mode=-1, vel=None, **kw):
    P = O.make_params(model, noise_mode=O.NOISE_OFF, start_pos=pos, start_rpy=rpy, **kw)
    if vel is not None:
        for i in range(3):
            P.start_vel[i] = vel[i]
    L = _lane(P, mode)
    traj = []
    for _ in range(steps):
        O.lib().orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
        traj.append([*L.p, *L.v, *L.w, *L.rpy, L.contact_step])
    return np.array(traj)


def test_contact_drop_comes_to_rest_at_half_height():
    """A level quad dropped from 0.2 m with the motors off. Contact points exist only once the box overlaps the slab
    (contact_margin 0: no speculative rows), so the impact tick may carry it at most v_impact dt into the floor -- no
    tunnelling --; restitution 0 (no bounce); the overlap is recovered at contact_erp per tick and it comes to rest with its
    collision box (0.09 x 0.09 x 0.02, cf2x.urdf:30-36) ON the floor: z = half-height (0.01) minus the allowed overlap
    (contact_slop, 1e-5) that keeps the contact report true while it rests."""
    SLOP = kat.CONTACT_SLOP
    t = _drop("quadx", [0.0, 0.0, 0.2], [0.0, 0.0, 0.0], 240)
    z, vz = t[:, 2], t[:, 5]
    k0 = int(np.argmax(t[:, 12]))
    v_imp = np.sqrt(2 * kat.G * 0.19)
    assert abs(k0 * 2 * kat.DT - np.sqrt(2 * 0.19 / kat.G)) < 3 * kat.DT  # free fall until the box reaches the floor
    assert z.min() > 0.01 - (v_imp + kat.G * kat.DT) * kat.DT     # no tunnelling: at most one tick of travel into the slab
    assert z[k0 + 2:].max() < 0.0101                          # restitution 0
    assert np.abs(z[-60:] - (0.01 - SLOP)).max() < 1e-6 and np.abs(vz[-60:]).max() < 1e-4  # at rest on the floor
    assert np.abs(t[-1, 6:12]).max() < 1e-4 and np.abs(t[-1, :2]).max() < 1e-4    # level drop: nothing sideways, no rotation
    assert t[k0:, 12].all()                                   # contact reported on every step from touch-down on


def test_contact_friction_stops_a_slide():
    """Coulomb friction mu = 0.5: a quad put on the floor with 1 m/s sideways decelerates at mu g and stops after
    v^2 / (2 mu g) = 0.102 m (discrete impulses: within 10 %)."""
    t = _drop("quadx", [0.0, 0.0, 0.01 - kat.CONTACT_SLOP], [0.0, 0.0, 0.0], 120, vel=[1.0, 0.0, 0.0])
    x, vx = t[:, 0], t[:, 3]
    assert abs(x[-1] - 1.0 / (2 * 0.5 * kat.G)) < 0.1 * 0.102 and abs(vx[-1]) < 1e-4
    k_stop = int(np.argmax(np.abs(vx) < 1e-4))
    assert abs(k_stop * 2 * kat.DT - 1.0 / (0.5 * kat.G)) < 0.03
    t0 = _drop("quadx", [0.0, 0.0, 0.01 - kat.CONTACT_SLOP], [0.0, 0.0, 0.0], 120, vel=[1.0, 0.0, 0.0], world_contact_friction=0.0)
    # frictionless: keeps sliding, slowed only by the body drag 7.35e-4 v^2 / m (0.027 m/s^2 at 1 m/s, for 1 s)
    assert abs(t0[-1, 3] - (1.0 - 7.35e-4 / kat.MASS)) < 2e-3


def test_contact_tilted_landing_rights_itself():
    t = _drop("quadx", [0.0, 0.0, 0.3], [0.5, -0.3, 1.0], 400)
    assert abs(t[-1, 2] - (0.01 - kat.CONTACT_SLOP)) < 1e-5 and np.abs(t[-1, 9:11]).max() < 1e-4 and np.abs(t[-1, 6:9]).max() < 1e-3


def test_rocket_settles_on_its_legs():
    """rocket.urdf:208-277: three leg boxes, bottoms 2.425 m under the base origin. Engine off, tank empty, dropped
    40 cm with a small tilt: it must end standing (z = 2.425, upright), not pass through the floor or topple."""
    t = _drop("rocket", [0.0, 0.0, 2.8], [0.03, -0.02, 0.4], 2400, mode=0, starting_fuel_ratio=0.0)
    assert abs(t[-1, 2] - (2.425 - kat.CONTACT_SLOP)) < 1e-4 and np.abs(t[-1, 9:11]).max() < 1e-3  # (2.425 minus the resting overlap)
    assert np.abs(t[-240:, 3:6]).max() < 1e-2 and np.abs(t[-240:, 6:9]).max() < 1e-2
    assert t[:, 2].min() > 2.425 - 0.02  # (impact: at most a tick of travel into the slab)


def test_contact_response_can_be_switched_off():
    t = _drop("quadx", [0.0, 0.0, 0.2], [0.0, 0.0, 0.0], 240, world_contact_response=0)
    assert t[-1, 2] < -3.0 and t[:, 12].any()  # detection only: the drone falls through the (10 m thick) slab


def test_dogfight_known_answers():
    """MAFixedwingDogfightEnv bookkeeping with answers that do not need the physics: 1 v 1, the hunter 25 m dead astern of its
    quarry on the same heading. Both fly the same trajectory, so the separation stays (25, 0, 0) and the engagement angles
    stay 0 (hunter) and pi (quarry): one hit per update -- the reset's update, then env_step_ratio = 4 per env step -- health
    1 - damage * hits (ma_fixedwing_dogfight_env.py:499-503), the quarry dies at hits >= (1 - 1e-3) / damage (:665-666) and the
    hunter wins in the same update (:682-690, reward overridden with 300); until then the hunter collects 20 per hit
    (+ cooperativeness, its own team's hits), the quarry -20 (1 - aggressiveness) per hit (:597-617)."""
    from kat import dogfight_tail_chase_expectations

    dmg, lethal = 0.05, 40.0
    pos = np.array([[0.0, 0.0, 60.0], [25.0, 0.0, 60.0]])
    W = O.OracleDogfight(pos, np.zeros((2, 3)), team_size=1, damage_per_hit=dmg, lethal_distance=lethal, lethal_angle=0.2, sparse_reward=True,
                         max_duration_seconds=10.0)
    obs = W.reset()
    exp = dogfight_tail_chase_expectations(dmg, aggressiveness=0.5, cooperativeness=0.5)
    assert np.array(W.D.received_hits[:2]).tolist() == [0, 1] and abs(W.health[1] - (1 - dmg)) < 1e-7
    # the other aircraft in the own body frame: 25 m away (exactly: both fly the same trajectory), ahead of the hunter, astern of the quarry
    assert abs(np.linalg.norm(obs[0][23 + 9:23 + 12]) - 25.0) < 1e-9 and abs(np.linalg.norm(obs[1][23 + 9:23 + 12]) - 25.0) < 1e-9
    assert obs[0][23 + 9] > 24.9 and obs[1][23 + 9] < -24.9
    assert obs[0][23 + 13] == 0.0 and obs[0][18] == 1.0 and abs(obs[0][23 + 12] - (1 - dmg)) < 1e-7  # opponent flag, own health, its health
    for k, (hits, r_hunter, r_quarry, done) in enumerate(exp):
        obs, rew, term, trunc = W.step(np.tile([0.0, 0.0, 0.0, 0.2], (2, 1)))
        assert int(W.D.received_hits[1]) == hits and int(W.D.received_hits[0]) == 0, (k, W.D.received_hits[1], hits)
        assert abs(rew[0] - r_hunter) < 1e-6 and abs(rew[1] - r_quarry) < 1e-6, (k, rew, r_hunter, r_quarry)
        assert bool(term[0]) == done and bool(term[1]) == done and not trunc.any()
        assert abs(W.health[1] - max(0.0, 1 - dmg * hits)) < 1e-6
        if done:
            assert (int(W.D.info_bits[0]) & 8) and (int(W.D.info_bits[1]) & 1)  # team_win / dead
            break
    else:
        raise AssertionError("the quarry never died")


def test_philox_round_function_against_the_published_vectors():
    """The counter-based generator both sides draw from (oracle/uav_oracle.c: orc_philox4x32_r; the device's philox4x32 is compared
    with it draw for draw by every Philox-noise parity test). Random123's known-answer vectors for Philox4x32-10 (and the zero
    vector of the seven-round variant) pin the round function and the key schedule; the product runs ORC_PHILOX_ROUNDS = 10."""
    import ctypes as C

    from oracle import oracle as O

    L = O.lib()

    def ph(key, ctr, rounds):
        out = (C.c_uint32 * 4)()
        L.orc_philox4x32_r(key, *ctr, rounds, out)
        return tuple(out)

    assert ph(0, (0, 0, 0, 0), 10) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert ph(0xFFFFFFFFFFFFFFFF, (0xFFFFFFFF,) * 4, 10) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert ph((0x299F31D0 << 32) | 0xA4093822, (0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), 10) == (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)
    assert ph(0, (0, 0, 0, 0), 7) == (0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48)
    out = (C.c_uint32 * 4)()
    L.orc_philox4x32(0, 0, 0, 0, 0, out)
    assert tuple(out) == ph(0, (0, 0, 0, 0), 10)  # (the product's round count)
    # the 23-bit uniforms the draws are made of: the mean of 40 000 of them
    u = (C.c_double * 4)()
    acc = []
    for i in range(10000):
        L.orc_uniform4(12345, i, 7, 0, 2, u)
        acc.extend(u)
    import numpy as np
    acc = np.array(acc)
    assert abs(acc.mean() - 0.5) < 5e-3 and abs(acc.var() - 1.0 / 12.0) < 2e-3


@pytest.mark.parametrize("env", ["hover", "quadx_waypoints"])
def test_consecutive_resets_draw_from_different_keys(env):
    """A reset's draws (settle noise, waypoints) are keyed by what the lane's PREVIOUS reset left behind, and that key is strictly
    increasing: round 5 stored the counter AT the reset, which is 0 again at a fresh lane's first reset, so that every lane's first and
    second episodes started from the same settle noise and flew to the same waypoints. Every one of five consecutive
    resets -- back to back, and with steps in between -- gives every lane a different observation, and the keys grow."""
    from oracle import oracle as O

    n = 64
    ob = O.OracleBatch(O.make_params(env, noise_mode=O.NOISE_PHILOX, seed=3), n)
    seen, keys = [], []
    rng = np.random.default_rng(0)
    for r in range(5):
        seen.append(ob.reset().copy())
        keys.append(np.array([ob.lanes[i].reset_key for i in range(n)], dtype=np.int64))
        for _ in range(r):  # (0, 1, 2 ... env steps between the resets)
            ob.step(rng.uniform(-0.2, 0.2, size=(n, 4)).astype(np.float32) + np.array([0, 0, 0, 0.4], dtype=np.float32))
    for a in range(5):
        for b in range(a + 1, 5):
            assert (np.abs(seen[a] - seen[b]).max(axis=1) > 1e-9).all(), (a, b)
    assert all((keys[r + 1] > keys[r]).all() for r in range(4)) and (keys[0] == 1).all()
