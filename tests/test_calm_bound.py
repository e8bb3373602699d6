"""The bound behind the specialised QuadX kernel's calm-wave path (quadx_fast.hpp: "this lane cannot come within reach of the floor
during this env step", which lets a wave run ticks without the contact response) restated in numpy and held against the fp64
oracle: over an env step no drone sinks further than |vz| T + A T (T + dt) / 2, A = g + thrust bound + drag bound -- for
thousands of lane-steps under random actions, motor noise on, many of them diving."""
import numpy as np

from oracle import oracle as O


def sink_bound(P, p, v, thr, n_ticks=None):
    w = P.world
    if n_ticks is None:
        n_ticks = P.env_step_ratio * w.ticks_per_control
    T = n_ticks * w.dt
    TT = 0.5 * T * (T + w.dt)
    smax = 1.0 + 9.0 * abs(P.noise_ratio[0])          # xi = num_motors + z, |z| <= 4.85 (quadk_from_params)
    fmaxM = P.thrust_coef[0] * P.max_rpm[0] ** 2 / P.mass
    # the motor state t' = ((1 - a) t + a pwm) s <= ((1 - a) t + a) smax: never beyond max(t, t*), t* the map's fixed point
    a = w.dt / P.motor_tau[0]
    assert (1.0 - a) * smax < 1.0
    tstar = a * smax / (1.0 - (1.0 - a) * smax)
    kt = 4.0 * fmaxM
    c = max(abs(P.drag_const[k]) for k in range(3)) / P.mass
    tm = np.abs(thr).max(axis=1)
    a_nd = kt * np.maximum(tm * tm, max(tstar * tstar, 1.0)) + abs(w.gravity_z)
    u = np.linalg.norm(v, axis=1) + a_nd * T
    return np.abs(v[:, 2]) * T + (c * u * u + a_nd) * TT


def test_no_drone_sinks_further_than_the_calm_bound():
    n, steps = 256, 160
    P = O.make_params("hover", noise_mode=O.NOISE_PHILOX, seed=3)
    ob = O.OracleBatch(P, n)
    ob.reset()
    rng = np.random.default_rng(0)
    checked, tight = 0, 0.0
    for k in range(steps):
        p0, v0, thr = ob.field("p"), ob.field("v"), ob.field("throttle")
        done0 = (ob.field("terminated") != 0) | (ob.field("truncated") != 0)   # these lanes are reset by this call (NEXT_STEP)
        a = rng.uniform(-1.0, 1.0, size=(n, 4))
        a[: n // 2, 3] = rng.uniform(-1.0, -0.3, size=n // 2)  # half of them dive
        ob.step(a, autoreset=1)
        p1 = ob.field("p")
        sink = sink_bound(P, p0, v0, thr)
        live = ~done0
        fell = p0[live, 2] - p1[live, 2]
        assert (fell <= sink[live] * 1.01 + 1e-3).all(), (k, float((fell - sink[live]).max()))
        checked += int(live.sum())
        tight = max(tight, float((fell / sink[live]).max()))
    assert checked > 0.8 * n * steps
    assert 0.2 < tight <= 1.0, tight        # the bound is reached to within a factor of five somewhere: not vacuous
    assert float(np.median(sink)) < 0.2     # ... and small enough to leave drones at z = 1 calm


def test_the_motor_state_never_exceeds_its_bound():
    """t' = ((1 - a) t + a pwm) (1 + xi m) with pwm <= 1, |xi| < 9: the state stays under max(t_0, t*) for ever."""
    P = O.make_params("hover", noise_mode=O.NOISE_PHILOX, seed=3)
    a, m = P.world.dt / P.motor_tau[0], abs(P.noise_ratio[0])
    smax = 1.0 + 9.0 * m
    tstar = a * smax / (1.0 - (1.0 - a) * smax)
    rng = np.random.default_rng(1)
    t = rng.uniform(0.0, 2.5, size=4096)
    bound = np.maximum(t, tstar)
    for _ in range(400):
        xi = np.clip(4.0 + rng.normal(size=t.shape), -8.99, 8.99)
        xi[:64] = 8.99  # the worst case, held
        t = ((1.0 - a) * t + a * rng.uniform(0.05, 1.0, size=t.shape)) * (1.0 + xi * m)
        assert (t <= bound * (1 + 1e-12)).all()
