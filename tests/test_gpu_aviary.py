"""GPU parity at the Aviary level (core/aviary.py surface): every QuadX flight mode (-1..7) and both
Fixedwing modes, per-lane spawn poses, setpoints that change during the run, Philox motor noise.
Compared per Aviary step against the fp64 oracle on `state(i)` (4,3) and `aux_state(i)`.

Tolerance: 1e-4 relative (vector-normalised). Every mode must hold it for EVERY lane over the first
25 Aviary steps (50 physics ticks). Over the full 120 steps it must hold for every lane in
the modes whose controller is well conditioned in fp32 (QuadX -1/0, Fixedwing); the cascaded QuadX
modes that go through the z PIDs amplify fp32 rounding by themselves (z_vel: kd/T = 6 per control
tick, cf2x.yaml:50-54) -- an fp32 build of the oracle drifts from the fp64 one just as far
(tests/tools/fp32_sensitivity.py, numbers in DESIGN.md) -- so there the full-horizon criterion is
the median lane error."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def sample_setpoint(rng, n, drone, mode):
    if drone == "fixedwing":
        d = 6 if mode == -1 else 4
        sp = rng.uniform(-1.0, 1.0, size=(n, d))
        sp[:, -1] = rng.uniform(0.0, 1.0, size=n)
        return sp
    if mode == -1:
        return rng.uniform(0.1, 0.6, size=(n, 4))
    u = lambda lo, hi, k=1: rng.uniform(lo, hi, size=(n, k))  # noqa: E731
    return {
        0: lambda: np.concatenate([u(-1, 1, 3), u(0.2, 0.6)], 1),
        1: lambda: np.concatenate([u(-0.4, 0.4, 3), u(-0.5, 0.5)], 1),
        2: lambda: np.concatenate([u(-0.5, 0.5, 3), u(0.5, 2.0)], 1),
        3: lambda: np.concatenate([u(-0.3, 0.3, 3), u(0.5, 2.0)], 1),
        4: lambda: np.concatenate([u(-1, 1, 2), u(-0.5, 0.5), u(0.5, 2.0)], 1),
        5: lambda: np.concatenate([u(-1, 1, 2), u(-0.5, 0.5), u(-0.5, 0.5)], 1),
        6: lambda: np.concatenate([u(-1, 1, 2), u(-0.5, 0.5), u(-0.5, 0.5)], 1),
        7: lambda: np.concatenate([u(-2, 2, 2), u(-1, 1), u(0.5, 2.5)], 1),
    }[mode]()


CASES = [("quadx", m, "cf2x") for m in range(-1, 8)] + [("fixedwing", 0, None), ("fixedwing", -1, None)] + \
        [("quadx", m, "primitive_drone") for m in (0, 6, 7)] + \
        [("fixedwing", 0, "acrowing"), ("fixedwing", -1, "acrowing")]  # drone_model=: quadx.py:29, ma_fixedwing_base_env.py:193-195


@pytest.mark.parametrize("drone,mode,model", CASES)
def test_aviary_parity(drone, mode, model):
    from pyflyt_amd.core import Aviary

    n, steps, seed = 128, 120, 40 + mode
    rng = np.random.default_rng(seed)
    z0 = 1.5 if drone == "quadx" else 10.0
    start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(z0, z0 + 1.0, size=(n, 1))], axis=1)
    start_orn = rng.uniform(-0.15, 0.15, size=(n, 3)) * np.array([1, 1, 5.0])
    primitive = model == "primitive_drone"
    env = Aviary(start_pos, start_orn, drone_type=drone, seed=seed,
                 drone_options=dict(drone_model=model) if model in ("primitive_drone", "acrowing") else None)
    env.set_mode(mode)

    lib = O.lib()
    Ps, Ls = [], []
    # the oracle sees the fp32-rounded spawn the device was given
    sp32 = start_pos.astype(np.float32).astype(np.float64)
    for i in range(n):
        P = O.make_params(model if model in ("primitive_drone", "acrowing") else drone, noise_mode=O.NOISE_PHILOX, seed=seed,
                          start_pos=sp32[i], start_rpy=start_orn[i])
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
        lib.orc_set_mode(C.byref(P), C.byref(L), mode)
        Ps.append(P); Ls.append(L)

    def ref_state():
        st = np.array([[list(L.w_b), list(L.rpy), list(L.v_b), list(L.p)] for L in Ls])
        aux = np.array([list(L.actuation) + [L.throttle[0]] if drone == "fixedwing" else list(L.throttle) for L in Ls])
        return st, aux

    # default setpoints after set_mode (quadx.py:275-290)
    spd = env.setpoints.cpu().numpy()
    ref_sp = np.array([list(L.setpoint)[: spd.shape[1]] for L in Ls])
    assert np.abs(spd - ref_sp).max() < 1e-5
    worst = 0.0
    ok = np.ones(n, dtype=bool)
    landed = np.zeros(n, dtype=bool)
    ok25 = None
    # primitive_drone's z_vel PID has kd/T = 0.2 * 120 = 24 per control tick (primitive_drone.yaml:50-54, cf2x: 6):
    # its cascaded modes chatter between the throttle limits and amplify fp32 rounding so fast that an
    # fp32 build of the ORACLE is itself 1e-3 .. 3e-2 (median lane) away from the fp64 one after 120
    # steps (`python tests/tools/fp32_sensitivity.py primitive_drone`: mode 6 86 %, mode 7 98 % of the
    # lanes beyond 1e-4). So: every lane within 1e-4 over the first 8 steps, bounded drift after that;
    # mode 0 (no z PIDs) holds the full criterion.
    strict_steps = 8 if (primitive and mode > 0) else 25
    for k in range(steps):
        if k % 20 == 5:
            sp = sample_setpoint(rng, n, drone, mode).astype(np.float32)
            env.set_all_setpoints(sp)
            for i, L in enumerate(Ls):
                for j in range(sp.shape[1]):
                    L.setpoint[j] = float(sp[i, j])
        env.step()
        for P, L in zip(Ps, Ls):
            lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
            L.rng_ctr += 1
        st, aux = ref_state()
        g = env.all_states.cpu().numpy().astype(np.float64)
        ga = env.all_aux_states.cpu().numpy().astype(np.float64)
        scale = np.maximum(1.0, np.linalg.norm(st, axis=2, keepdims=True))
        e = np.maximum((np.abs(g - st) / scale).reshape(n, -1).max(1), np.abs(ga - aux).max(1))
        contact = np.array([bool(L.contact_step) for L in Ls])
        # a lane that reaches the floor leaves the strict comparison one step before its first reported contact: from
        # there on it carries the contact solver's impulses (landings have their own test, test_landing_parity)
        landed |= contact | np.array([L.p[2] - P.bound_radius < 0.05 for P, L in zip(Ps, Ls)])
        ok &= landed | ((e < RTOL) & (env.contact_array.cpu().numpy() == contact))
        worst = max(worst, e[ok & ~landed].max() if (ok & ~landed).any() else 0.0)
        if k == strict_steps - 1:
            ok25 = ok.copy()
    med = float(np.median(e))
    print(f"aviary {drone}{'/primitive' if primitive else ''} mode {mode}: worst rel err {worst:.2e}, dropped@{strict_steps} {1 - ok25.mean():.4f}, dropped@{steps} {1 - ok.mean():.4f}, "
          f"median lane err at end {med:.1e}")
    assert ok25.all(), f"lanes beyond 1e-4 within the first {strict_steps} Aviary steps: {np.nonzero(~ok25)[0][:8]}"
    assert med < (RTOL if not (primitive and mode > 0) else 0.1)
    if drone == "fixedwing" or mode in (-1, 0):  # no z PIDs in the loop: every lane, all 120 steps
        assert ok.all(), f"lanes beyond 1e-4 over {steps} Aviary steps: {np.nonzero(~ok)[0][:8]}"
    env.disconnect()


def test_primitive_drone_prop_disc_contact():
    """Tilted drops of the primitive drone: a prop disc (cylinder collider, primitive_drone.urdf:42-47)
    reaches the floor before the base box; the contact flag must rise on the same Aviary step as in
    the oracle for every lane."""
    from pyflyt_amd.core import Aviary

    n = 128
    rng = np.random.default_rng(5)
    start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(0.25, 0.5, size=(n, 1))], axis=1)
    start_orn = np.concatenate([rng.uniform(-0.7, 0.7, size=(n, 2)), rng.uniform(-3, 3, size=(n, 1))], axis=1)
    env = Aviary(start_pos, start_orn, drone_type="quadx", seed=1, motor_noise=False, drone_options=dict(drone_model="primitive_drone"))
    env.set_mode(0)
    lib = O.lib()
    sp32 = start_pos.astype(np.float32).astype(np.float64)
    Ps, Ls = [], []
    for i in range(n):
        P = O.make_params("primitive_drone", noise_mode=O.NOISE_OFF, start_pos=sp32[i], start_rpy=start_orn[i])
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
        lib.orc_set_mode(C.byref(P), C.byref(L), 0)
        Ps.append(P); Ls.append(L)
    first_g = np.full(n, -1); first_r = np.full(n, -1)
    for k in range(80):
        env.step()
        cg = env.contact_array.cpu().numpy()
        for i, (P, L) in enumerate(zip(Ps, Ls)):
            lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
            if L.contact_step and first_r[i] < 0:
                first_r[i] = k
        first_g[(first_g < 0) & cg] = k
    assert (first_r >= 0).all()
    # a lane whose lowest point crosses z = 0 within fp32 rounding of a tick boundary may flip by one step
    assert (np.abs(first_g - first_r) <= 1).all() and (first_g == first_r).mean() >= 0.97
    env.disconnect()


def test_aviary_fused_steps_equal_single_steps():
    """step(n_steps=k) == k x step(): the fused launch holds the setpoints as the reference does."""
    from pyflyt_amd.core import Aviary

    pos = np.tile(np.array([[0.0, 0.0, 2.0]]), (64, 1))
    a = Aviary(pos, np.zeros((64, 3)), "quadx", seed=3)
    b = Aviary(pos, np.zeros((64, 3)), "quadx", seed=3)
    for e in (a, b):
        e.set_mode(6)
        e.set_all_setpoints(np.tile(np.array([[0.5, -0.3, 0.2, 0.1]]), (64, 1)))
    a.step(n_steps=8)
    for _ in range(8):
        b.step()
    # (not bit-identical: the compiler contracts mul+add into fma differently in the two inlined copies)
    assert torch.allclose(a.all_states, b.all_states, rtol=1e-5, atol=1e-6)
    assert torch.allclose(a.all_aux_states, b.all_aux_states, rtol=1e-5, atol=1e-6)
    assert a.physics_steps == b.physics_steps == 16


def test_aviary_errors():
    from pyflyt_amd.core import Aviary, AviaryInitException

    with pytest.raises(AviaryInitException):
        Aviary(np.zeros((4, 2)), np.zeros((4, 2)))
    with pytest.raises(AviaryInitException):
        Aviary(np.zeros((4, 3)), np.zeros((4, 3)), drone_type="quadplane")
    env = Aviary(np.array([[0, 0, 1.0]]), np.zeros((1, 3)))
    with pytest.raises(ValueError):
        env.set_mode(8)
    rk = Aviary(np.array([[0, 0, 50.0]]), np.zeros((1, 3)), drone_type="rocket")
    with pytest.raises(ValueError):
        rk.set_mode(1)  # rocket.py:238-247: mode 0 only
    rk.disconnect()


@pytest.mark.parametrize("drone", ["quadx", "fixedwing"])
@pytest.mark.parametrize("world", [
    dict(physics_hz=480),                                   # 4 ticks per control step, dt = 1/480
    dict(use_gyro_term=False),                              # the doubtful Bullet facts are parameters:
    dict(max_coord_vel=6.0),                                #   a velocity clamp that actually bites
    dict(gravity_z=-3.71, world_scale=2.0),
])
def test_world_options_reach_the_kernels(drone, world):
    """Non-default world constants take the generic path end to end: same constants on both sides,
    same trajectories. (These are exactly the named [BULLET-FROM-MEMORY] parameters, so a correction
    from a real PyBullet run is a parameter change, not a code change.)"""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    n, steps, seed = 96, 60, 90
    rng = np.random.default_rng(seed)
    z0 = 1.5 if drone == "quadx" else 10.0
    start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(z0, z0 + 1.0, size=(n, 1))], axis=1).astype(np.float32)
    start_orn = rng.uniform(-0.15, 0.15, size=(n, 3)) * np.array([1, 1, 5.0])
    from pyflyt_amd.params import quat_from_euler

    P = build_params(drone, "none", noise="philox", autoreset="off", seed=seed, world_options=world)
    eng = BatchEngine(P, n, device="cuda:0")
    pose = torch.tensor(np.concatenate([start_pos, np.stack([quat_from_euler(o) for o in start_orn])], axis=1), dtype=torch.float32, device="cuda:0").contiguous()
    eng.aviary_reset(pose)
    sp = torch.zeros(n, 4, device="cuda:0")
    eng.aviary_set_mode(0, sp)
    hz = world.get("physics_hz", 240)
    over = dict(world_dt=1.0 / hz, world_ticks_per_control=hz // 120)
    if "use_gyro_term" in world:
        over["world_use_gyro_term"] = int(world["use_gyro_term"])
    if "max_coord_vel" in world:
        over["world_max_coord_vel"] = world["max_coord_vel"]
    if "gravity_z" in world:
        over["world_gravity_z"] = world["gravity_z"]
    if "world_scale" in world:
        over["world_plane_half_xy"] = 15.0 * world["world_scale"]
        over["world_plane_half_z"] = 5.0 * world["world_scale"]
    lib = O.lib()
    Ps, Ls = [], []
    for i in range(n):
        Pi = O.make_params(drone, noise_mode=O.NOISE_PHILOX, seed=seed, start_pos=start_pos[i].astype(np.float64), start_rpy=start_orn[i], **over)
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(Pi), C.byref(L), i)
        lib.orc_set_mode(C.byref(Pi), C.byref(L), 0)
        Ps.append(Pi); Ls.append(L)
    worst = 0.0
    for k in range(steps):
        if k % 15 == 3:
            s = sample_setpoint(rng, n, drone, 0).astype(np.float32)
            sp.copy_(torch.tensor(s))
            for i, L in enumerate(Ls):
                for j in range(4):
                    L.setpoint[j] = float(s[i, j])
        st_g, aux_g = eng.aviary_step(sp)
        for Pi, L in zip(Ps, Ls):
            lib.orc_aviary_step(C.byref(Pi), C.byref(L), None, 0, 0)
            L.rng_ctr += 1
        st = np.array([list(L.w_b) + list(L.rpy) + list(L.v_b) + list(L.p) for L in Ls]).reshape(n, 4, 3)
        g = st_g.cpu().numpy().astype(np.float64).reshape(n, 4, 3)
        scale = np.maximum(1.0, np.linalg.norm(st, axis=2, keepdims=True))
        worst = max(worst, float((np.abs(g - st) / scale).max()))
    print(f"world {world} {drone}: worst {worst:.2e}")
    assert worst < RTOL
    if "max_coord_vel" in world and drone == "fixedwing":
        assert np.abs(np.array([list(L.v) for L in Ls])).max() <= world["max_coord_vel"] + 1e-9  # 20 m/s spawn, clamped
    eng.close()


def test_default_control_reaches_setpoints():
    """Mirror of the reference's tests/test_core.py:65-93 (`test_default_control`): position control
    (mode 7) to (1, 0, 1) for 500 Aviary steps, then to (0, 0, 2) with a 45 degree yaw for 500 more.
    The reference only checks that this runs; here the drone must also arrive (and the oracle agrees)."""
    from pyflyt_amd.core import Aviary

    env = Aviary(np.array([[0.0, 0.0, 1.0]]), np.array([[0.0, 0.0, 0.0]]), drone_type="quadx", seed=0)
    env.set_mode(7)
    P = O.make_params("quadx", noise_mode=O.NOISE_PHILOX, seed=0, start_pos=[0.0, 0.0, 1.0])
    L = O.Lane()
    lib = O.lib()
    lib.orc_aviary_reset(C.byref(P), C.byref(L), 0)
    lib.orc_set_mode(C.byref(P), C.byref(L), 7)
    for target, steps in (((1.0, 0.0, 0.0, 1.0), 500), ((0.0, 0.0, np.pi / 4, 2.0), 500)):
        env.set_setpoint(0, np.array(target))
        for j, x in enumerate(target):
            L.setpoint[j] = np.float32(x)
        for _ in range(steps):
            env.step()
            lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
            L.rng_ctr += 1
        st = env.state(0).cpu().numpy()
        want = np.array([target[0], target[1], target[3]])
        # the cascade is P-only in position (cf2x.yaml:36-41): after ~4 s it is within ~10 cm, still settling
        assert np.abs(st[3] - want).max() < 0.15, (st[3], want)            # lin_pos row
        assert abs(st[1][2] - target[2]) < 0.05                            # yaw
        assert np.abs(np.array(list(L.p)) - st[3]).max() < 0.05            # and the oracle is at the same place
    env.disconnect()


@pytest.mark.parametrize("mode,steps", [(0, 40), (1, 8), (7, 6)])
def test_multi_spawn_different_control_rates(mode, steps):
    """Mirror of the reference's tests/test_core.py:34-62 (`test_multi_spawn`): three drones in one Aviary
    with control_hz 60 / 120 / 240. The Aviary steps at the slowest controller's rate (4 physics ticks,
    aviary.py:288-289); each drone's controller fires at its own rate with its own PID period. Checked
    against per-drone oracles stepped at their own rates (mode 7 as in the reference's test, over the
    window in which its fp32-sensitive z PIDs still hold 1e-4; the rate and angle modes for longer)."""
    from pyflyt_amd.core import Aviary

    rates = [60, 120, 240] * 16
    n = len(rates)
    rng = np.random.default_rng(2)
    # (high enough that no drone reaches the floor within the run: this test is about control rates, landings have their own)
    start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(3.0, 4.0, size=(n, 1))], axis=1).astype(np.float32)
    start_orn = np.zeros((n, 3))
    env = Aviary(start_pos, start_orn, drone_type="quadx", seed=6, drone_options=[dict(control_hz=hz) for hz in rates])
    assert env.updates_per_step == 4
    env.set_mode(mode)
    lib = O.lib()
    Ps, Ls = [], []
    for i, hz in enumerate(rates):
        P = O.make_params("quadx", noise_mode=O.NOISE_OFF, start_pos=start_pos[i].astype(np.float64), control_period=1.0 / hz,
                          world_ticks_per_control=240 // hz)
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
        lib.orc_set_mode(C.byref(P), C.byref(L), mode)
        Ps.append(P); Ls.append(L)
    env.disconnect()
    env = Aviary(start_pos, start_orn, drone_type="quadx", seed=6, motor_noise=False, drone_options=[dict(control_hz=hz) for hz in rates])
    env.set_mode(mode)
    sp = sample_setpoint(rng, n, "quadx", mode).astype(np.float32)
    env.set_all_setpoints(sp)
    for i, L in enumerate(Ls):
        for j in range(4):
            L.setpoint[j] = float(sp[i, j])
    worst = 0.0
    for k in range(steps):
        env.step()
        for P, L, hz in zip(Ps, Ls, rates):
            for _ in range(hz // 60):  # an Aviary step of this world = 4 ticks = hz/60 of the drone's own control steps
                lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
        st = np.array([list(L.w_b) + list(L.rpy) + list(L.v_b) + list(L.p) for L in Ls]).reshape(n, 4, 3)
        g = env.all_states.cpu().numpy().astype(np.float64)
        scale = np.maximum(1.0, np.linalg.norm(st, axis=2, keepdims=True))
        worst = max(worst, float((np.abs(g - st) / scale).max()))
    print(f"mixed control rates mode {mode}: worst {worst:.2e}")
    assert worst < RTOL
    with pytest.raises(AssertionError):  # aviary.py:292-297
        Aviary(start_pos[:2], start_orn[:2], drone_type="quadx", drone_options=[dict(control_hz=80), dict(control_hz=120)])
    env.disconnect()


@pytest.mark.parametrize("fuel", [0.05, 0.6])
def test_rocket_aviary_parity(fuel):
    """Rocket (drones/rocket.py) through the batched Aviary against the oracle (itself pinned on three
    reference-generated trajectories): grid fins, gimballed booster, fuel burn with the composite mass /
    centre of mass / inertia rebuilt every tick, per-axis body drag, Philox booster noise."""
    from pyflyt_amd.core import Aviary

    n, steps, seed = 128, 150, 77
    rng = np.random.default_rng(seed)
    start_pos = np.concatenate([rng.uniform(-5, 5, size=(n, 2)), rng.uniform(100.0, 200.0, size=(n, 1))], axis=1).astype(np.float32)
    start_orn = rng.uniform(-0.2, 0.2, size=(n, 3)) * np.array([1, 1, 5.0])
    env = Aviary(start_pos, start_orn, drone_type="rocket", seed=seed, drone_options=dict(starting_fuel_ratio=fuel))
    env.set_mode(0)
    assert env.setpoints.shape == (n, 7) and env.all_aux_states.shape == (n, 9)
    lib = O.lib()
    Ps, Ls = [], []
    for i in range(n):
        P = O.make_params("rocket", noise_mode=O.NOISE_PHILOX, seed=seed, start_pos=start_pos[i].astype(np.float64),
                          start_rpy=start_orn[i], starting_fuel_ratio=fuel)
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
        lib.orc_set_mode(C.byref(P), C.byref(L), 0)
        Ps.append(P); Ls.append(L)
    worst = 0.0
    for k in range(steps):
        if k % 25 == 3:  # fins x, y, yaw | ignition | throttle | gimbal 1, 2 (rocket.py:230-236)
            sp = np.concatenate([rng.uniform(-0.6, 0.6, size=(n, 3)), (rng.random((n, 1)) < 0.8).astype(np.float64),
                                 rng.uniform(0, 1, size=(n, 1)), rng.uniform(-1, 1, size=(n, 2))], axis=1).astype(np.float32)
            env.set_all_setpoints(sp)
            for i, L in enumerate(Ls):
                for j in range(7):
                    L.setpoint[j] = float(sp[i, j])
        env.step()
        for P, L in zip(Ps, Ls):
            lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
            L.rng_ctr += 1
        st = np.array([list(L.w_b) + list(L.rpy) + list(L.v_b) + list(L.p) for L in Ls]).reshape(n, 4, 3)
        aux = np.array([list(L.actuation)[:4] + [float(L.ignition), L.fuel_ratio, L.throttle[0]] + list(L.gimbal) for L in Ls])
        g = env.all_states.cpu().numpy().astype(np.float64)
        ga = env.all_aux_states.cpu().numpy().astype(np.float64)
        scale = np.maximum(1.0, np.linalg.norm(st, axis=2, keepdims=True))
        worst = max(worst, float((np.abs(g - st) / scale).max()), float(np.abs(ga - aux).max()))
    print(f"rocket fuel {fuel}: worst {worst:.2e}")
    assert worst < RTOL
    assert (aux[:, 5] < fuel).any()  # fuel was burnt
    env.disconnect()


def test_rocket_drop_contact():
    """Engine off, tilted, from a few metres: a leg (yaw-rotated box), the booster or the body cylinder
    reaches the floor; the contact flag must rise with the oracle's (within one Aviary step at a rounding
    boundary)."""
    from pyflyt_amd.core import Aviary

    n = 96
    rng = np.random.default_rng(8)
    start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(3.0, 5.0, size=(n, 1))], axis=1).astype(np.float32)
    start_orn = np.concatenate([rng.uniform(-0.6, 0.6, size=(n, 2)), rng.uniform(-3, 3, size=(n, 1))], axis=1)
    env = Aviary(start_pos, start_orn, drone_type="rocket", seed=1, motor_noise=False, drone_options=dict(starting_fuel_ratio=0.0))
    lib = O.lib()
    Ps, Ls = [], []
    for i in range(n):
        P = O.make_params("rocket", noise_mode=O.NOISE_OFF, start_pos=start_pos[i].astype(np.float64), start_rpy=start_orn[i], starting_fuel_ratio=0.0)
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
        Ps.append(P); Ls.append(L)
    first_g = np.full(n, -1); first_r = np.full(n, -1)
    for k in range(120):
        env.step()
        cg = env.contact_array.cpu().numpy()
        for i, (P, L) in enumerate(zip(Ps, Ls)):
            lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
            if L.contact_step and first_r[i] < 0:
                first_r[i] = k
        first_g[(first_g < 0) & cg] = k
    assert (first_r >= 0).all()
    assert (np.abs(first_g - first_r) <= 1).all() and (first_g == first_r).mean() >= 0.97
    env.disconnect()


def test_mixed_drones():
    """Mirror of the reference's tests/test_core.py:228-259 (`test_mixed_drones`): a rocket, a quadx and a
    fixedwing in one Aviary with per-drone options and per-drone flight modes. Each drone must fly exactly
    as it does alone (same seed, same global lane), and the combined accessors keep the caller's order."""
    from pyflyt_amd.core import Aviary, MixedAviary

    start_pos = np.array([[0.0, 5.0, 5.0], [0.0, 0.0, 1.0], [5.0, 0.0, 1.0], [1.0, 1.0, 2.0]])
    start_orn = np.zeros_like(start_pos)
    start_orn[0, 0] = np.pi / 2
    types = ["rocket", "quadx", "fixedwing", "quadx"]
    options = [dict(), dict(use_camera=True), dict(starting_velocity=np.array([0.0, 0.0, 0.0])), dict(use_camera=True)]
    env = Aviary(start_pos=start_pos, start_orn=start_orn, render=False, drone_type=types, drone_options=options, seed=11)
    assert isinstance(env, MixedAviary) and env.num_drones == 4
    env.set_mode([0, 7, 0, 7])
    alone = [Aviary(start_pos[[i]], start_orn[[i]], drone_type=t, drone_options=options[i], seed=11, lane_offset={0: 0, 1: 1, 2: 2, 3: 2}[i])
             for i, t in enumerate(types)]
    # (the two quadx share one engine: lanes 1 and 2 of the global numbering -> offsets 1 and 1+1)
    for a, m in zip(alone, [0, 7, 0, 7]):
        a.set_mode(m)
    env.set_setpoint(1, np.array([1.0, 0.0, 0.0, 2.0])); alone[1].set_setpoint(0, np.array([1.0, 0.0, 0.0, 2.0]))
    env.set_setpoint(0, np.array([0.1, 0.0, 0.0, 1.0, 0.5, 0.2, -0.2])); alone[0].set_setpoint(0, np.array([0.1, 0.0, 0.0, 1.0, 0.5, 0.2, -0.2]))
    for _ in range(100):
        _ = env.all_states
        env.step()
        for a in alone:
            a.step()
    st = env.all_states
    assert st.shape == (4, 4, 3)
    for i, a in enumerate(alone):
        assert torch.equal(st[i], a.state(0)), i
        assert torch.equal(env.aux_state(i), a.aux_state(0))
    assert [len(x) for x in env.all_aux_states] == [9, 4, 6, 4]
    assert env.contact_array.shape == (4,)
    env.set_mode([0, 7, 0, 6])  # the two quadx now in different modes: a per-lane mode buffer inside their engine
    env.step()
    env.disconnect()
    for a in alone:
        a.disconnect()


def test_custom_controller():
    """Mirror of the reference's tests/test_core.py:141-192 (`test_custom_controller`): controller id 8 on top
    of base mode 6, steering to (1, 1, 1) with a constant yaw rate -- here batched over device tensors, and
    checked against the oracle running mode 6 with the same law evaluated in numpy."""
    from pyflyt_amd.core import Aviary

    class CustomController:
        def __init__(self):
            self.calls = 0

        def reset(self):
            pass

        def step(self, state, setpoint):  # state [N, 4, 3], setpoint [N, 4]
            self.calls += 1
            target_velocity = torch.tensor([1.0, 1.0, 1.0], device=state.device) - state[:, 3]
            yaw_rate = torch.full_like(target_velocity[:, :1], 0.5)
            return torch.cat([target_velocity[:, :2], yaw_rate, target_velocity[:, 2:]], dim=1)

    n = 32
    rng = np.random.default_rng(4)
    start_pos = np.concatenate([rng.uniform(-0.5, 0.5, size=(n, 2)), rng.uniform(0.8, 1.2, size=(n, 1))], axis=1).astype(np.float32)
    env = Aviary(start_pos, np.zeros((n, 3)), drone_type="quadx", seed=9, motor_noise=False)
    env.drones[0].register_controller(controller_constructor=CustomController, controller_id=8, base_mode=6)
    env.set_mode(8)
    lib = O.lib()
    Ps, Ls = [], []
    for i in range(n):
        P = O.make_params("quadx", noise_mode=O.NOISE_OFF, start_pos=start_pos[i].astype(np.float64))
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
        lib.orc_set_mode(C.byref(P), C.byref(L), 6)
        Ps.append(P); Ls.append(L)
    worst = 0.0
    for k in range(600):
        env.step()
        for P, L in zip(Ps, Ls):
            tv = np.array([1.0, 1.0, 1.0]) - np.array(list(L.p))
            for j, x in enumerate([tv[0], tv[1], 0.5, tv[2]]):
                L.setpoint[j] = float(np.float32(x))
            lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
        if k < 12:  # mode 6 is one of the fp32-sensitive cascades: point-wise parity over its strict window only
            st = np.array([list(L.w_b) + list(L.rpy) + list(L.v_b) + list(L.p) for L in Ls]).reshape(n, 4, 3)
            worst = max(worst, float(np.abs(env.all_states.cpu().numpy() - st).max()))
    assert worst < 1e-3, worst
    pos = env.all_states[:, 3].cpu().numpy()
    assert np.abs(pos - 1.0).max() < 0.2, pos  # arriving at (1, 1, 1), as the oracle does:
    assert np.abs(np.array([list(L.p) for L in Ls]) - 1.0).max() < 0.2
    assert env._controller.calls == 600
    with pytest.raises(AssertionError):
        env.register_controller(controller_id=3, controller_constructor=CustomController, base_mode=6)  # a default mode id
    env.disconnect()


def test_per_drone_flight_modes_and_spawn_velocities():
    """`Aviary.set_mode([...])` with a different mode per QuadX (core/aviary.py:449-455) and per-drone
    `starting_velocity` for Fixedwings (fixedwing.py:35,201; the dogfight env spawns every aircraft with its
    own velocity, ma_fixedwing_dogfight_env.py:218-222): each drone must fly exactly as in a single-mode /
    single-velocity Aviary of its own."""
    from pyflyt_amd.core import Aviary

    modes = [0, 7, 6, -1, 4, 7, 1, 0]
    n = len(modes)
    rng = np.random.default_rng(12)
    pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(1.0, 2.0, size=(n, 1))], axis=1)
    orn = rng.uniform(-0.1, 0.1, size=(n, 3))
    env = Aviary(pos, orn, drone_type="quadx", seed=21)
    env.set_mode(modes)
    sps = [sample_setpoint(rng, 1, "quadx", m)[0].astype(np.float32) for m in modes]
    alone = []
    for i, m in enumerate(modes):
        a = Aviary(pos[[i]], orn[[i]], drone_type="quadx", seed=21, lane_offset=i)
        a.set_mode(m)
        assert torch.allclose(a.setpoints[0], env.setpoints[i])  # the per-mode default setpoints (quadx.py:275-290)
        a.set_setpoint(0, sps[i]); env.set_setpoint(i, sps[i])
        alone.append(a)
    for _ in range(40):
        env.step()
        for a in alone:
            a.step()
    for i, a in enumerate(alone):
        assert torch.allclose(env.state(i), a.state(0), rtol=1e-5, atol=1e-6), (i, modes[i])
        assert torch.allclose(env.aux_state(i), a.aux_state(0), rtol=1e-5, atol=1e-6)
        a.disconnect()
    env.disconnect()

    vels = [np.array([20.0, 0.0, 0.0]), np.array([0.0, 15.0, 1.0]), np.array([-12.0, -12.0, 0.0])]
    fpos = np.array([[0.0, 0.0, 30.0]] * 3); forn = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 1.57], [0.0, 0.1, -2.36]])
    env = Aviary(fpos, forn, drone_type="fixedwing", seed=5, drone_options=[dict(starting_velocity=v) for v in vels])
    for i, v in enumerate(vels):
        a = Aviary(fpos[[i]], forn[[i]], drone_type="fixedwing", seed=5, lane_offset=i, drone_options=dict(starting_velocity=v))
        for _ in range(30):
            a.step()
        alone.append(a)
    for _ in range(30):
        env.step()
    for i in range(3):
        assert torch.allclose(env.state(i), alone[n + i].state(0), rtol=1e-5, atol=1e-5), i
        alone[n + i].disconnect()
    env.disconnect()


def test_set_armed():
    """core/aviary.py:423-438,510-521: a disarmed drone is skipped by control / physics / state updates --
    it free-falls (Bullet still integrates it), its reported state stays frozen; re-arming resumes it."""
    from pyflyt_amd.core import Aviary

    pos = np.array([[0.0, 0.0, 3.0], [1.0, 0.0, 3.0], [2.0, 0.0, 3.0]])
    env = Aviary(pos, np.zeros((3, 3)), drone_type="quadx", seed=2, motor_noise=False)
    env.set_mode(7)
    env.set_armed([True, False, True])
    frozen = env.state(1).clone()
    for _ in range(30):
        env.step()
    assert torch.equal(env.state(1), frozen)                       # no read-back
    z = env.engine.state[0][:, 2].cpu().numpy()
    t = 60 / 240.0
    assert abs(z[1] - (3.0 - 0.5 * 9.81 * t * (t + 1 / 240.0))) < 1e-3   # semi-implicit free fall: z0 - g dt^2 n(n+1)/2
    assert abs(z[0] - 3.0) < 0.05 and abs(z[2] - 3.0) < 0.05       # the armed ones hold position (mode 7)
    env.set_armed(True)
    env.step()
    assert not torch.equal(env.state(1), frozen) and env.state(1)[2, 2] < -1.0   # now reported: falling at ~2.5 m/s
    env.disconnect()


@pytest.mark.parametrize("drone,model,z0,tilt,steps,settle,impact_tol,strict_late,min_rest", [
    ("quadx", "cf2x", 0.25, 0.6, 240, 30, 2e-3, True, 0.99),
    ("quadx", "primitive_drone", 0.45, 0.6, 500, 60, 0.5, False, 0.9),
    ("fixedwing", None, 0.6, 0.3, 400, 60, 0.5, False, 0.0),   # (keeps sliding on its six boxes: nothing at rest within the run)
    ("rocket", None, 2.45, 0.02, 900, 200, 0.5, False, 0.9)])
def test_landing_parity(drone, model, z0, tilt, steps, settle, impact_tol, strict_late, min_rest):
    """The contact response (uav_vehicles.hpp:contact_solve_impl) against the oracle's through whole landings: tilted drops with
    the motors off, from free fall through the first impacts to rest. Three regimes, each with its own bound:
      free fall (until the body comes within 5 cm of the floor's reach): 1e-4 for every lane;
      the impact transient: touch-down is non-smooth (clamps at zero normal impulse and at the friction cone, vertices
        entering and leaving the contact set) and plain fp32 arithmetic drifts from fp64 there by itself -- an fp32 build
        of the ORACLE is 4e-4 (quad) to 5e-1 (a toppling rocket) away from the fp64 one
        (tests/tools/fp32_contact_sensitivity.py); the quad is held to 2e-3 and must be BACK within 1e-4 fifteen steps later;
      the outcome: wherever the oracle has come to rest, the device rests in the same pose (1e-4 m, 1e-3 rad; measured 2e-9 m
        for the quads, 3e-5 m for the 9.5 m rocket)."""
    from pyflyt_amd.core import Aviary

    n, seed = 128, 77
    rng = np.random.default_rng(seed)
    start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(z0, z0 + 0.2, size=(n, 1))], axis=1)
    start_orn = np.concatenate([rng.uniform(-tilt, tilt, size=(n, 2)), rng.uniform(-3, 3, size=(n, 1))], axis=1)
    opts = {}
    if model in ("primitive_drone",):
        opts["drone_model"] = model
    if drone == "rocket":
        opts["starting_fuel_ratio"] = 0.0
    if drone == "fixedwing":
        opts["starting_velocity"] = (0.0, 0.0, 0.0)  # (a drop, not a fly-by: the 30 m slab is left within a second at 20 m/s)
    env = Aviary(start_pos, start_orn, drone_type=drone, motor_noise=False, drone_options=opts or None)
    mode = 0 if drone != "quadx" else -1
    env.set_mode(mode)
    env.set_all_setpoints(np.zeros((n, env.setpoints.shape[1])))
    lib = O.lib()
    sp32 = start_pos.astype(np.float32).astype(np.float64)
    Ps, Ls = [], []
    extra = dict(starting_fuel_ratio=0.0) if drone == "rocket" else {}
    if drone == "fixedwing":
        extra["start_vel"] = [0.0, 0.0, 0.0]
    for i in range(n):
        P = O.make_params(model if model == "primitive_drone" else drone, noise_mode=O.NOISE_OFF, start_pos=sp32[i], start_rpy=start_orn[i], **extra)
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
        lib.orc_set_mode(C.byref(P), C.byref(L), mode)
        for j in range(8):
            L.setpoint[j] = 0.0
        Ps.append(P); Ls.append(L)
    first = np.full(n, -1)
    ok_early = np.ones(n, dtype=bool)
    ok_late = np.ones(n, dtype=bool)
    worst_early = worst_impact = worst_late = 0.0
    for k in range(steps):
        env.step()
        for P, L in zip(Ps, Ls):
            lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
        st = np.array([[list(L.w_b), list(L.rpy), list(L.v_b), list(L.p)] for L in Ls])
        contact = np.array([bool(L.contact_step) for L in Ls])
        near = np.array([L.p[2] - P.bound_radius < 0.05 for P, L in zip(Ps, Ls)])  # contact constraints may act from here on
        first[(first < 0) & (contact | near)] = k
        g = env.all_states.cpu().numpy().astype(np.float64)
        scale = np.maximum(1.0, np.linalg.norm(st, axis=2, keepdims=True))
        e = (np.abs(g - st) / scale).reshape(n, -1).max(1)
        early = first < 0
        late = (first >= 0) & (k >= first + settle)
        ok_early &= ~early | (e < RTOL)
        ok_late &= ~late | (e < RTOL)
        worst_early = max(worst_early, e[early].max() if early.any() else 0.0)
        worst_impact = max(worst_impact, e[~early & ~late].max() if (~early & ~late).any() else 0.0)
        worst_late = max(worst_late, e[late].max() if late.any() else 0.0)
    assert (first >= 0).all()
    g = env.all_states.cpu().numpy().astype(np.float64)
    dz = np.abs(g[:, 3, 2] - st[:, 3, 2])
    dang = np.abs(g[:, 1, :2] - st[:, 1, :2]).max(1)
    rest = (np.abs(st[:, 2]).max(1) < 1e-3) & (np.abs(st[:, 0]).max(1) < 1e-3)  # lanes the ORACLE has brought to rest
    print(f"landing {drone}/{model}: worst in free fall {worst_early:.2e}, through the impact transient {worst_impact:.2e}, "
          f"{settle}+ steps after the first touch {worst_late:.2e} (beyond 1e-4: {int((~ok_late).sum())} lanes); at rest in the oracle {int(rest.sum())}/{n}: "
          f"final |dz| max {dz[rest].max() if rest.any() else 0:.2e}, |d roll,pitch| max {dang[rest].max() if rest.any() else 0:.2e}")
    assert ok_early.all()                # free fall: 1e-4 for every lane
    assert worst_impact < impact_tol     # the impact transient: the fp32 sensitivity of the non-smooth model (fp32_contact_sensitivity.py)
    if strict_late:
        assert ok_late.all()             # and the quad is back inside 1e-4 once the transient is over
    assert rest.mean() >= min_rest
    if rest.any():
        # same resting pose wherever the oracle has come to rest
        assert dz[rest].max() < 5e-4 and dang[rest].max() < 1e-3
        gr = g[rest]
        assert np.abs(gr[:, 2]).max() < 5e-2 and np.abs(gr[:, 0]).max() < 5e-2  # and the device is (all but) at rest there too
    env.disconnect()


@pytest.mark.parametrize("world", [
    dict(),                                                      # the defaults
    dict(contact_report_distance=0.02),                          # a pair reported 2 cm before it touches
    dict(contact_margin=0.02, contact_manifold_points=8, contact_iters=10, contact_residual_threshold=0.0, contact_slop=1e-3),  # round 3's model
    dict(contact_break_distance=0.005, contact_iters=20),
])
def test_contact_options_reach_the_kernels(world):
    """Every doubtful fact of the contact model is a named parameter (include/pyflyt_amd.h at pf_params.contact_response, SURVEY
    section 8(c)): non-default values reach the device kernels and the oracle alike -- tilted quads dropped with the motors off,
    the step of the first contact REPORT (core/aviary.py:523-525), the free fall before it and the resting pose after it."""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine
    from pyflyt_amd.params import quat_from_euler

    n, steps, seed = 96, 200, 5
    rng = np.random.default_rng(seed)
    start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(0.2, 0.4, size=(n, 1))], axis=1).astype(np.float32)
    start_orn = np.concatenate([rng.uniform(-0.5, 0.5, size=(n, 2)), rng.uniform(-3, 3, size=(n, 1))], axis=1)
    P = build_params("quadx", "none", noise="off", autoreset="off", seed=seed, world_options=world or None)
    for k, v in world.items():
        assert abs(float(getattr(P, k)) - float(v)) < 1e-9 * max(1.0, abs(float(v))) + 1e-12, k  # (fp32 field)
    eng = BatchEngine(P, n, device="cuda:0")
    pose = torch.tensor(np.concatenate([start_pos, np.stack([quat_from_euler(o) for o in start_orn])], axis=1), dtype=torch.float32, device="cuda:0").contiguous()
    eng.aviary_reset(pose)
    sp = torch.zeros(n, 4, device="cuda:0")
    eng.aviary_set_mode(-1, sp)
    sp.zero_()
    lib = O.lib()
    Ps, Ls = [], []
    over = {"world_" + k: v for k, v in world.items()}
    for i in range(n):
        Pi = O.make_params("quadx", noise_mode=O.NOISE_OFF, seed=seed, start_pos=start_pos[i].astype(np.float64), start_rpy=start_orn[i], **over)
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(Pi), C.byref(L), i)
        lib.orc_set_mode(C.byref(Pi), C.byref(L), -1)
        for j in range(8):
            L.setpoint[j] = 0.0
        Ps.append(Pi); Ls.append(L)
    first_g, first_o = np.full(n, -1), np.full(n, -1)
    gap_o = np.full(n, np.nan)
    worst_fall = 0.0
    agree = total = 0
    for k in range(steps):
        st_g, _ = eng.aviary_step(sp)
        cg = eng.out_contact.cpu().numpy().astype(bool)  # contact_array after this Aviary step
        low_before = np.array([L.p[2] for L in Ls])
        for Pi, L in zip(Ps, Ls):
            lib.orc_aviary_step(C.byref(Pi), C.byref(L), None, 0, 0)
        co = np.array([bool(L.contact_step) for L in Ls])
        st = np.array([list(L.w_b) + list(L.rpy) + list(L.v_b) + list(L.p) for L in Ls]).reshape(n, 4, 3)
        g = st_g.cpu().numpy().astype(np.float64).reshape(n, 4, 3)
        new_o = (first_o < 0) & co
        first_o[new_o] = k
        first_g[(first_g < 0) & cg] = k
        falling = (first_o < 0) & (first_g < 0)
        if falling.any():
            e = np.abs(g - st)[falling] / np.maximum(1.0, np.linalg.norm(st, axis=2, keepdims=True))[falling]
            worst_fall = max(worst_fall, float(e.max()))
        agree += int((cg == co).sum()); total += n
    assert (first_o >= 0).all() and (first_g >= 0).all()
    assert np.abs(first_g - first_o).max() <= 1, (first_g - first_o)       # the same step, up to a pose one ulp either side of the threshold
    assert (first_g == first_o).mean() > 0.95
    # (with a speculative margin the rows act before the pair is reported: "falling" then includes the first constrained ticks)
    assert worst_fall < (1e-4 if world.get("contact_margin", 0.0) == 0.0 else 2e-3)
    assert agree / total > 0.97, agree / total
    g = eng.aviary_step(sp)[0].cpu().numpy().astype(np.float64).reshape(n, 4, 3)
    for Pi, L in zip(Ps, Ls):
        lib.orc_aviary_step(C.byref(Pi), C.byref(L), None, 0, 0)
    zo = np.array([L.p[2] for L in Ls])
    rest = np.array([max(abs(x) for x in list(L.v) + list(L.w)) < 1e-3 for L in Ls])
    assert rest.mean() > 0.9
    assert np.abs(g[rest, 3, 2] - zo[rest]).max() < 5e-4
    slop = world.get("contact_slop", 1e-5)
    assert np.abs(zo[rest] - (0.01 - slop)).max() < 1e-4                    # the resting overlap IS the slop parameter
    if world.get("contact_report_distance", 0.0) > 0.0:                      # reported before it touches: earlier than with the defaults
        Pd = O.make_params("quadx", noise_mode=O.NOISE_OFF, seed=seed, start_pos=start_pos[0].astype(np.float64), start_rpy=start_orn[0])
        Ld = O.Lane()
        lib.orc_aviary_reset(C.byref(Pd), C.byref(Ld), 0)
        lib.orc_set_mode(C.byref(Pd), C.byref(Ld), -1)
        for j in range(8):
            Ld.setpoint[j] = 0.0
        k0 = 0
        while not Ld.contact_step:
            lib.orc_aviary_step(C.byref(Pd), C.byref(Ld), None, 0, 0)
            k0 += 1
        assert first_o[0] < k0 - 1 or first_o[0] == 0
