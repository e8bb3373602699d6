"""GPU parity: the HIP path (through the C ABI) against the fp64 oracle on identical initial
states, action sequences and noise.

Tolerance (fp32 kernel vs fp64 oracle, per north_star): |gpu - ref| <= RTOL * max(1, ||ref_vec||)
with RTOL = 1e-4 for every observation element, every step, where ref_vec is the physical vector
the element belongs to (angular velocity, attitude, velocity, position, action, actuators, each
body-frame target delta).

Drop policy (strict):
  * QuadX runs: NO lane may leave the comparison -- every lane must hold the bound and report identical
    terminated / truncated flags at every step (`max_bad` = 0).
  * One event is counted separately instead of dropping the lane: both sides end an episode in the same env step
    with identical flags, but the terminal observation differs because the crossing moved over an INNER
    Aviary-step boundary (the reference breaks out of its inner loop there). Both sides re-initialise the lane
    next and agree again. Bound: <= max(2, 0.2 % of the episodes ended) (measured: 1 of 115 666 and 1 of 1 108).
  * Fixedwing runs (120 s episodes at 20 m/s towards 100 m domes: thousands of steps in which a waypoint-reach
    or dome crossing can fall within fp32 rounding of a step boundary): a lane may leave the comparison ONLY
    with a classified cause -- the two sides must show the same discrete event (episode end, target reached)
    at DIFFERENT steps within +-1 step of the first violation, i.e. fp32 rounding moved a threshold crossing
    over a step boundary, after which the trajectories are no longer comparable point-wise. A lane that
    diverges without such an event flip (e.g. a wrong stall branch) fails the test. The classified fraction
    must still stay below 0.5 %."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

RTOL = 1e-4
# The terminal observation of an episode that ends ON THE FLOOR carries the impact impulse of the contact solver (Gauss-Seidel
# sweeps with clamps at zero normal impulse and at the friction cone): a float32 build of the ORACLE is up to 4e-4 (quad) away from
# the fp64 one on that one observation and back to 4e-6 a few steps later (tests/tools/fp32_contact_sensitivity.py) -- a property
# of the non-smooth model in fp32, not of the kernels. Round 5 measured what each test needs (every run prints its worst such
# observation): 1.1e-5 ... 5.1e-5 under the action space's own draws -- the bound went from 2e-3 to 1e-3 --, and 1.1e-3 in the one
# test that flies every episode into the floor under low thrust (test_floor_endings: its own bound, with a lower bound).
RTOL_IMPACT = 1e-3


def _engine(vehicle, task, n, **kw):
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    P = build_params(vehicle, task, **kw)
    return BatchEngine(P, n, device="cuda:0")


def _oracle(env, n, noise, seed=0, **over):
    mode = {"off": O.NOISE_OFF, "inject": O.NOISE_INJECT, "philox": O.NOISE_PHILOX}[noise]
    return O.OracleBatch(O.make_params(env, noise_mode=mode, seed=seed, **over), n)


def obs_groups(D, quat, aux, nt, tw=3):
    """Index groups of the flattened observation: each physical vector is one group (a target delta with its
    yaw error, tw = 4, is one group: the yaw error is then held to 1e-4 * max(1, distance) rad)."""
    g, k = [], 0
    for w in (3, 4 if quat else 3, 3, 3, 4, aux) + (tw,) * nt:
        g.append((k, k + w))
        k += w
    assert k == D, (k, D)
    return g


def relerr(a, ref, groups=None):
    """|a - ref| / max(1, ||ref_vector||): relative error with each physical vector (angular velocity,
    attitude, velocity, position, ..., each target delta) normalised by its own magnitude."""
    if groups is None:
        return np.abs(a - ref) / np.maximum(1.0, np.abs(ref))
    out = np.zeros_like(ref)
    for lo, hi in groups:
        scale = np.maximum(1.0, np.linalg.norm(ref[..., lo:hi], axis=-1, keepdims=True))
        out[..., lo:hi] = np.abs(a[..., lo:hi] - ref[..., lo:hi]) / scale
    return out


def sample_actions(rng, n, low, high):
    return rng.uniform(low, high, size=(n, 4)).astype(np.float32)


QUAD_LOW, QUAD_HIGH = np.array([-np.pi] * 3 + [0.0]), np.array([np.pi] * 3 + [0.8])
FW_LOW, FW_HIGH = -np.ones(4), np.ones(4)


def run_env_parity(vehicle, task, env_name, n, steps, noise, autoreset, low, high, seed=0, gentle=None, max_bad=None, rtol_impact=None, corrupt=None, diagnose=False, **over):
    """corrupt = (step, lane, group, word, delta): add delta to one word of the DEVICE's state before that step (the harness's own negative
    test); diagnose: return (worst, episodes ended, dropped lanes, step of each lane's first violation) instead of asserting on them."""
    RTOL_IMPACT = rtol_impact if rtol_impact is not None else globals()["RTOL_IMPACT"]
    if max_bad is None:
        max_bad = 0.0 if vehicle == "quadx" else 0.005
    eng = _engine(vehicle, task, n, noise=noise, autoreset=autoreset, seed=seed,
                  **{k: v for k, v in over.items() if k in ("goal_reach_distance", "max_duration_seconds", "angle_representation", "sparse_reward",
                                                             "agent_hz", "num_targets", "flight_dome_size", "use_yaw_targets", "goal_reach_angle")})
    oover = {}
    if "goal_reach_distance" in over:
        oover["goal_reach_distance"] = over["goal_reach_distance"]
    hz = over.get("agent_hz", 40 if task == "hover" else 30)
    if "max_duration_seconds" in over or "agent_hz" in over:
        oover["max_steps"] = int(over.get("max_duration_seconds", 10.0 if task == "hover" else (10.0 if vehicle == "quadx" else 120.0)) * hz)
    if "agent_hz" in over:
        oover["env_step_ratio"] = 120 // hz
    if "num_targets" in over:
        oover["num_targets"] = over["num_targets"]
    if "flight_dome_size" in over:
        oover["dome"] = over["flight_dome_size"]
    if over.get("angle_representation") == "euler":
        oover["angle_repr"] = 0
    if over.get("sparse_reward"):
        oover["sparse_reward"] = 1
    if over.get("use_yaw_targets"):
        oover["use_yaw_targets"] = 1
        oover["goal_reach_angle"] = over.get("goal_reach_angle", 0.1)
    yaw = bool(over.get("use_yaw_targets"))
    orc = _oracle(env_name, n, noise, seed=seed, **oover)
    rng = np.random.default_rng(seed + 1)
    T, TR = eng.ticks_per_step, eng.settle_ticks
    nm = eng.params.n_motors
    nt = eng.params.num_targets

    def draws():
        if noise != "inject":
            return None, None, None, None, None, None
        xi = rng.normal(nm, 1.0, size=(n, T))
        xr = rng.normal(nm, 1.0, size=(n, TR))
        ut = np.concatenate([rng.uniform(0, 2 * np.pi, size=(n, 2 * nt)), rng.uniform(1.0, eng.params.dome * 0.9, size=(n, nt))] +
                            ([rng.uniform(-np.pi, np.pi, size=(n, nt))] if yaw else []), axis=1) if nt else None
        dev = lambda a: None if a is None else torch.tensor(np.ascontiguousarray(a.T), dtype=torch.float32, device="cuda:0")  # noqa: E731
        # the oracle sees exactly the fp32-rounded draws the device sees
        f = lambda a: None if a is None else np.ascontiguousarray(a.astype(np.float32).astype(np.float64))  # noqa: E731
        return f(xi), f(xr), f(ut), dev(xi), dev(xr), dev(ut)

    G = obs_groups(eng.obs_dim, bool(eng.params.angle_repr), 4 if vehicle == "quadx" else 6, nt, 4 if yaw else 3)
    xi, xr, ut, dxi, dxr, dut = draws()
    obs_g = eng.env_reset(xi_reset=dxr, u_targets=dut).cpu().numpy().astype(np.float64)
    obs_r = orc.reset(xi_reset=xr, u_targets=ut)
    assert relerr(obs_g, obs_r, G).max() < RTOL, relerr(obs_g, obs_r, G).max()
    ok = np.ones(n, dtype=bool)  # lanes still comparable point-wise
    n_term_mis = 0               # terminal observations that differ although both sides ended the episode in the same step
    first_bad = np.full(n, -1)   # step of the first violation
    ev_g, ev_r = [], []          # per step: (episode ended, targets left) on each side -- the discrete events
    worst = 0.0
    worst_impact = 0.0           # the largest error among the observations held to RTOL_IMPACT (printed: the bound stays tied to it)
    n_done = 0
    lane_steps = 0
    n_final = n_final_bad = 0
    amode = {"off": 0, "next_step": 1, "same_step": 2}[autoreset]
    for k in range(steps):
        a = sample_actions(rng, n, low, high) if gentle is None else gentle(rng, n)
        xi, xr, ut, dxi, dxr, dut = draws()
        if corrupt is not None and corrupt[0] == k:
            eng.state[corrupt[2], corrupt[1], corrupt[3]] += corrupt[4]
        og, rg, tg, trg = eng.env_step(torch.tensor(a, device="cuda:0"), xi=dxi, xi_reset=dxr, u_targets=dut)
        og, rg = og.cpu().numpy().astype(np.float64), rg.cpu().numpy().astype(np.float64)
        tg, trg = tg.cpu().numpy(), trg.cpu().numpy()
        orr, rr, tr, trr, fin = orc.step(a, xi=xi, xi_reset=xr, u_targets=ut, autoreset=amode)
        e = relerr(og, orr, G).max(axis=1)
        er = np.abs(rg - rr) / np.maximum(1.0, np.abs(rr))
        from pyflyt_amd import _lib as PL

        # an observation that carries the floor's impulses: the step that reports the collision on both sides, or a lane
        # within reach of the floor (the speculative contact constraint stops the fall one tick before the report)
        zlow = orc.field("p")[:, 2] - float(orc.P.bound_radius)
        impact = (((eng.flags().cpu().numpy() & PL.F_INFO_COLLISION) != 0) & (orc.field("info_collision") != 0) & tg & tr) | (zlow < 0.05)
        if autoreset == "same_step":
            impact = np.zeros(n, dtype=bool)  # (the lane was re-initialised inside the step; its terminal observation is checked through final_obs below)
        good = (tg == tr) & (trg == trr) & (e < np.where(impact, RTOL_IMPACT, RTOL)) & (er < 1e-3)
        if (impact & ok & good).any():
            worst_impact = max(worst_impact, float(e[impact & ok & good].max()))
        e = np.where(impact, 0.0, e)  # (held to RTOL_IMPACT above, not part of `worst`)
        # Both sides END the episode in this env step with identical flags, but at different INNER Aviary steps
        # (quadx_base_env.py:289-290 breaks out of the inner loop once terminated): fp32 rounding moved a dome / floor /
        # reach crossing over an inner-step boundary. Only this terminal observation (and its reward) shows it; both
        # sides re-initialise the lane next and agree again, so the lane STAYS in the comparison and the event is
        # counted (bounded below) instead of being tolerated silently.
        term_mis = ok & ~good & (tg == tr) & (trg == trr) & (tr | trr)
        n_term_mis += int(term_mis.sum())
        good |= term_mis
        e = np.where(term_mis, 0.0, e)  # (counted above, not part of `worst`)
        # (restored in round 5: these two lines had been lost with the change that introduced term_mis, and from then on no lane
        #  could leave the comparison -- `dropped lanes` read 0 whatever happened, only the callers that assert on `worst` noticed)
        first_bad[ok & ~good] = k
        ok &= good
        nl_g = eng.ints()[:, 3].cpu().numpy().copy() if nt else np.zeros(n, dtype=np.int32)
        nl_r = orc.field("n_targets_left") if nt else np.zeros(n, dtype=np.int32)
        ev_g.append(np.stack([(tg | trg).astype(np.int32), nl_g], axis=1))
        ev_r.append(np.stack([(tr | trr).astype(np.int32), nl_r], axis=1))
        if ok.any():
            worst = max(worst, e[ok].max())
        lane_steps += int(ok.sum())
        n_done += int((tr | trr)[ok].sum())
        if autoreset == "same_step" and eng.final_obs is not None:
            d = ok & (tr | trr)
            if d.any():
                # The terminal observation. Both sides can end an episode in the same env step but at
                # different INNER Aviary steps when fp32 rounding moves a dome / contact / reach crossing
                # over an inner-step boundary: flags and the (reset) observation still agree, only this
                # vector shows it. Counted like the dropped lanes (same bound), not tolerated silently.
                ef = relerr(eng.final_obs.cpu().numpy().astype(np.float64)[d], fin[d], G).max(axis=1)
                hit = (eng.final_info[:, 0].cpu().numpy()[d] & PL.F_INFO_COLLISION) != 0  # floor impact: RTOL_IMPACT (see the top of the file)
                n_final += int(d.sum())
                n_final_bad += int((ef >= np.where(hit, RTOL_IMPACT, RTOL)).sum())
                if (hit & (ef < RTOL_IMPACT)).any():
                    worst_impact = max(worst_impact, float(ef[hit & (ef < RTOL_IMPACT)].max()))
        if autoreset == "off":
            # finished lanes are reset together on both sides
            done = (tr | trr | tg | trg)
            if done.any():
                m = torch.tensor(done, device="cuda:0")
                xi, xr, ut, dxi, dxr, dut = draws()
                obs_g = eng.env_reset(mask=m, xi_reset=dxr, u_targets=dut).cpu().numpy().astype(np.float64)
                obs_r = orc.reset(mask=done, xi_reset=xr, u_targets=ut)
                sel = ok & done
                if sel.any():
                    assert relerr(obs_g[sel], obs_r[sel], G).max() < RTOL
    frac_bad = 1.0 - ok.mean()
    # classify every dropped lane: the same discrete event on both sides at different steps within +-1 step
    ev_g, ev_r = np.stack(ev_g), np.stack(ev_r)  # [steps, n, 2]
    unexplained = []
    for i in np.nonzero(~ok)[0]:
        k0 = int(first_bad[i])
        lo, hi = max(0, k0 - 1), min(steps, k0 + 2)
        flipped = (ev_g[lo:hi, i] != ev_r[lo:hi, i]).any()
        if not flipped:
            unexplained.append((int(i), k0))
    print(f"{vehicle}/{task} noise={noise} autoreset={autoreset}: worst rel err {worst:.2e} over {lane_steps} lane-steps, "
          f"dropped lanes {frac_bad:.4f} ({int((~ok).sum())} of {n}; unclassified: {len(unexplained)}), "
          f"terminal observations off by an inner step {n_term_mis} of {n_done} episodes ended; worst observation with a floor impact in it {worst_impact:.2e}")
    if diagnose:
        return worst, n_done, np.nonzero(~ok)[0], first_bad
    assert not unexplained, f"lanes left the comparison without a discrete-event flip: {unexplained[:8]}"
    assert frac_bad <= max_bad, frac_bad
    assert n_term_mis <= max(2, int(2e-3 * n_done)), (n_term_mis, n_done)
    assert n_final_bad <= max(1, int(max_bad * n_final)), (n_final_bad, n_final)
    if rtol_impact is not None:  # a bound of its own stays tied to what is measured: within a factor of ten of it
        assert worst_impact > rtol_impact / 10.0, (worst_impact, rtol_impact)
    return worst, n_done


@pytest.mark.parametrize("vehicle,task,env_name,over", [
    ("quadx", "hover", "hover", dict(agent_hz=30)),                       # 4 Aviary steps per env step
    ("quadx", "hover", "hover", dict(agent_hz=60, flight_dome_size=1.5)),  # 2, small dome: many OOB endings
    ("quadx", "hover", "hover", dict(agent_hz=120, sparse_reward=True)),   # 1
    ("quadx", "waypoints", "quadx_waypoints", dict(num_targets=1, goal_reach_distance=1.5)),
    ("quadx", "waypoints", "quadx_waypoints", dict(num_targets=3, agent_hz=40, angle_representation="euler")),
    ("fixedwing", "waypoints", "fixedwing_waypoints", dict(agent_hz=60, num_targets=2)),
    ("fixedwing", "waypoints", "fixedwing_waypoints", dict(agent_hz=40, flight_dome_size=60.0, sparse_reward=True)),
])
def test_env_constructor_knobs(vehicle, task, env_name, over):
    """The reference envs' constructor arguments that change the kernel's loop structure or observation
    width (agent_hz -> Aviary steps per env step, num_targets -> observation width, dome, reward style)."""
    low, high = (QUAD_LOW, QUAD_HIGH) if vehicle == "quadx" else (FW_LOW, FW_HIGH)
    worst, n_done = run_env_parity(vehicle, task, env_name, 512, 100, "philox", "next_step", low, high, seed=41, **over)
    assert worst < RTOL


def test_smoke_entry():
    import __graft_entry__ as g

    g.smoke()


@pytest.mark.parametrize("noise,autoreset", [("off", "off"), ("inject", "off"), ("philox", "next_step"), ("philox", "same_step")])
def test_hover_parity(noise, autoreset):
    worst, n_done = run_env_parity("quadx", "hover", "hover", 1024, 150, noise, autoreset, QUAD_LOW, QUAD_HIGH, seed=3)
    assert n_done > 100  # random actions end episodes quickly: resets are exercised


def test_hover_truncation_and_euler():
    def gentle(rng, n):
        return np.concatenate([rng.uniform(-0.3, 0.3, size=(n, 3)), rng.uniform(0.33, 0.40, size=(n, 1))], axis=1).astype(np.float32)

    run_env_parity("quadx", "hover", "hover", 256, 60, "philox", "next_step", QUAD_LOW, QUAD_HIGH, seed=5, gentle=gentle,
                   max_duration_seconds=0.5, angle_representation="euler")


def test_hover_floor_contact():
    def low(rng, n):
        return np.concatenate([rng.uniform(-0.5, 0.5, size=(n, 3)), rng.uniform(0.0, 0.25, size=(n, 1))], axis=1).astype(np.float32)

    # (every episode ends on the floor, 768 of them, motors running: measured worst terminal observation 1.1e-3)
    eng_done = run_env_parity("quadx", "hover", "hover", 256, 60, "philox", "next_step", QUAD_LOW, QUAD_HIGH, seed=9, gentle=low, rtol_impact=3e-3)
    assert eng_done[1] > 200


@pytest.mark.parametrize("noise,autoreset", [("inject", "off"), ("philox", "next_step"), ("philox", "same_step")])
def test_quadx_waypoints_parity(noise, autoreset):
    run_env_parity("quadx", "waypoints", "quadx_waypoints", 1024, 120, noise, autoreset, QUAD_LOW, QUAD_HIGH, seed=11)


def test_quadx_waypoints_reach():
    def gentle(rng, n):
        return np.concatenate([rng.uniform(-0.3, 0.3, size=(n, 3)), rng.uniform(0.33, 0.40, size=(n, 1))], axis=1).astype(np.float32)

    run_env_parity("quadx", "waypoints", "quadx_waypoints", 512, 150, "philox", "next_step", QUAD_LOW, QUAD_HIGH, seed=13,
                   gentle=gentle, goal_reach_distance=2.5)


@pytest.mark.parametrize("kernel", ["specialised", "generic"])
@pytest.mark.parametrize("noise,autoreset", [("inject", "off"), ("philox", "next_step"), ("philox", "same_step")])
def test_quadx_waypoints_yaw_targets_parity(monkeypatch, noise, autoreset, kernel):
    """use_yaw_targets=True (quadx_waypoints_env.py:40-42; waypoint_handler.py:85-89,144-156,167-179): 4-wide target
    deltas, the extra reset draws, the reach gate on the yaw error. Gentle flight with wide gates so that targets are
    reached (and refused on the yaw gate alone)."""
    if kernel == "generic":
        monkeypatch.setenv("PF_DISABLE_FAST", "1")

    def gentle(rng, n):
        return np.concatenate([rng.uniform(-0.3, 0.3, size=(n, 2)), rng.uniform(-1.5, 1.5, size=(n, 1)), rng.uniform(0.33, 0.40, size=(n, 1))], axis=1).astype(np.float32)

    run_env_parity("quadx", "waypoints", "quadx_waypoints", 512, 150, noise, autoreset, QUAD_LOW, QUAD_HIGH, seed=29, gentle=gentle,
                   use_yaw_targets=True, goal_reach_distance=2.5, goal_reach_angle=1.2)
    run_env_parity("quadx", "waypoints", "quadx_waypoints", 512, 60, noise, autoreset, QUAD_LOW, QUAD_HIGH, seed=31, use_yaw_targets=True)


@pytest.mark.parametrize("noise,autoreset", [("inject", "off"), ("philox", "next_step"), ("philox", "same_step")])
def test_fixedwing_waypoints_parity(noise, autoreset):
    run_env_parity("fixedwing", "waypoints", "fixedwing_waypoints", 1024, 150, noise, autoreset, FW_LOW, FW_HIGH, seed=17)


def test_fixedwing_waypoints_reach():
    def gentle(rng, n):
        return np.concatenate([rng.uniform(-0.3, 0.3, size=(n, 3)), rng.uniform(-0.2, 0.8, size=(n, 1))], axis=1).astype(np.float32)

    run_env_parity("fixedwing", "waypoints", "fixedwing_waypoints", 512, 300, "philox", "next_step", FW_LOW, FW_HIGH, seed=19,
                   gentle=gentle, goal_reach_distance=40.0)


@pytest.mark.parametrize("vehicle,task,env_name,low,high", [
    ("quadx", "hover", "hover", QUAD_LOW, QUAD_HIGH),
    ("quadx", "waypoints", "quadx_waypoints", QUAD_LOW, QUAD_HIGH),
    ("fixedwing", "waypoints", "fixedwing_waypoints", FW_LOW, FW_HIGH),
])
def test_generic_kernel_parity(monkeypatch, vehicle, task, env_name, low, high):
    """The specialised kernels (quadx_fast.hpp, fixedwing_fast.hpp) are selected from the parameter
    block; configurations outside their envelope run the generic env_kernel. PF_DISABLE_FAST forces
    that path for the reference configurations so that it stays covered by the same oracle."""
    monkeypatch.setenv("PF_DISABLE_FAST", "1")
    run_env_parity(vehicle, task, env_name, 512, 100, "philox", "next_step", low, high, seed=37)


def test_the_parity_harness_drops_exactly_the_lane_that_is_wrong():
    """The harness's own negative test (round 5 found that `run_env_parity` had not been able to drop a lane for three rounds): one
    word of ONE lane's device state is moved by 1e-3 before one step -- the harness reports exactly that lane as dropped, at that
    step, keeps every other lane in the comparison, and a run with the default policy (no lane may leave) fails."""
    lane, step = 37, 6
    corrupt = (step, lane, 0, 0, 1e-3)  # position x, ten times the tolerance
    worst, n_done, dropped, first_bad = run_env_parity("quadx", "hover", "hover", 256, 40, "philox", "next_step", QUAD_LOW, QUAD_HIGH, seed=3,
                                                       corrupt=corrupt, diagnose=True)
    assert dropped.tolist() == [lane] and first_bad[lane] == step and worst < RTOL
    with pytest.raises(AssertionError):
        run_env_parity("quadx", "hover", "hover", 256, 40, "philox", "next_step", QUAD_LOW, QUAD_HIGH, seed=3, corrupt=corrupt)
    # the same run untouched: nothing dropped
    worst, n_done, dropped, _ = run_env_parity("quadx", "hover", "hover", 256, 40, "philox", "next_step", QUAD_LOW, QUAD_HIGH, seed=3, diagnose=True)
    assert dropped.size == 0 and worst < RTOL
