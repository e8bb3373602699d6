"""GPU parity: the HIP path (through the C ABI) against the fp64 oracle on identical initial
states, action sequences and noise. Tolerance (fp32 kernel vs fp64 oracle, stated per north_star):
|gpu - ref| <= 1e-4 * max(1, |ref|) for every observation element of every lane whose
termination history agrees; lanes that flip a termination threshold by rounding are counted and
must stay below 0.5 % of the batch."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _engine(vehicle, task, n, **kw):
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    P = build_params(vehicle, task, **kw)
    return BatchEngine(P, n, device="cuda:0")


def _oracle(env, n, noise, seed=0, **over):
    mode = {"off": O.NOISE_OFF, "inject": O.NOISE_INJECT, "philox": O.NOISE_PHILOX}[noise]
    return O.OracleBatch(O.make_params(env, noise_mode=mode, seed=seed, **over), n)


def relerr(a, ref):
    return np.abs(a - ref) / np.maximum(1.0, np.abs(ref))


def sample_actions(rng, n, low, high):
    return rng.uniform(low, high, size=(n, 4)).astype(np.float32)


QUAD_LOW, QUAD_HIGH = np.array([-np.pi] * 3 + [0.0]), np.array([np.pi] * 3 + [0.8])
FW_LOW, FW_HIGH = -np.ones(4), np.ones(4)


def run_env_parity(vehicle, task, env_name, n, steps, noise, autoreset, low, high, seed=0, gentle=None, **over):
    eng = _engine(vehicle, task, n, noise=noise, autoreset=autoreset, seed=seed,
                  **{k: v for k, v in over.items() if k in ("goal_reach_distance", "max_duration_seconds", "angle_representation", "sparse_reward")})
    oover = {}
    if "goal_reach_distance" in over:
        oover["goal_reach_distance"] = over["goal_reach_distance"]
    if "max_duration_seconds" in over:
        oover["max_steps"] = int(over["max_duration_seconds"] * (40 if task == "hover" else 30))
    if over.get("angle_representation") == "euler":
        oover["angle_repr"] = 0
    if over.get("sparse_reward"):
        oover["sparse_reward"] = 1
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
        ut = np.concatenate([rng.uniform(0, 2 * np.pi, size=(n, 2 * nt)), rng.uniform(1.0, eng.params.dome * 0.9, size=(n, nt))], axis=1) if nt else None
        dev = lambda a: None if a is None else torch.tensor(np.ascontiguousarray(a.T), dtype=torch.float32, device="cuda:0")  # noqa: E731
        # the oracle sees exactly the fp32-rounded draws the device sees
        f = lambda a: None if a is None else np.ascontiguousarray(a.astype(np.float32).astype(np.float64))  # noqa: E731
        return f(xi), f(xr), f(ut), dev(xi), dev(xr), dev(ut)

    xi, xr, ut, dxi, dxr, dut = draws()
    obs_g = eng.env_reset(xi_reset=dxr, u_targets=dut).cpu().numpy().astype(np.float64)
    obs_r = orc.reset(xi_reset=xr, u_targets=ut)
    assert relerr(obs_g, obs_r).max() < RTOL, relerr(obs_g, obs_r).max()
    ok = np.ones(n, dtype=bool)  # lanes whose done-history still agrees
    worst = 0.0
    n_done = 0
    amode = {"off": 0, "next_step": 1, "same_step": 2}[autoreset]
    for k in range(steps):
        a = sample_actions(rng, n, low, high) if gentle is None else gentle(rng, n)
        xi, xr, ut, dxi, dxr, dut = draws()
        og, rg, tg, trg = eng.env_step(torch.tensor(a, device="cuda:0"), xi=dxi, xi_reset=dxr, u_targets=dut)
        og, rg = og.cpu().numpy().astype(np.float64), rg.cpu().numpy().astype(np.float64)
        tg, trg = tg.cpu().numpy(), trg.cpu().numpy()
        orr, rr, tr, trr, fin = orc.step(a, xi=xi, xi_reset=xr, u_targets=ut, autoreset=amode)
        agree = (tg == tr) & (trg == trr)
        ok &= agree
        e = relerr(og[ok], orr[ok])
        if e.size:
            worst = max(worst, e.max())
            assert e.max() < RTOL, (k, e.max(), np.unravel_index(np.argmax(e), e.shape))
            er = np.abs(rg[ok] - rr[ok]) / np.maximum(1.0, np.abs(rr[ok]))
            assert er.max() < 1e-3, (k, er.max())
        n_done += int((tr | trr)[ok].sum())
        if autoreset == "same_step" and eng.final_obs is not None:
            d = ok & (tr | trr)
            if d.any():
                ef = relerr(eng.final_obs.cpu().numpy().astype(np.float64)[d], fin[d])
                assert ef.max() < RTOL, (k, ef.max())
        if autoreset == "off":
            # freeze finished lanes on both sides: reset them together
            done = (tr | trr | tg | trg)
            if done.any():
                m = torch.tensor(done, device="cuda:0")
                xi, xr, ut, dxi, dxr, dut = draws()
                obs_g = eng.env_reset(mask=m, xi_reset=dxr, u_targets=dut).cpu().numpy().astype(np.float64)
                obs_r = orc.reset(mask=done, xi_reset=xr, u_targets=ut)
                assert relerr(obs_g[ok & done], obs_r[ok & done]).max() < RTOL
    frac_bad = 1.0 - ok.mean()
    print(f"{vehicle}/{task} noise={noise} autoreset={autoreset}: worst rel err {worst:.2e}, diverged lanes {frac_bad:.4f}, episodes ended {n_done}")
    assert frac_bad <= 0.005, frac_bad
    return worst, n_done


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

    eng_done = run_env_parity("quadx", "hover", "hover", 256, 60, "philox", "next_step", QUAD_LOW, QUAD_HIGH, seed=9, gentle=low)
    assert eng_done[1] > 200


@pytest.mark.parametrize("noise,autoreset", [("inject", "off"), ("philox", "next_step"), ("philox", "same_step")])
def test_quadx_waypoints_parity(noise, autoreset):
    run_env_parity("quadx", "waypoints", "quadx_waypoints", 1024, 120, noise, autoreset, QUAD_LOW, QUAD_HIGH, seed=11)


def test_quadx_waypoints_reach():
    def gentle(rng, n):
        return np.concatenate([rng.uniform(-0.3, 0.3, size=(n, 3)), rng.uniform(0.33, 0.40, size=(n, 1))], axis=1).astype(np.float32)

    run_env_parity("quadx", "waypoints", "quadx_waypoints", 512, 150, "philox", "next_step", QUAD_LOW, QUAD_HIGH, seed=13,
                   gentle=gentle, goal_reach_distance=2.5)


@pytest.mark.parametrize("noise,autoreset", [("inject", "off"), ("philox", "next_step"), ("philox", "same_step")])
def test_fixedwing_waypoints_parity(noise, autoreset):
    run_env_parity("fixedwing", "waypoints", "fixedwing_waypoints", 1024, 150, noise, autoreset, FW_LOW, FW_HIGH, seed=17)


def test_fixedwing_waypoints_reach():
    def gentle(rng, n):
        return np.concatenate([rng.uniform(-0.3, 0.3, size=(n, 3)), rng.uniform(-0.2, 0.8, size=(n, 1))], axis=1).astype(np.float32)

    run_env_parity("fixedwing", "waypoints", "fixedwing_waypoints", 512, 300, "philox", "next_step", FW_LOW, FW_HIGH, seed=19,
                   gentle=gentle, goal_reach_distance=40.0)
