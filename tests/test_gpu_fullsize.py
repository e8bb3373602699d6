"""GPU parity at BASELINE.json's sizes and horizon.

* north_star: "within 1e-4 rel state error ... over 1 000 steps" -- config[1] (QuadX-Hover, batch
  4096) is stepped 1 000 env steps (6 000 physics ticks) against the fp64 oracle with the motor noise
  on and NEXT_STEP auto-reset; half of the lanes fly gentle actions so that whole 400-step episodes
  (and their truncation) are covered, the other half fly the benchmark's uniformly random actions.
* configs[2..4] batch sizes (65 536 per GPU, 524 288 for the 8-GPU config's total): the oracle cannot
  follow a batch that size in seconds, so full-size runs are checked through properties that do not
  depend on the size: lanes are independent and the RNG is keyed by the global lane index, hence
  (i) any slice of the big batch must match an oracle batch started at that lane offset, and
  (ii) the big batch must be bit-identical to the concatenation of separately run shards.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402

from test_gpu_parity import QUAD_HIGH, QUAD_LOW, RTOL, obs_groups, relerr, run_env_parity  # noqa: E402

pytestmark = pytest.mark.gpu


def test_hover_config2_1000_steps():
    def mixed(rng, n):
        a = rng.uniform(QUAD_LOW, QUAD_HIGH, size=(n, 4))
        g = np.concatenate([rng.uniform(-0.3, 0.3, size=(n, 3)), rng.uniform(0.33, 0.40, size=(n, 1))], axis=1)
        a[: n // 2] = g[: n // 2]
        return a.astype(np.float32)

    # 4.1 M lane-steps, ~116 000 episodes. A lane may leave the comparison only with a CLASSIFIED discrete-event flip -- both sides end
    # the episode, one env step apart, because fp32 rounding moved a dome crossing over an env-step boundary -- and at most two of
    # the 4 096 may (measured: one, in step 499; round 4's noise realisation had none; run_env_parity asserts the classification)
    worst, n_done = run_env_parity("quadx", "hover", "hover", 4096, 1000, "philox", "next_step", QUAD_LOW, QUAD_HIGH,
                                   seed=23, gentle=mixed, max_bad=2.0 / 4096)
    assert worst < RTOL
    assert n_done > 4096  # every lane went through several episodes


@pytest.mark.parametrize("env_id,name,n", [
    ("PyFlyt/QuadX-Hover-v4", "hover", 65536),
    ("PyFlyt/QuadX-Hover-v4", "hover", 524288),
    ("PyFlyt/QuadX-Waypoints-v4", "quadx_waypoints", 65536),
    ("PyFlyt/Fixedwing-Waypoints-v4", "fixedwing_waypoints", 65536),
])
def test_fullsize_slices_match_oracle(env_id, name, n):
    from pyflyt_amd.gym_envs import make_vec

    seed, steps, w = 29, 40, 320
    kw = {"flatten": False} if "Waypoints" in env_id else {}
    env = make_vec(env_id, n, seed=seed, **kw)
    flat = lambda o: o if torch.is_tensor(o) else torch.cat([o["attitude"], o["target_deltas"].flatten(1)], dim=1)  # noqa: E731
    # first lanes, a window straddling wavefront boundaries mid-batch, the last lanes
    offs = [0, n // 2 - 37, n - w]
    P = O.make_params(name, noise_mode=O.NOISE_PHILOX, seed=seed)
    orcs = [O.OracleBatch(P, w, lane0=o) for o in offs]
    og = flat(env.reset(seed=seed)[0])
    nt = int(P.num_targets) if "waypoints" in name else 0
    G = obs_groups(og.shape[1], True, 4 if "fixedwing" not in name else 6, nt)
    ok = [np.ones(w, dtype=bool) for _ in offs]
    for o, orc in zip(offs, orcs):
        assert relerr(og[o:o + w].cpu().numpy().astype(np.float64), orc.reset(), G).max() < RTOL
    worst = 0.0
    worst_impact = 0.0
    for k in range(steps):
        a = env.sample_actions(k)
        obs, rew, term, trunc, _ = env.step(a)
        obs = flat(obs)
        for j, (o, orc) in enumerate(zip(offs, orcs)):
            ro, rr, rt, ru, _ = orc.step(a[o:o + w].cpu().numpy(), autoreset=1)
            e = relerr(obs[o:o + w].cpu().numpy().astype(np.float64), ro, G).max(axis=1)
            er = np.abs(rew[o:o + w].cpu().numpy() - rr) / np.maximum(1.0, np.abs(rr))
            # an observation within reach of the floor carries the contact solve's impulses: RTOL_IMPACT there (z is entry 12
            # of the attitude block; tests/test_gpu_golden.py, tests/tools/fp32_contact_sensitivity.py), 1e-4 everywhere else
            tol = np.where(ro[:, 12] < 0.12, 1e-3, RTOL) if "fixedwing" not in name else RTOL  # (measured within reach of the floor: 2e-7; the bound 5e-3 until round 5)
            ok[j] &= (term[o:o + w].cpu().numpy() == rt) & (trunc[o:o + w].cpu().numpy() == ru) & (e < tol) & (er < 1e-3)
            low = (ro[:, 12] < 0.12) if "fixedwing" not in name else np.zeros(len(e), dtype=bool)
            if (ok[j] & ~low).any():
                worst = max(worst, e[ok[j] & ~low].max())
            if (ok[j] & low).any():
                worst_impact = max(worst_impact, e[ok[j] & low].max())
    bad = 1.0 - np.concatenate(ok).mean()
    print(f"{name} n={n}: worst rel err {worst:.2e}, dropped {bad:.4f}; worst observation within reach of the floor {worst_impact:.2e}")
    # strict for the quadrotor configurations: no lane may leave the comparison; the aeroplane may lose lanes to classified
    # discrete-event flips within one step (tests/test_gpu_parity.py), at most 0.5 %
    assert bad <= (0.005 if "fixedwing" in name else 0.0), bad
    assert torch.isfinite(obs).all()
    env.close()


def test_fullsize_shards_concatenate_bit_exact():
    """65 536 lanes as one batch == 2 x 32 768 == the first 1/8 slice of the 524 288-lane batch."""
    from pyflyt_amd.gym_envs import QuadXHoverVecEnv

    n = 65536
    full = QuadXHoverVecEnv(n, seed=31)
    big = QuadXHoverVecEnv(8 * n, seed=31)
    halves = [QuadXHoverVecEnv(n // 2, seed=31, lane_offset=k * (n // 2)) for k in range(2)]
    of = full.reset(seed=31)[0]
    ob = big.reset(seed=31)[0]
    oh = torch.cat([h.reset(seed=31)[0] for h in halves])
    assert torch.equal(of, oh) and torch.equal(of, ob[:n])
    for k in range(20):
        a = big.sample_actions(k)
        rb = big.step(a)
        rf = full.step(a[:n].contiguous())
        rh = [h.step(a[j * (n // 2):(j + 1) * (n // 2)].contiguous()) for j, h in enumerate(halves)]
        for j in range(4):
            assert torch.equal(rf[j], torch.cat([rh[0][j], rh[1][j]]))
            assert torch.equal(rf[j], rb[j][:n])
    for e in [full, big] + halves:
        e.close()
