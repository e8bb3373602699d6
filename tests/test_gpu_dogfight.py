"""MAFixedwingDogfightEnv on the device (pyflyt_amd/csrc/dogfight.hpp) against
  * the three trajectories recorded from the reference's own env (tests/golden/env_dogfight_*.npz, the reference's motor
    noise and spawn poses injected), and
  * the fp64 oracle (oracle/uav_oracle.c:orc_dogfight_*) over many worlds with the counter RNG on both sides: spawn circle,
    motor noise, random actions.
fp32 device, fp64 checker: observations to 2e-4 relative (positions of aircraft that have flown ~100 m), rewards to
2e-3 (+1e-3 relative); a threshold event (cone of fire, range, a death) may fall on different sides in fp32 -- such worlds
are counted and bounded, not compared further."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
RTOL_IMPACT = 5e-3  # after a ground impact in the world (see test_gpu_golden.py / test_gpu_parity.py)
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _engine(n_worlds, noise, seed=0, sample_spawn=False, lane_offset=0, **kw):
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    df = dict(sample_spawn=sample_spawn)
    for k in ("assisted_flight", "team_size", "damage_per_hit", "lethal_distance", "lethal_angle", "aggressiveness", "cooperativeness", "spawn_min_radius", "spawn_max_radius"):
        if k in kw:
            df[k] = kw.pop(k)
    P = build_params("fixedwing", "dogfight", noise=noise, autoreset="off", seed=seed, angle_representation="euler",
                     vehicle_options=dict(drone_model="acrowing"), world_options=dict(world_scale=5.0), dogfight=df, **kw)
    A = 2 * df.get("team_size", 2)
    return BatchEngine(P, n_worlds * A, device="cuda:0", lane_offset=lane_offset), A


def _set_spawn(eng, pos, rpy):
    """state groups 13 / 14: (x, y, z, roll), (pitch, yaw, -, -) per lane."""
    n = eng.n
    g = torch.zeros(2, n, 4, dtype=torch.float32, device="cuda:0")
    g[0, :, :3] = torch.tensor(pos, dtype=torch.float32)
    g[0, :, 3] = torch.tensor(rpy[:, 0], dtype=torch.float32)
    g[1, :, 0] = torch.tensor(rpy[:, 1], dtype=torch.float32)
    g[1, :, 1] = torch.tensor(rpy[:, 2], dtype=torch.float32)
    eng.state[13:15] = g


def _rel(a, b):
    return np.abs(a - b) / np.maximum(1.0, np.abs(b))


@pytest.mark.parametrize("name", ["env_dogfight_default", "env_dogfight_engage", "env_dogfight_crash", "env_dogfight_team1_sparse", "env_dogfight_team3",
                                  "env_dogfight_unassisted"])
def test_dogfight_golden_replay(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    E = 16  # copies of the recorded world side by side (one wave)
    eng, A = _engine(E, "inject", team_size=int(g["team_size"]), damage_per_hit=float(g["damage_per_hit"]), lethal_distance=float(g["lethal_distance"]),
                     lethal_angle=float(g["lethal_angle"]), aggressiveness=float(g["aggressiveness"]), cooperativeness=float(g["cooperativeness"]),
                     sparse_reward=bool(g["sparse_reward"]), flight_dome_size=float(g["dome"]), max_duration_seconds=int(g["max_steps"]) / 30.0,
                     assisted_flight=int(g["action_dim"]) == 4)
    S = 19 + int(g["action_dim"])  # width of the own block
    assert eng.obs_dim == g["reset_obs"].shape[1] and int(eng.params.max_steps) == int(g["max_steps"])
    _set_spawn(eng, np.tile(g["start_pos"], (E, 1)), np.tile(g["start_orn"], (E, 1)))
    tile = lambda x: torch.tensor(np.tile(x, (1, E)), dtype=torch.float32, device="cuda:0").contiguous()  # [ticks, A] -> [ticks, n]
    eng.env_reset(xi_reset=tile(g["reset_xi"]))
    obs = eng.obs.cpu().numpy().astype(np.float64).reshape(E, A, -1)
    assert _rel(obs, g["reset_obs"][None]).max() < 1e-4
    assert np.abs(obs - obs[0:1]).max() == 0.0  # identical worlds stay bit-identical
    worst_o = worst_r = worst_impact = 0.0
    layout_mismatch = np.zeros(A, dtype=int)  # steps in which an observer's rows of the others are laid out differently
    crashed, touched = set(), False  # an aircraft of the world has hit the ground: from then on its tumbling shows the fp32 sensitivity of
    # impacts (tests/tools/fp32_contact_sensitivity.py) in everybody's observation of it -> the impact tolerance
    for k in range(len(g["action"])):
        act = torch.tensor(np.tile(g["action"][k], (E, 1)), dtype=torch.float32, device="cuda:0")
        o, r, t, u = eng.env_step(act, xi=tile(g["xi"][k]))
        from pyflyt_amd import _lib as PL
        crashed |= set(np.nonzero((eng.flags()[:A] & PL.F_CONTACT).cpu().numpy())[0].tolist())
        touched = len(crashed) > 0
        o = o.cpu().numpy().astype(np.float64).reshape(E, A, -1)[0]
        r, t, u = r.cpu().numpy().reshape(E, A)[0], t.cpu().numpy().reshape(E, A)[0], u.cpu().numpy().reshape(E, A)[0]
        st = eng.state[6].cpu().numpy().reshape(E, A, 4)[0]
        health, hits = st[:, 0], st[:, 2].view(np.int32)
        for i in range(A):
            if g["alive"][k][i]:
                eo = _rel(o[i], g["obs"][k][i]).max()
                er = abs(r[i] - g["reward"][k][i]) / max(1.0, abs(g["reward"][k][i]))
                if touched:
                    # the crashed aircraft tumbles over the ground: chaotic in fp32 vs fp64 within a few bounces (and the step at
                    # which it comes to rest and drops out of the others' rows may differ by one). From the impact on: everybody's
                    # OWN block to the flight tolerance, the rows of the others to the impact tolerance while the layouts agree
                    eo = _rel(o[i][:S], g["obs"][k][i][:S]).max()
                    worst_o = max(worst_o, eo)
                    assert eo < 5e-4, (name, k, i, eo)
                    rows_d, rows_g = o[i][S:].reshape(A - 1, 14), g["obs"][k][i][S:].reshape(A - 1, 14)
                    nz_d, nz_g = np.abs(rows_d).sum(1) > 0, np.abs(rows_g).sum(1) > 0
                    if not (nz_d == nz_g).all():
                        # the wreck came to rest -- and left the others' rows -- one step apart on the two sides: counted and
                        # bounded below, not skipped silently
                        layout_mismatch[i] += 1
                    else:
                        others = [j for j in range(A) if j != i]
                        if nz_g.sum() < len(others):  # a row has dropped out: the aircraft at rest on the ground
                            others = [j for j in others if j not in crashed]
                        flying = np.array([j not in crashed for j in others] + [False] * (A - 1 - len(others)))
                        if flying.any():
                            worst_impact = max(worst_impact, _rel(rows_d[flying], rows_g[flying]).max())
                            assert _rel(rows_d[flying], rows_g[flying]).max() < RTOL_IMPACT, (name, k, i)
                else:
                    worst_o = max(worst_o, eo)
                    assert eo < 5e-4, (name, k, i, eo, int(_rel(o[i], g["obs"][k][i]).argmax()))
                worst_r = max(worst_r, er)
                assert er < 3e-3, (name, k, i, r[i], g["reward"][k][i])
                assert bool(t[i]) == bool(g["term"][k][i]) and bool(u[i]) == bool(g["trunc"][k][i]), (name, k, i)
            else:
                assert r[i] == 0.0 and not t[i] and not u[i]  # culled agents report nothing
        np.testing.assert_allclose(health, g["health"][k], atol=2e-5)
        assert (hits == g["received_hits"][k]).all(), (name, k, hits, g["received_hits"][k])
    # the row layouts may disagree only around the step in which a wreck comes to rest: at most two steps per observer
    assert layout_mismatch.max() <= 2, (name, layout_mismatch)
    if name == "env_dogfight_crash":  # the dead aircraft at rest has dropped out of the survivors' observations on both sides
        assert (np.abs(o[0][23:].reshape(A - 1, 14)).sum(1) > 0).sum() == (np.abs(g["obs"][-1][0][23:].reshape(A - 1, 14)).sum(1) > 0).sum() == 2
    print(f"{name}: worst obs {worst_o:.2e} (after a ground impact {worst_impact:.2e}) reward {worst_r:.2e} over {len(g['action'])} steps; "
          f"row-layout mismatches per observer {layout_mismatch.tolist()}")


def test_midair_collision_pushes_the_aircraft_apart():
    """tests/golden/env_dogfight_midair.npz: two aircraft meet head-on (ma_fixedwing_dogfight_env.py:672-676: both are out in that
    step) and fly on as wrecks that the two survivors keep observing. What the wrecks do afterwards is stepSimulation's contact
    response BETWEEN the aircraft (shared_world.hpp: pair_stage_dev, in the dogfight kernel since round 4): the device follows the
    recording made with it, and is metres away from the control recording made without it."""
    g, g0 = np.load(os.path.join(GOLD, "env_dogfight_midair.npz")), np.load(os.path.join(GOLD, "env_dogfight_midair_nopair.npz"))
    E = 16
    eng, A = _engine(E, "inject", team_size=int(g["team_size"]), damage_per_hit=float(g["damage_per_hit"]), lethal_distance=float(g["lethal_distance"]),
                     lethal_angle=float(g["lethal_angle"]), aggressiveness=float(g["aggressiveness"]), cooperativeness=float(g["cooperativeness"]),
                     sparse_reward=bool(g["sparse_reward"]), flight_dome_size=float(g["dome"]), max_duration_seconds=int(g["max_steps"]) / 30.0)
    S = 19 + int(g["action_dim"])
    _set_spawn(eng, np.tile(g["start_pos"], (E, 1)), np.tile(g["start_orn"], (E, 1)))
    tile = lambda x: torch.tensor(np.tile(x, (1, E)), dtype=torch.float32, device="cuda:0").contiguous()  # noqa: E731
    eng.env_reset(xi_reset=tile(g["reset_xi"]))
    k0 = int(np.argmax(g["term"][:, 0]))
    assert g["term"][k0, 2] and not g["term"][:, [1, 3]].any()
    worst_before = worst_after = gap_control = 0.0
    for k in range(min(len(g["action"]), k0 + 25)):
        act = torch.tensor(np.tile(g["action"][k], (E, 1)), dtype=torch.float32, device="cuda:0")
        o, r, t, u = eng.env_step(act, xi=tile(g["xi"][k]))
        o = o.cpu().numpy().astype(np.float64).reshape(E, A, -1)[0]
        t = t.cpu().numpy().reshape(E, A)[0]
        for i in range(A):
            if not g["alive"][k][i]:
                continue
            assert bool(t[i]) == bool(g["term"][k][i]), (k, i)
            e = _rel(o[i], g["obs"][k][i]).max()
            if k < k0:
                worst_before = max(worst_before, e)
                assert e < 5e-4, (k, i, e)
            elif i in (1, 3):  # the survivors: their own block to the flight tolerance, their rows of the wrecks to the impact tolerance
                assert _rel(o[i][:S], g["obs"][k][i][:S]).max() < 5e-4, (k, i)
                worst_after = max(worst_after, e)
                gap_control = max(gap_control, _rel(g0["obs"][k][i], g["obs"][k][i]).max())
        if k == k0:  # (-1000 for the collision, then overridden by the element-wise team-win rule: both teams lost their opponent's last ... as recorded)
            np.testing.assert_allclose(r.cpu().numpy().reshape(E, A)[0][[0, 2]], g["reward"][k][[0, 2]], rtol=1e-5)
    print(f"mid-air collision at step {k0}: worst before {worst_before:.2e}; the survivors' view of the wrecks over the next 24 steps: device vs recording {worst_after:.2e}, "
          f"recording without the pair stage vs recording {gap_control:.2e}")
    assert worst_after < 6e-2          # (measured 2.2e-2) a head-on impact at 40 m/s closing speed, tumbling wrecks: the fp32 sensitivity of such a transient
    assert gap_control > 20 * worst_after


def _uniforms(seed, lane_id, ctr, count, stream):
    """The device's reset-time uniforms for a world (Noise::uniform): Philox4x32 keyed (seed, first lane of the world, event
    counter, flat >> 2, stream), through the oracle's generator."""
    import ctypes as C
    u, buf = np.zeros(count), (C.c_double * 4)()
    for call in range((count + 3) // 4):
        O.lib().orc_uniform4(seed, lane_id, ctr, call, stream, buf)
        for j in range(4):
            if 4 * call + j < count:
                u[4 * call + j] = buf[j]
    return u


def _run_vs_oracle(eng, A, worlds, steps, act_fn, label):
    Wn, n = len(worlds), len(worlds) * A
    ro = np.stack([w.reset() for w in worlds])
    do = eng.obs.cpu().numpy().astype(np.float64).reshape(Wn, A, -1)
    assert _rel(do, ro).max() < 2e-4, _rel(do, ro).max()
    ok = np.ones(Wn, dtype=bool)
    worst_o = worst_r = 0.0
    events = dict(hits=0, dead=0, win=0, oob=0, trunc=0, collision=0)
    for k in range(steps):
        act = act_fn(k).astype(np.float32)
        o, r, t, u = eng.env_step(torch.tensor(act.reshape(n, 4), device="cuda:0"))
        o = o.cpu().numpy().astype(np.float64).reshape(Wn, A, -1)
        r, t, u = r.cpu().numpy().reshape(Wn, A), t.cpu().numpy().reshape(Wn, A), u.cpu().numpy().reshape(Wn, A)
        st = eng.state[6].cpu().numpy().reshape(Wn, A, 4)
        for w, W in enumerate(worlds):
            if not ok[w]:
                continue
            alive = W.alive.copy()
            oo, rr, tt, uu = W.step(act[w].astype(np.float64))
            if (t[w][alive] != tt[alive]).any() or (u[w][alive] != uu[alive]).any() or (st[w, :, 2].view(np.int32) != np.array(W.D.received_hits[:A])).any():
                ok[w] = False  # a threshold event (cone of fire, range, death) fell on the other side in fp32: stop comparing this world
                continue
            if any(L.contact_step for L in W.Ls):
                events["collision"] += 1
                ok[w] = False  # ground impacts are compared in the golden replay; the tumbling afterwards is chaotic
                continue
            eo = max((_rel(o[w][i], oo[i]).max() for i in range(A) if alive[i]), default=0.0)
            er = max((abs(r[w][i] - rr[i]) / max(1.0, abs(rr[i])) for i in range(A) if alive[i]), default=0.0)
            worst_o, worst_r = max(worst_o, eo), max(worst_r, er)
            bits = np.array(W.D.info_bits[:A])
            events["hits"] += int(np.array(W.D.cur_hit).sum()); events["dead"] += int(((bits & 1) != 0).any())
            events["win"] += int(((bits & 8) != 0).any()); events["oob"] += int(((bits & 4) != 0).any()); events["trunc"] += int(uu.any())
    frac = (ok.sum() + events["collision"]) / Wn
    print(f"{label}: worlds compared to the end (or to a ground impact) {frac:.3f}, worst obs {worst_o:.2e} reward {worst_r:.2e}, events {events}")
    return frac, worst_o, worst_r, events


def test_dogfight_sampled_spawn_vs_oracle():
    """The world's spawn circle from the counter RNG (_get_start_pos_orn, :176-213) against the oracle's restatement fed the same
    uniforms, then 40 steps of random actions with Philox motor noise on both sides."""
    Wn, seed, lane_offset = 32, 13, 4096
    kw = dict(flight_dome_size=75.0, max_duration_seconds=1.0)
    eng, A = _engine(Wn, "philox", seed=seed, sample_spawn=True, lane_offset=lane_offset, **kw)
    eng.env_reset()
    sp = eng.state[13:15].cpu().numpy()
    dev_pos, dev_yaw = sp[0, :, :3].reshape(Wn, A, 3), sp[1, :, 1].reshape(Wn, A)
    worlds = []
    for w in range(Wn):
        pos, rpy, vel = O.dogfight_spawn(2, 10.0, 50.0, _uniforms(seed, lane_offset + w * A, 0, 1 + 3 * A, 2))
        assert np.abs(pos - dev_pos[w]).max() < 2e-4 and np.abs(rpy[:, 2] - dev_yaw[w]).max() < 2e-6, w
        worlds.append(O.OracleDogfight(pos, rpy, start_vel=vel, noise_mode=O.NOISE_PHILOX, seed=seed, lane_id0=lane_offset + w * A,
                                       dome=kw["flight_dome_size"], max_duration_seconds=kw["max_duration_seconds"]))
    assert np.abs(dev_pos[0] - dev_pos[1]).max() > 1.0  # every world draws its own circle
    rng = np.random.default_rng(5)
    frac, wo, wr, ev = _run_vs_oracle(eng, A, worlds, 40, lambda k: rng.uniform(-1.0, 1.0, size=(Wn, A, 4)), "dogfight, sampled spawn")
    assert frac >= 0.9 and wo < 5e-4 and wr < 5e-3
    assert ev["trunc"] > 0 and ev["oob"] > 0  # 1 s episodes; spawned up to 70 m out at 20 m/s inside a 75 m dome


def test_dogfight_engagements_vs_oracle():
    """Worlds set up for a fight: every aircraft of team 1 flies 15-40 m ahead of one of team 0, roughly on its heading, inside a
    generous cone of fire -- hits from the first step, health running out, deaths, (element-wise) team wins, culled aircraft
    flying on. Every world against its own fp64 oracle world; Philox motor noise, gentle random actions."""
    Wn, seed, lane_offset = 48, 21, 640
    kw = dict(damage_per_hit=0.006, lethal_distance=45.0, lethal_angle=0.3, flight_dome_size=500.0, max_duration_seconds=1.5)
    eng, A = _engine(Wn, "philox", seed=seed, sample_spawn=False, lane_offset=lane_offset, **kw)
    rng = np.random.default_rng(3)
    pos, rpy = np.zeros((Wn, A, 3)), np.zeros((Wn, A, 3))
    for w in range(Wn):
        for i in range(2):  # team 0
            pos[w, i] = [rng.uniform(-60, 60), rng.uniform(-60, 60) + 150.0 * i, rng.uniform(40, 60)]
            rpy[w, i, 2] = rng.uniform(-np.pi + 0.2, np.pi - 0.2)
            j = 2 + i      # its quarry
            d = rng.uniform(15, 40)
            off = rng.uniform(-0.15, 0.15)
            pos[w, j] = pos[w, i] + d * np.array([np.cos(rpy[w, i, 2] + off), np.sin(rpy[w, i, 2] + off), rng.uniform(-0.05, 0.05)])
            rpy[w, j, 2] = rpy[w, i, 2] + rng.uniform(-0.1, 0.1)
    _set_spawn(eng, pos.reshape(-1, 3), rpy.reshape(-1, 3))
    eng.env_reset()
    worlds = [O.OracleDogfight(pos[w], rpy[w], noise_mode=O.NOISE_PHILOX, seed=seed, lane_id0=lane_offset + w * A,
                               damage_per_hit=kw["damage_per_hit"], lethal_distance=kw["lethal_distance"], lethal_angle=kw["lethal_angle"],
                               dome=kw["flight_dome_size"], max_duration_seconds=kw["max_duration_seconds"]) for w in range(Wn)]
    base = np.array([0.0, 0.0, 0.0, 0.4])
    frac, wo, wr, ev = _run_vs_oracle(eng, A, worlds, 50, lambda k: base + rng.uniform(-0.15, 0.15, size=(Wn, A, 4)), "dogfight, engagements")
    assert frac >= 0.8 and wo < 5e-4 and wr < 5e-3
    assert ev["hits"] > 500 and ev["dead"] > 0 and ev["win"] > 0 and ev["trunc"] > 0


def test_dogfight_masked_reset_whole_worlds():
    eng, A = _engine(8, "philox", seed=3, sample_spawn=True)
    eng.env_reset()
    act = torch.zeros(eng.n, 4, device="cuda:0"); act[:, 3] = 0.3
    for _ in range(5):
        eng.env_step(act)
    before = eng.state.clone()
    mask = torch.zeros(eng.n, dtype=torch.bool, device="cuda:0"); mask[2 * A:3 * A] = True  # world 2 only
    eng.env_reset(mask=mask)
    keep = ~mask
    assert torch.equal(eng.state[:, keep], before[:, keep])                 # the other worlds are untouched
    assert (eng.state[6, mask, 0] == 1.0).all() and (eng.state[5, mask, 0].view(torch.int32) == 0).all()  # health 1, step_count 0
    assert not torch.equal(eng.state[13, mask], before[13, mask])           # a new spawn circle (the event counter moved on)
    # a mask that names ONE aircraft of a world resets that world -- widened on the device (shared_world.hpp: widen_to_world)
    before = eng.state.clone()
    one = torch.zeros(eng.n, dtype=torch.bool, device="cuda:0"); one[1] = True
    eng.env_reset(mask=one)
    world0 = torch.zeros(eng.n, dtype=torch.bool, device="cuda:0"); world0[:A] = True
    assert torch.equal(eng.state[:, ~world0], before[:, ~world0])
    assert (eng.state[6, world0, 0] == 1.0).all() and (eng.state[5, world0, 0].view(torch.int32) == 0).all()


def test_dogfight_refusals():
    from pyflyt_amd import PyFlytAmdError, build_params
    from pyflyt_amd.engine import BatchEngine

    P = build_params("fixedwing", "dogfight", autoreset="off", angle_representation="euler", vehicle_options=dict(drone_model="acrowing"))
    with pytest.raises(PyFlytAmdError):
        BatchEngine(P, 6, device="cuda:0")  # not a whole number of worlds
    P = build_params("fixedwing", "dogfight", autoreset="next_step", angle_representation="euler", vehicle_options=dict(drone_model="acrowing"))
    with pytest.raises(PyFlytAmdError):
        BatchEngine(P, 8, device="cuda:0")  # PettingZoo envs have no auto-reset


@pytest.mark.parametrize("team_size", [1, 2, 3, 4])
def test_dogfight_pz_api(team_size):
    """The PettingZoo-shaped façade: agent naming, spaces, dict in / dict out, culling, infos; world sizes that do not divide the
    wavefront (team_size 3: six aircraft per world, ten worlds per wave); the same seed draws the same spawn circle."""
    from pyflyt_amd.pz_envs import MAFixedwingDogfightEnv

    A, E = 2 * team_size, 11
    env = MAFixedwingDogfightEnv(team_size=team_size, num_envs=E, seed=4, max_duration_seconds=1.0, flight_dome_size=90.0,
                                 lethal_distance=80.0, lethal_angle_radians=0.6, damage_per_hit=0.02)
    assert env.possible_agents == [f"uav_{i}" for i in range(A)]
    assert env.observation_space("uav_0").shape == (23 + (A - 1) * 14,) and env.action_space("uav_0").shape == (4,)
    obs, infos = env.reset(seed=4)
    assert set(obs) == set(env.possible_agents) and all(o.shape == (E, 23 + (A - 1) * 14) for o in obs.values())
    spawn0 = env.start_pos.clone()
    assert torch.isfinite(torch.stack(list(obs.values()))).all()
    # aircraft spawn on a circle of radius 10..50 around the origin, 10..50 m up, health 1
    r = spawn0[..., :2].norm(dim=-1)
    assert (r >= 10 - 1e-3).all() and (r <= 50 + 1e-3).all() and (spawn0[..., 2] >= 10 - 1e-3).all() and (env.healths == 1.0).all()
    g = torch.Generator().manual_seed(0)
    seen_done, k = set(), 0
    while env.agents:
        acts = {a: torch.rand(E, 4, generator=g) * 2 - 1 for a in env.agents}
        obs, rew, term, trunc, infos = env.step(acts)
        assert set(obs) == set(acts) == set(rew) == set(term) == set(trunc) == set(infos)
        for a in acts:
            assert obs[a].shape == (E, 23 + (A - 1) * 14) and rew[a].shape == (E,) and torch.isfinite(obs[a]).all() and torch.isfinite(rew[a]).all()
            assert set(infos[a]) == {"health", "received_hits", "dead", "collision", "out_of_bounds", "team_win"}
            if bool((term[a] | trunc[a]).all()):
                seen_done.add(a)
        k += 1
        assert k <= 40
    assert seen_done == set(env.possible_agents)  # 1 s episodes: everybody is truncated at the latest
    from pyflyt_amd import _lib as PL
    assert not (env.engine.flags() & PL.F_NONFINITE).any()
    obs2, _ = env.reset(seed=4)
    assert torch.equal(env.start_pos, spawn0)  # same seed, same circle
    env.reset(seed=5)
    assert not torch.equal(env.start_pos, spawn0)
    env.close()


def test_dogfight_pz_dict_observation():
    """flatten_observation=False: the reference's Dict form as a view of the flattened vector."""
    from pyflyt_amd.pz_envs import MAFixedwingDogfightEnv

    env = MAFixedwingDogfightEnv(team_size=2, flatten_observation=False, seed=2)
    flat = MAFixedwingDogfightEnv(team_size=2, flatten_observation=True, seed=2)
    o1, _ = env.reset(seed=2)
    o2, _ = flat.reset(seed=2)
    for a in env.possible_agents:
        assert o1[a]["self"].shape == (23,) and o1[a]["others"].shape == (3, 14)
        assert torch.equal(torch.cat([o1[a]["self"], o1[a]["others"].flatten()]), o2[a])
    env.close(); flat.close()
    env = MAFixedwingDogfightEnv(team_size=1, flatten_observation=False, num_envs=5, seed=2)
    o, _ = env.reset()
    assert o["uav_0"]["self"].shape == (5, 23) and o["uav_0"]["others"].shape == (5, 1, 14) and o["uav_0"]["others_mask"].all()
    env.close()
    env = MAFixedwingDogfightEnv(assisted_flight=False, flatten_observation=False, seed=2)  # six-wide actions, own block 25 wide
    o, _ = env.reset()
    assert env.action_space("uav_0").shape == (6,) and o["uav_0"]["self"].shape == (25,)
    o, r, t, u, i = env.step({a: torch.zeros(6) for a in env.agents})
    assert o["uav_0"]["self"].shape == (25,)
    env.close()


def test_dogfight_freeze_wrecks():
    """dogfight=dict(freeze_wrecks=True): an aircraft stops where it hits the ground and leaves the others' observations within
    two updates; until the impact the trajectory is the default one bit for bit; nobody else is affected afterwards."""
    g = np.load(os.path.join(GOLD, "env_dogfight_crash.npz"))
    kw = dict(flight_dome_size=float(g["dome"]), max_duration_seconds=int(g["max_steps"]) / 30.0)
    a, A = _engine(1, "inject", **kw)
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine
    P = build_params("fixedwing", "dogfight", noise="inject", autoreset="off", angle_representation="euler", vehicle_options=dict(drone_model="acrowing"),
                     world_options=dict(world_scale=5.0), dogfight=dict(sample_spawn=False, freeze_wrecks=True), **kw)
    b = BatchEngine(P, A, device="cuda:0")
    for e in (a, b):
        _set_spawn(e, g["start_pos"], g["start_orn"])
        e.env_reset(xi_reset=torch.tensor(g["reset_xi"], dtype=torch.float32, device="cuda:0").contiguous())
    hit = None
    for k in range(len(g["action"])):
        act = torch.tensor(g["action"][k], dtype=torch.float32, device="cuda:0")
        xi = torch.tensor(g["xi"][k], dtype=torch.float32, device="cuda:0").contiguous()
        oa = a.env_step(act, xi=xi)[0].clone()
        ob = b.env_step(act, xi=xi)[0].clone()
        from pyflyt_amd import _lib as PL
        if hit is None:
            assert torch.equal(a.state[:6], b.state[:6]) or bool((b.flags() & PL.F_CONTACT).any()), k
            if bool((b.flags() & PL.F_CONTACT).any()):
                hit = k
        else:
            # the aircraft that never touched the ground fly on identically (drone-drone coupling aside: none here)
            fly = [i for i in range(A) if not bool(b.flags()[i] & PL.F_CONTACT)]
            assert torch.equal(a.state[:5, fly], b.state[:5, fly]), k
            if k >= hit + 2:
                wreck = [i for i in range(A) if i not in fly]
                assert (b.state[6, wreck, 3].view(torch.int32) & 8).ne(0).all()          # inactive
                assert (b.state[2, wreck, :3].abs().max() == 0)                           # at rest where it hit
                rows = ob[fly[0]][23:].view(A - 1, 14)
                assert int((rows.abs().sum(1) > 0).sum()) == A - 1 - len(wreck)           # gone from the survivors' observations
    assert hit is not None and hit > 50


def test_dogfight_deterministic():
    """pettingzoo's check_environment_deterministic_parallel in spirit (tests/test_pz_envs.py): two envs, the same seed, the same
    actions -> bit-identical observations, rewards and flags, step after step (counter RNG, no atomics, no reductions)."""
    from pyflyt_amd.pz_envs import MAFixedwingDogfightEnv

    envs = [MAFixedwingDogfightEnv(team_size=2, num_envs=37, seed=9, max_duration_seconds=1.0, lethal_distance=80.0, lethal_angle_radians=0.6) for _ in range(2)]
    o = [e.reset(seed=9)[0] for e in envs]
    assert all(torch.equal(o[0][a], o[1][a]) for a in o[0])
    g = torch.Generator().manual_seed(1)
    for k in range(35):
        if not envs[0].agents:
            break
        assert envs[0].agents == envs[1].agents
        acts = {a: torch.rand(37, 4, generator=g) * 2 - 1 for a in envs[0].agents}
        outs = [e.step(acts) for e in envs]
        for a in acts:
            for x, y in zip(outs[0][:4], outs[1][:4]):
                assert torch.equal(x[a], y[a]), (k, a)
    for e in envs:
        e.close()


def test_facade_per_copy_reset_and_latched_done():
    """MAFixedwingDogfightEnv(num_envs > 1): `done` latches an agent's episode end per copy, `reset_envs(mask)` restarts the
    selected copies only, and a re-seeded reset keeps the action memories (groups 7, 8 and 15) as the reference keeps
    current_actions / past_actions across resets."""
    from pyflyt_amd.pz_envs import MAFixedwingDogfightEnv

    E = 6
    env = MAFixedwingDogfightEnv(num_envs=E, seed=3, max_duration_seconds=0.5, assisted_flight=False)  # 15-step episodes: truncation
    A = env.num_possible_agents
    env.reset(seed=3)
    act = {ag: torch.zeros(E, 6, device="cuda:0") + torch.tensor([0.0, 0.0, 0.0, 0.1, 0.2, 0.6], device="cuda:0") for ag in env.possible_agents}
    seen_done = torch.zeros(E, A, dtype=torch.bool, device="cuda:0")
    for k in range(40):
        if not env.agents:
            break
        o, r, t, u, info = env.step({ag: act[ag] for ag in env.agents})
        for ag in t:
            seen_done[:, env.agent_name_mapping[ag]] |= (t[ag] | u[ag])
        assert torch.equal(env.done, seen_done), k
    assert bool(env.done.all())  # every episode was truncated at the latest
    steps_before = env.engine.ints()[:, 0].clone()
    mask = torch.tensor([True, False, False, True, False, False], device="cuda:0")
    env.reset_envs(mask)
    steps = env.engine.ints()[:, 0].view(E, A)
    assert (steps[mask] == 0).all() and torch.equal(steps[~mask], steps_before.view(E, A)[~mask])
    assert not bool(env.done[mask].any()) and bool(env.done[~mask].all())
    # a re-seeded reset rebuilds the engine: the action memories survive it
    mem = (env.engine.state[7:9].clone(), env.engine.state[15].clone())
    assert float(mem[1].abs().sum()) > 0  # (entries 4, 5 of the six-wide actions were non-zero)
    env.reset(seed=4)
    assert torch.equal(env.engine.state[7:9], mem[0]) and torch.equal(env.engine.state[15], mem[1])
    env.close()


def test_dead_aircraft_at_a_momentary_standstill_moves_on():
    """`inactive` (dead, below 2 m, slower than 0.1 m/s: ma_fixedwing_dogfight_env.py:505-510) is recomputed on every update and
    the reference keeps stepping the body in Bullet: a dead aircraft that passes through a standstill in mid-air -- the apex of a
    bounce -- is not a wreck at rest. It must keep falling; only after it has sat on the ground, still, for kDfRestUpdates
    consecutive updates does the device stop integrating it (DF_AT_REST = 8192)."""
    eng, A = _engine(1, "inject", max_duration_seconds=3.0)
    g = np.load(os.path.join(GOLD, "env_dogfight_crash.npz"))
    _set_spawn(eng, g["start_pos"], g["start_orn"])
    eng.env_reset(xi_reset=torch.zeros_like(torch.tensor(g["reset_xi"], dtype=torch.float32)).to("cuda:0").contiguous())
    s = eng.state
    s[0, 0, :3] = torch.tensor([0.0, 0.0, 1.5], device="cuda:0")   # 1.5 m up, level (spawn attitude), ...
    s[2, 0, :] = 0.0; s[3, 0, :2] = 0.0                              # ... at a standstill
    s[6, 0, 0] = 0.0                                                 # health 0: dead
    fl = s[6, :, 3].view(torch.int32)
    fl[0] = int(fl[0]) & ~1                                          # culled (not DF_ALIVE)
    n_xi = g["xi"][0].shape[0]
    ps = []
    for k in range(12):
        eng.env_step(torch.zeros(A, 4, device="cuda:0"), xi=torch.zeros(n_xi, A, device="cuda:0"))
        f0 = int(eng.state[6, 0, 3].view(torch.int32))
        ps.append(eng.state[0, 0, :3].cpu().numpy().copy())
        assert not (f0 & 8192) or ps[-1][2] < 0.6, (k, f0, ps[-1])  # "at rest" only on the ground
    ps = np.array(ps)
    assert np.abs(ps[0] - np.array([0.0, 0.0, 1.5])).max() > 1e-3, ps[0]          # it moved in the very first step ...
    assert (np.abs(np.diff(ps, axis=0)).max(axis=1) > 1e-4).all(), ps             # ... and in every step after it: not frozen


def test_a_moving_aircraft_pushes_a_wreck_at_rest():
    """A wreck the device has stopped integrating (DF_AT_REST) is still a body in Bullet's world: the reference keeps stepping it, and
    an aircraft that slides into it pushes it along. The device wakes the wreck when a moving body comes within reach (world_exchange,
    round 5) -- before that the pair stage solved the hit against a body that never read its impulse back, and the momentum vanished.
    Scenario: a dead aircraft dropped from 0.4 m comes to rest; a second dead aircraft, level, 8 m/s, 3 m behind it and 0.2 m up, lands
    and runs into it. Device against the fp64 oracle world from the same state (the oracle's wreck is put at the device's resting pose
    when the second aircraft is placed: what is compared is the hit, not the 6 mm by which the two landings differ)."""
    g = np.load(os.path.join(GOLD, "env_dogfight_crash.npz"))
    eng, A = _engine(1, "inject", max_duration_seconds=30.0)
    _set_spawn(eng, g["start_pos"], g["start_orn"])
    eng.env_reset(xi_reset=torch.zeros(g["reset_xi"].shape, dtype=torch.float32, device="cuda:0").contiguous())
    W = O.OracleDogfight(g["start_pos"], g["start_orn"], noise_mode=O.NOISE_OFF, max_duration_seconds=30.0)
    W.reset()
    n_xi = g["xi"][0].shape[0]

    def place(i, p, q, v, w):
        s = eng.state
        s[0, i, :3] = torch.tensor(p, dtype=torch.float32, device="cuda:0"); s[1, i, :] = torch.tensor(q, dtype=torch.float32, device="cuda:0")
        s[2, i, :3] = torch.tensor(v, dtype=torch.float32, device="cuda:0"); s[2, i, 3] = float(w[0]); s[3, i, 0] = float(w[1]); s[3, i, 1] = float(w[2])
        s[6, i, 0] = 0.0                                         # health 0: dead ...
        fl = s[6, :, 3].view(torch.int32); fl[i] = int(fl[i]) & ~1   # ... and culled (not DF_ALIVE)
        L = W.Ls[i]
        for k in range(3):
            L.p[k], L.v[k], L.w[k] = float(np.float32(p[k])), float(np.float32(v[k])), float(np.float32(w[k]))
        for k in range(4):
            L.q[k] = float(np.float32(q[k]))
        W.D.alive[i] = 0; W.D.health[i] = 0.0

    def step():
        eng.env_step(torch.zeros(A, 4, device="cuda:0"), xi=torch.zeros(n_xi, A, device="cuda:0"))
        W.step(np.zeros((A, 4)))
        return int(eng.state[6, 0, 3].view(torch.int32))

    place(0, [10.0, 0.0, 0.4], [0, 0, 0, 1], [0, 0, 0], [0, 0, 0])
    rest_at = next((k for k in range(40) if step() & 8192), None)
    assert rest_at is not None, "the dropped wreck never came to rest"
    p_rest, q_rest = eng.state[0, 0, :3].cpu().numpy().copy(), eng.state[1, 0, :].cpu().numpy().copy()
    assert abs(float(p_rest[2]) - 0.1) < 5e-3 and np.abs(p_rest - np.array(W.Ls[0].p[:])).max() < 1e-2   # both sides: flat on the ground, where it fell
    place(0, p_rest, q_rest, [0, 0, 0], [0, 0, 0])
    place(1, [float(p_rest[0]) - 3.0, float(p_rest[1]), float(p_rest[2]) + 0.2], [0, 0, 0, 1], [8.0, 0, 0], [0, 0, 0])
    woke_at, worst, moved = None, 0.0, 0.0
    for k in range(14):
        f0 = step()
        pd = eng.state[0, :2, :3].cpu().numpy().astype(np.float64)
        po = np.array([W.Ls[i].p[:] for i in range(2)])
        if woke_at is None and not (f0 & 8192):
            woke_at = k
        if woke_at is None:  # nothing within reach yet: the wreck is not integrated, bit for bit where it was
            assert np.array_equal(eng.state[0, 0, :3].cpu().numpy(), p_rest), (k, pd[0], p_rest)
        err = float(_rel(pd, po).max())
        worst = max(worst, err)
        moved = float(pd[0][0] - p_rest[0])
        assert err < RTOL_IMPACT, (k, err, pd, po)
    vd, vo = eng.state[2, 0, :3].cpu().numpy().astype(np.float64), np.array(W.Ls[0].v[:])
    print(f"wreck at rest after {rest_at + 1} steps, woken {woke_at} steps after the second aircraft was placed, pushed {moved:.3f} m since; worst position error "
          f"(both bodies, 14 steps) {worst:.2e}; the wreck's velocity now: device {vd}, oracle {vo}")
    assert woke_at is not None and woke_at <= 5, woke_at         # (the hit comes 5 steps in; the wake reaches a bounding radius ahead of it)
    assert moved > 0.5, moved                                     # it was pushed along (the momentum used to vanish: moved == 0)
    assert abs(vd[0] - vo[0]) < 0.05 * abs(vo[0]) and vo[0] > 2.0, (vd, vo)   # ... and it is still sliding as fast as the oracle's
