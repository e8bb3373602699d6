"""PettingZoo-shaped multi-agent QuadX hover on the GPU: parity against the oracle through the dict
API, and the API invariants the reference's tests/test_pz_envs.py checks (observations inside the
space, agents culled once done, same seed => same rollout)."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kernel", ["specialised", "generic"])
def test_ma_hover_parity_and_api(monkeypatch, kernel):
    """Both device paths of the MA task: the specialised QuadX kernel (quadx_fast.hpp, level spawns) and
    the generic env_kernel (forced with PF_DISABLE_FAST, also taken when a spawn is tilted)."""
    from pyflyt_amd.pz_envs import MAQuadXHoverEnv

    if kernel == "generic":
        monkeypatch.setenv("PF_DISABLE_FAST", "1")
    E, seed = 64, 4
    env = MAQuadXHoverEnv(num_envs=E, seed=seed, flight_dome_size=2.5, max_duration_seconds=1.0)
    A = env.num_possible_agents
    lib = O.lib()
    Ps = [O.make_params("ma_hover", noise_mode=O.NOISE_PHILOX, seed=seed, start_pos=env.start_pos[i], dome=2.5, max_steps=40) for i in range(A)]
    lanes = [[O.Lane() for _ in range(A)] for _ in range(E)]
    obs, infos = env.reset(seed=seed)
    assert set(obs) == set(env.possible_agents) and env.agents == env.possible_agents
    for e in range(E):
        for i in range(A):
            lib.orc_env_reset(C.byref(Ps[i]), C.byref(lanes[e][i]), e * A + i, None, None)
    ref = np.array([[np.frombuffer(lanes[e][i].obs, dtype=np.float64, count=24) for i in range(A)] for e in range(E)])
    got = np.stack([obs[a].cpu().numpy() for a in env.possible_agents], axis=1)
    assert got.shape == (E, A, 24) and np.abs(got - ref).max() < 1e-5
    rng = np.random.default_rng(0)
    worst, done_seen = 0.0, 0
    for k in range(45):
        if not env.agents:
            break
        acts = {a: torch.tensor(np.concatenate([rng.uniform(-1, 1, size=(E, 3)), rng.uniform(0.2, 0.7, size=(E, 1))], 1).astype(np.float32), device="cuda:0")
                for a in env.agents}
        o, r, t, u, info = env.step(acts)
        for e in range(E):
            for i, a in enumerate(env.possible_agents):
                act = acts[a][e].cpu().numpy().astype(np.float64) if a in acts else np.zeros(4)
                lib.orc_env_step(C.byref(Ps[i]), C.byref(lanes[e][i]), act.ctypes.data_as(C.POINTER(C.c_double)), None)
        for a in o:
            i = env.agent_name_mapping[a]
            ref_o = np.array([np.frombuffer(lanes[e][i].obs, dtype=np.float64, count=24) for e in range(E)])
            ref_r = np.array([lanes[e][i].reward for e in range(E)])
            ref_t = np.array([bool(lanes[e][i].terminated) for e in range(E)])
            ref_u = np.array([bool(lanes[e][i].truncated) for e in range(E)])
            # strict: identical flags for every copy of every agent at every step; observation within 1e-4 of
            # each physical vector's magnitude; reward within 1e-4 relative (it carries -100 penalties)
            assert (t[a].cpu().numpy() == ref_t).all() and (u[a].cpu().numpy() == ref_u).all(), (k, a)
            got_o = o[a].cpu().numpy().astype(np.float64)
            err = 0.0
            for lo, hi in ((0, 3), (3, 7), (7, 10), (10, 13), (13, 17), (17, 21), (21, 24)):
                scale = np.maximum(1.0, np.linalg.norm(ref_o[:, lo:hi], axis=1, keepdims=True))
                err = max(err, float((np.abs(got_o[:, lo:hi] - ref_o[:, lo:hi]) / scale).max()))
            worst = max(worst, err)
            assert (np.abs(r[a].cpu().numpy() - ref_r) / np.maximum(1.0, np.abs(ref_r))).max() < 1e-4
            assert env.observation_space(a).shape == (24,) and torch.isfinite(o[a]).all()
            done_seen += int((ref_t | ref_u).sum())
    print(f"ma hover: worst obs err {worst:.2e}, episodes ended {done_seen}")
    assert worst < 1e-4 and done_seen > 0
    env.close()


def test_ma_hover_single_env_shapes_and_culling():
    from pyflyt_amd.pz_envs import MAQuadXHoverEnv

    env = MAQuadXHoverEnv(flight_dome_size=1.8, seed=1)
    obs, _ = env.reset(seed=1)
    assert obs["uav_0"].shape == (24,)
    for k in range(200):
        if not env.agents:
            break
        acts = {a: np.array([0.0, 0.0, 0.0, 0.8]) for a in env.agents}  # full thrust: leaves the dome
        o, r, t, u, info = env.step(acts)
        assert set(o) == set(acts) and all(isinstance(float(r[a]), float) for a in r)
    assert env.agents == []  # all culled
    env.close()


def _vec_err(got, ref):
    e = 0.0
    for lo, hi in ((0, 3), (3, 7), (7, 10), (10, 13), (13, 17), (17, 21), (21, 24)):
        scale = max(1.0, float(np.linalg.norm(ref[lo:hi])))
        e = max(e, float(np.abs(got[lo:hi] - ref[lo:hi]).max()) / scale)
    return e


@pytest.mark.parametrize("fixture", ["env_ma_quadx_hover_shared", "env_ma_quadx_hover_stack"])
@pytest.mark.parametrize("kernel", ["specialised", "generic"])
def test_shared_world_fixture_replay(golden_dir, monkeypatch, kernel, fixture):
    """tests/golden/env_ma_quadx_hover_shared.npz -- the reference's PettingZoo env on a world where two agents fly into
    each other and a dead drone ends up on the floor -- replayed through the HIP path (agents_per_world = 4: the four lanes
    of a world exchange poses through LDS every tick)."""
    import os

    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine
    from pyflyt_amd.params import quat_from_euler

    if kernel == "generic":
        monkeypatch.setenv("PF_DISABLE_FAST", "1")
    # (`stack`: a culled drone falls onto a live one -- the contact response BETWEEN the drones, ma_quadx_base_env.py:365-369)
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    A = g["start_pos"].shape[0]
    P = build_params("quadx", "ma_hover", noise="inject", autoreset="off", start_pos=g["start_pos"][np.argmin(g["start_pos"][:, 2])],
                     start_orn=g["start_orn"][0], flight_dome_size=float(g["dome"]), max_duration_seconds=int(g["max_steps"]) / 40.0,
                     agents_per_world=A, world_options=dict(contact_response=True))
    eng = BatchEngine(P, A, device="cuda:0")
    assert (eng.lib.pf_ctx_is_specialised(eng._ctx) != 0) == (kernel == "specialised")
    pose = np.concatenate([g["start_pos"], np.stack([quat_from_euler(o) for o in g["start_orn"]])], axis=1)
    side = np.zeros((A, 12), dtype=np.float32)
    side[:, :7] = pose
    eng.state[12:15] = torch.tensor(side, device="cuda:0").view(A, 3, 4).permute(1, 0, 2)
    t32 = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda:0")  # noqa: E731
    resets = set(int(k) for k in g["reset_before"])
    ri = 0

    def do_reset():
        nonlocal ri
        obs = eng.env_reset(xi_reset=t32(g["reset_xi"][ri])).double().cpu().numpy()
        for i in range(A):
            assert _vec_err(obs[i], g["reset_obs"][ri][i]) < 1e-4
        ri += 1

    do_reset()
    worst, hits, worst_pos, touched = 0.0, 0, 0.0, False
    for k in range(len(g["action"])):
        if k in resets:
            do_reset()
        obs, rew, term, trunc = eng.env_step(t32(g["action"][k]), xi=t32(g["xi"][k]))
        o = obs.double().cpu().numpy()
        touched = touched or bool(g["world_contact"][k])  # (a contact anywhere in the world this episode: drone-drone or floor)
        for i in range(A):
            if g["alive"][k][i]:
                e = _vec_err(o[i], g["obs"][k][i])
                # within reach of the floor, or in a world whose drones have pushed each other: the contact solver's impulses (RTOL_IMPACT)
                near_floor = g["obs"][k][i][12] < 0.12 or touched
                worst = max(worst, 0.0 if near_floor else e)
                assert e < (5e-3 if near_floor else 1e-4), (k, i, e)
                assert bool(term[i]) == bool(g["term"][k][i]) and bool(trunc[i]) == bool(g["trunc"][k][i]), (k, i)
                assert abs(float(rew[i]) - g["reward"][k][i]) <= 1e-3 * max(1.0, abs(g["reward"][k][i]))
        hits += int(g["drone_contact"][k].any())
        # the position of EVERY drone of the world, culled ones included (they stay in the world and push / are pushed): 1e-4 in
        # free flight, the impact tolerance from the first contact of the episode on (drone-drone or floor)
        pe = np.abs(eng.state[0, :, :3].double().cpu().numpy() - g["all_pos"][k]).max()
        worst_pos = max(worst_pos, pe)
        assert pe < (5e-3 if touched else 1e-4), (k, pe, touched)
        if k + 1 in resets:
            touched = False
    print(f"{fixture} [{kernel}]: worst obs {worst:.2e}, worst position of any drone {worst_pos:.2e}, steps with a drone-drone hit {hits}")
    assert hits > 0 and ri == len(g["reset_obs"])


def test_shared_world_parity_and_effect():
    """MAQuadXHoverEnv(shared_world=True) against the oracle's world-level step (Philox noise, 32 copies of a 4-agent env whose
    agents 0 and 1 steer into each other), and against the same env with independent lanes: the hit must end both episodes
    in the shared world and not in the independent one."""
    from pyflyt_amd.pz_envs import MAQuadXHoverEnv

    E, seed = 32, 11
    start_pos = np.array([[-0.15, 0.0, 1.0], [0.15, 0.0, 1.01], [0.0, 1.0, 1.0], [0.0, -1.0, 1.2]])
    kw = dict(start_pos=start_pos, start_orn=np.zeros((4, 3)), num_envs=E, seed=seed, flight_dome_size=3.0, max_duration_seconds=2.0)
    env = MAQuadXHoverEnv(shared_world=True, **kw)
    ind = MAQuadXHoverEnv(shared_world=False, **kw)
    A = env.num_possible_agents
    worlds = []
    for e in range(E):
        Ps = [O.make_params("ma_hover", noise_mode=O.NOISE_PHILOX, seed=seed, start_pos=start_pos[i], dome=3.0, max_steps=80, world_contact_response=1)
              for i in range(A)]
        worlds.append(O.OracleWorld(Ps, lane_id0=e * A))
    obs, _ = env.reset(seed=seed)
    ind.reset(seed=seed)
    ref0 = np.stack([w.reset() for w in worlds])  # [E, A, 24]
    got0 = np.stack([obs[a].cpu().numpy() for a in env.possible_agents], axis=1)
    assert np.abs(got0 - ref0).max() < 1e-5
    rng = np.random.default_rng(2)
    alive = np.ones((E, A), dtype=bool)      # per copy (the façade culls an agent once it is done in EVERY copy)
    hit_shared = np.zeros(E, dtype=bool)
    worst = 0.0
    for k in range(60):
        base = np.array([[0.0, 0.7, 0.0, 0.36], [0.0, -0.7, 0.0, 0.36], [0.0, 0.0, 0.3, 0.37], [0.0, 0.0, 0.0, 0.33]])
        acts_np = base[None] + rng.uniform(-0.05, 0.05, size=(E, A, 4)) * np.array([1, 1, 1, 0.2])
        acts = {a: torch.tensor(acts_np[:, i].astype(np.float32), device="cuda:0") for i, a in enumerate(env.possible_agents) if a in env.agents}
        if not acts:
            break
        o, r, t, u, info = env.step(acts)
        acts_i = {a: torch.tensor(acts_np[:, i].astype(np.float32), device="cuda:0") for i, a in enumerate(ind.possible_agents) if a in ind.agents}
        if acts_i:
            ind.step(acts_i)
        step_acts = np.zeros((E, A, 4))
        for i, a in enumerate(env.possible_agents):
            if a in acts:
                step_acts[:, i] = acts_np[:, i].astype(np.float32)
        for e, w in enumerate(worlds):
            ro, rr, rt, ru = w.step(step_acts[e])
            hit_shared[e] |= bool(w.Ls[0].contact_step and w.Ls[1].contact_step and w.Ls[0].p[2] > 0.3)
            for i, a in enumerate(env.possible_agents):
                if a in o and alive[e, i]:
                    got = o[a][e].cpu().numpy().astype(np.float64)
                    # the impact tolerance within reach of the floor and from a drone-drone hit on (the contact solve's impulses)
                    near_floor = ro[i][12] < 0.12 or bool(hit_shared[e])
                    err = _vec_err(got, ro[i])
                    worst = max(worst, 0.0 if near_floor else err)
                    assert err < (5e-3 if near_floor else 1e-4), (k, e, a, err)
                    assert bool(t[a][e]) == bool(rt[i]) and bool(u[a][e]) == bool(ru[i]), (k, e, a)
                    if rt[i] or ru[i]:
                        alive[e, i] = False
    print(f"shared world parity: worst {worst:.2e}; copies with a mid-air hit between agents 0 and 1: {int(hit_shared.sum())}/{E}")
    assert hit_shared.mean() > 0.5
    # the independent-lane env flies the same commands without that hit: agents 0 and 1 survive the step of the hit
    assert "uav_0" in ind.agents or ind.step_count >= 40
    env.close(); ind.close()


@pytest.mark.parametrize("kernel", ["specialised", "generic"])
def test_partial_reset_mask_is_widened_to_the_world(monkeypatch, kernel):
    """pf_env_reset with a mask that names ONE agent of a shared world resets that whole world (the agents of a world exchange
    data inside the kernels) and leaves the other worlds alone -- decided on the device, no host-side check."""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    if kernel == "generic":
        monkeypatch.setenv("PF_DISABLE_FAST", "1")
    A, n = 4, 16
    P = build_params("quadx", "ma_hover", noise="philox", autoreset="off", seed=5, agents_per_world=A, world_options=dict(contact_response=True))
    eng = BatchEngine(P, n, device="cuda:0")
    assert (eng.lib.pf_ctx_is_specialised(eng._ctx) != 0) == (kernel == "specialised")
    pos = torch.tensor([[-1.0, 0.0, 1.0], [1.0, 0.0, 1.0], [0.0, 1.0, 1.0], [0.0, -1.0, 1.0]], device="cuda:0").repeat(n // A, 1)
    eng.state[12, :, 0:3] = pos
    eng.state[12, :, 3] = 0.0; eng.state[13, :, 0] = 0.0; eng.state[13, :, 1] = 0.0; eng.state[13, :, 2] = 1.0
    eng.env_reset()
    act = torch.zeros(n, 4, device="cuda:0"); act[:, 3] = 0.4
    for _ in range(5):
        eng.env_step(act)
    before = eng.state.clone()
    assert (eng.ints()[:, 0] == 5).all()
    mask = torch.zeros(n, dtype=torch.bool, device="cuda:0")
    mask[6] = True  # one agent of world 1
    eng.env_reset(mask=mask)
    steps = eng.ints()[:, 0].cpu().numpy()
    assert (steps[4:8] == 0).all() and (steps[:4] == 5).all() and (steps[8:] == 5).all(), steps
    others = [i for i in range(n) if not 4 <= i < 8]
    assert torch.equal(eng.state[:12, others], before[:12, others])


# ---------------------------------------------------------------------------------------------------------------- round 6: the facade's own cost
@pytest.mark.parametrize("shared", [False, True])
def test_action_buffers_and_no_cull_path_equals_the_dict_path(shared):
    """step() fed with the env's own action views (action_buffers(): no copy in), culling off (no host synchronisation), against
    a second env fed with numpy arrays through the copying path and culling on: same observations, rewards, flags and lazy infos,
    bit for bit; the culled env's agent list shrinks, the other's stays whole."""
    from pyflyt_amd import _lib as L
    from pyflyt_amd.pz_envs import MAQuadXHoverEnv

    E = 32
    kw = dict(num_envs=E, seed=2, flight_dome_size=2.2, max_duration_seconds=1.0, shared_world=shared)
    fast, ref = MAQuadXHoverEnv(cull_agents=False, **kw), MAQuadXHoverEnv(**kw)
    o1, _ = fast.reset(seed=2)
    o2, _ = ref.reset(seed=2)
    bufs = fast.action_buffers()
    assert set(bufs) == set(fast.possible_agents) and all(v.shape == (E, 4) for v in bufs.values())
    rng = np.random.default_rng(1)
    ended = 0
    for k in range(60):
        acts = {a: np.concatenate([rng.uniform(-1, 1, size=(E, 3)), rng.uniform(0.2, 0.75, size=(E, 1))], 1).astype(np.float32) for a in fast.possible_agents}
        for a in fast.possible_agents:
            bufs[a].copy_(torch.from_numpy(acts[a]))
        f = fast.step(bufs)
        # the reference zeroes a culled agent's action (ma_quadx_base_env.py:329): the culling env is fed its live agents only, the
        # other one is handed zeros for the same agents so that the two worlds stay the same
        live = list(ref.agents)
        for a in fast.possible_agents:
            if a not in live:
                bufs[a].zero_()
        if len(live) < len(fast.possible_agents):
            f = None  # (this step's actions differed for the culled agents: compare from the next step on)
        r = ref.step({a: acts[a] for a in live})
        if f is not None:
            for a in live:
                for x, y in zip(f[:4], r[:4]):
                    assert torch.equal(x[a], y[a]), (k, a)
                assert torch.equal(f[4][a]["collision"], r[4][a]["collision"]) and torch.equal(f[4][a]["out_of_bounds"], r[4][a]["out_of_bounds"])
                fl = fast._split(fast.engine.flags())[fast.agent_name_mapping[a]]
                assert torch.equal(f[4][a]["collision"], (fl & L.F_INFO_COLLISION) != 0)
                ended += int((r[2][a] | r[3][a]).sum())
        assert fast.agents == fast.possible_agents
        if not ref.agents:
            break
    assert ended > 0 and len(ref.agents) < len(ref.possible_agents)  # (max_duration 1 s: every copy truncates together at step 41)
    fast.close(); ref.close()


def test_ma_step_captured_in_a_hip_graph():
    """The PettingZoo step with cull_agents=False inside a HIP graph (policy -> action_buffers -> step): replays equal the eager loop."""
    from pyflyt_amd.pz_envs import MAQuadXHoverEnv

    E, g = 256, 10
    envs = [MAQuadXHoverEnv(num_envs=E, seed=5, cull_agents=False) for _ in range(2)]
    for e in envs:
        e.reset(seed=5)
    W = torch.zeros(24, 4, device="cuda")
    W[0, 0] = W[1, 1] = W[2, 2] = -0.2
    W[12, 3] = -0.3
    b = torch.tensor([0.0, 0.0, 0.0, 0.68], device="cuda")

    def loop(e, k):
        bufs = e.action_buffers()
        for _ in range(k):
            torch.addmm(b, e.engine.obs, W, out=e._act_flat)
            out = e.step(bufs)
        return out

    loop(envs[0], 1); loop(envs[1], 1)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        stream.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            loop(envs[0], g)
        stream.synchronize()
    for r in range(4):
        graph.replay()
        o, rw, t, u, info = loop(envs[1], g)
        torch.cuda.synchronize()
        assert torch.equal(envs[0].engine.obs, envs[1].engine.obs) and torch.equal(envs[0].engine.reward, envs[1].engine.reward)
        assert torch.isfinite(o["uav_0"]).all()
    for e in envs:
        e.close()
