"""pf_rollout (k env steps per launch, state resident in registers, actions sampled on device) must be
BIT-IDENTICAL to k x (pf_sample_actions + pf_env_step): same Philox keys, same arithmetic, every step's
observation / reward / flags written. Also against the fp64 oracle over the same action sequence."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


def _engine(task, n, noise, autoreset, seed, lane_offset=0, **kw):
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    vehicle = "quadx"
    if task.startswith("fixedwing_"):
        vehicle, task = "fixedwing", task[len("fixedwing_"):]
    P = build_params(vehicle, task, noise=noise, autoreset=autoreset, seed=seed, **kw)
    return BatchEngine(P, n, device="cuda:0", lane_offset=lane_offset)


@pytest.mark.parametrize("task", ["hover", "waypoints", "fixedwing_waypoints"])
@pytest.mark.parametrize("noise,autoreset", [("philox", "next_step"), ("philox", "same_step"), ("off", "next_step")])
def test_rollout_bit_identical_to_single_steps(task, noise, autoreset):
    n, k, seed = 1000, 96, 21  # 1000: a ragged last wave
    if task == "fixedwing_waypoints":
        k = 300  # the plane's episodes are longer: enough steps for a few hundred of them to end inside the rollouts
    a = _engine(task, n, noise, autoreset, seed, lane_offset=4096)
    b = _engine(task, n, noise, autoreset, seed, lane_offset=4096)
    a.env_reset()
    b.env_reset()
    assert torch.equal(a.state, b.state)
    # two launches of k/2 steps each: the state written back by the first must continue exactly
    chunks = []
    for c in range(2):
        obs, rew, term, trunc, acts = a.rollout(k // 2, step_index0=c * (k // 2))
        chunks.append([x.clone() for x in (obs, rew, term, trunc, acts)] +
                      [a._traj["final_obs"].clone() if a.final_obs is not None else None,
                       a._traj["final_info"].clone() if a.final_info is not None else None])
    act = torch.empty(n, 4, device="cuda:0")
    n_done = 0
    for s in range(k):
        c, j = divmod(s, k // 2)
        b.sample_actions(act, s)
        o, r, t, tr = b.env_step(act)
        assert torch.equal(act, chunks[c][4][j]), f"step {s}: sampled action"
        assert torch.equal(o, chunks[c][0][j]), f"step {s}: obs max diff {(o - chunks[c][0][j]).abs().max().item()}"
        assert torch.equal(r, chunks[c][1][j]), f"step {s}: reward"
        assert torch.equal(t, chunks[c][2][j]) and torch.equal(tr, chunks[c][3][j]), f"step {s}: flags"
        done = t | tr
        n_done += int(done.sum())
        if autoreset == "same_step" and done.any():
            assert torch.equal(b.final_obs[done], chunks[c][5][j][done]), f"step {s}: final_obs"
            assert torch.equal(b.final_info[done], chunks[c][6][j][done]), f"step {s}: final_info"
    assert torch.equal(a.state, b.state)
    print(f"{task}: {n_done} episode ends inside the rollouts")
    assert n_done > 200  # random actions end episodes quickly: the in-loop resets are exercised


@pytest.mark.parametrize("mode", [-1, 4, 7])
def test_rollout_bit_identical_cascaded_modes(mode):
    """The cascaded-PID instantiation (flight modes other than 0) of the specialised kernel: pf_rollout == k x pf_env_step,
    controller memories (state groups 7-11) included."""
    n, k, seed = 1000, 96, 33
    # (1 s episodes: every lane goes through the in-loop reset -- the settle recurrence with the mode's z PIDs -- twice)
    a = _engine("hover", n, "philox", "next_step", seed, lane_offset=512, flight_mode=mode, max_duration_seconds=1.0)
    b = _engine("hover", n, "philox", "next_step", seed, lane_offset=512, flight_mode=mode, max_duration_seconds=1.0)
    assert a.lib.pf_ctx_is_specialised(a._ctx) == 1
    a.env_reset(); b.env_reset()
    assert torch.equal(a.state, b.state)
    obs, rew, term, trunc, acts = a.rollout(k)
    act = torch.empty(n, 4, device="cuda:0")
    n_done = 0
    for s in range(k):
        b.sample_actions(act, s)
        o, r, t, tr = b.env_step(act)
        assert torch.equal(act, acts[s]) and torch.equal(o, obs[s]) and torch.equal(r, rew[s]), (mode, s, (o - obs[s]).abs().max().item())
        assert torch.equal(t, term[s]) and torch.equal(tr, trunc[s]), (mode, s)
        n_done += int((t | tr).sum())
    assert torch.equal(a.state, b.state)
    assert n_done > 1500, n_done


def test_rollout_shared_world_ma_hover():
    """PettingZoo task, one world per 4 agents, no auto-reset: a rollout over a given action sequence == k x pf_env_step."""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    n, k, A = 256, 50, 4
    eng = []
    for _ in range(2):
        P = build_params("quadx", "ma_hover", noise="philox", autoreset="off", seed=3, agents_per_world=A, flight_dome_size=3.0,
                         world_options=dict(contact_response=True))
        e = BatchEngine(P, n, device="cuda:0")
        assert e.lib.pf_ctx_is_specialised(e._ctx) == 1
        # spawn poses: four agents 30 cm apart, level (state groups 12-14: position, quaternion)
        pos = torch.tensor([[-0.15, 0.0, 1.0], [0.15, 0.0, 1.02], [0.0, 0.3, 1.0], [0.0, -0.3, 0.6]], device="cuda:0").repeat(n // A, 1)
        e.state[12, :, 0:3] = pos
        e.state[12, :, 3] = 0.0; e.state[13, :, 0] = 0.0; e.state[13, :, 1] = 0.0; e.state[13, :, 2] = 1.0
        e.env_reset()
        eng.append(e)
    a, b = eng
    assert torch.equal(a.state, b.state)
    rng = np.random.default_rng(1)
    seq = torch.tensor(rng.uniform([-1, -1, -1, 0.1], [1, 1, 1, 0.7], size=(k, n, 4)), dtype=torch.float32, device="cuda:0")
    obs, rew, term, trunc, _ = a.rollout(k, actions=seq)
    hits = 0
    for s in range(k):
        o, r, t, tr = b.env_step(seq[s].contiguous())
        assert torch.equal(o, obs[s]) and torch.equal(r, rew[s]) and torch.equal(t, term[s]) and torch.equal(tr, trunc[s]), s
        hits += int(t.sum())
    assert torch.equal(a.state, b.state)
    assert hits > 0  # (drone-drone or floor contacts occurred)


def test_rollout_given_action_sequence():
    """An open-loop action sequence [k, n, 4] instead of on-device sampling."""
    n, k = 320, 40
    a = _engine("hover", n, "philox", "next_step", 5)
    b = _engine("hover", n, "philox", "next_step", 5)
    a.env_reset(); b.env_reset()
    rng = np.random.default_rng(0)
    seq = torch.tensor(rng.uniform([-3, -3, -3, 0], [3, 3, 3, 0.8], size=(k, n, 4)), dtype=torch.float32, device="cuda:0")
    obs, rew, term, trunc, _ = a.rollout(k, actions=seq)
    for s in range(k):
        o, r, t, tr = b.env_step(seq[s].contiguous())
        assert torch.equal(o, obs[s]) and torch.equal(r, rew[s]) and torch.equal(t, term[s]) and torch.equal(tr, trunc[s]), s
    assert torch.equal(a.state, b.state)


def test_rollout_given_action_sequence_fixedwing():
    n, k = 320, 60
    a = _engine("fixedwing_waypoints", n, "philox", "next_step", 5)
    b = _engine("fixedwing_waypoints", n, "philox", "next_step", 5)
    a.env_reset(); b.env_reset()
    rng = np.random.default_rng(0)
    seq = torch.tensor(rng.uniform([-1, -1, -1, 0], [1, 1, 1, 1], size=(k, n, 4)), dtype=torch.float32, device="cuda:0")
    obs, rew, term, trunc, _ = a.rollout(k, actions=seq)
    for s in range(k):
        o, r, t, tr = b.env_step(seq[s].contiguous())
        assert torch.equal(o, obs[s]) and torch.equal(r, rew[s]) and torch.equal(t, term[s]) and torch.equal(tr, trunc[s]), s
    assert torch.equal(a.state, b.state)


def test_rollout_against_oracle():
    """The rollout's trajectory against the fp64 oracle fed the same (device-sampled) actions."""
    n, k, seed = 512, 60, 9
    a = _engine("hover", n, "philox", "next_step", seed)
    a.env_reset()
    obs, rew, term, trunc, acts = a.rollout(k)
    orc = O.OracleBatch(O.make_params("hover", noise_mode=O.NOISE_PHILOX, seed=seed), n)
    orc.reset()
    ok = np.ones(n, dtype=bool)
    worst = 0.0
    for s in range(k):
        ro, rr, rt, rtr, _ = orc.step(acts[s].cpu().numpy(), autoreset=1)
        e = (np.abs(obs[s].cpu().numpy().astype(np.float64) - ro) / np.maximum(1.0, np.abs(ro))).max(axis=1)
        ok &= (e < 1e-4) & (term[s].cpu().numpy() == rt) & (trunc[s].cpu().numpy() == rtr)
        worst = max(worst, e[ok].max())
    print(f"rollout vs oracle: worst {worst:.2e}, lanes compared to the end {ok.mean():.4f}")
    assert ok.all() and worst < 1e-4


def test_rollout_refusals():
    from pyflyt_amd import PyFlytAmdError

    e = _engine("hover", 64, "philox", "off", 1)
    e.env_reset()
    with pytest.raises(PyFlytAmdError):
        e.rollout(4)  # no auto-reset mode
    e = _engine("hover", 64, "inject", "next_step", 1)
    e.env_reset(xi_reset=torch.zeros(e.settle_ticks, 64, device="cuda:0"))
    with pytest.raises(PyFlytAmdError):
        e.rollout(4)  # injected noise is a per-step protocol


@pytest.mark.parametrize("given", [False, True])
def test_rollout_outside_the_specialised_kernels(given):
    """pf_rollout on a configuration only the generic env kernel runs (a cascaded flight mode with the contact response opted
    out): one launch, the state resident in registers (env_kernel's roll_steps) -- the trajectory layout and the results of
    k x (pf_sample_actions + pf_env_step), bit for bit, with or without the sampled actions written out."""
    n, k = 200, 12
    kw = dict(flight_mode=6, world_options=dict(contact_response=False), max_duration_seconds=0.2)
    a, b = (_engine("hover", n, "philox", "next_step", 3, **kw) for _ in range(2))
    assert a.lib.pf_ctx_is_specialised(a._ctx) == 0
    a.env_reset(); b.env_reset()
    seq = None
    if given:
        seq = torch.empty(k, n, 4, device="cuda:0")
        for s in range(k):
            a.sample_actions(seq[s], 100 + s)
    obs, rew, term, trunc, acts = a.rollout(k, step_index0=100, actions=seq)
    act = torch.empty(n, 4, device="cuda:0")
    n_done = 0
    for s in range(k):
        b.sample_actions(act, 100 + s)
        o, r, t, u = b.env_step(act)
        assert torch.equal(acts[s], act) and torch.equal(obs[s], o) and torch.equal(rew[s], r), s
        assert torch.equal(term[s], t) and torch.equal(trunc[s], u), s
        n_done += int((t | u).sum())
    assert torch.equal(a.state, b.state) and n_done > 0
    if not given:  # ... and without keeping the draws
        obs2, *_rest, none = a.rollout(4, step_index0=900, store_actions=False)
        assert none is None
        for s in range(4):
            b.sample_actions(act, 900 + s)
            assert torch.equal(obs2[s], b.env_step(act)[0]), s


GENERIC_CASES = {
    # every env task on the GENERIC env kernel (PF_DISABLE_FAST: the switch the fixture tests use to replay on both kernels)
    "hover_next_step": ("hover", "philox", "next_step", dict(), 96, 200),
    "hover_same_step": ("hover", "philox", "same_step", dict(), 96, 200),
    "hover_noise_off_mode7": ("hover", "off", "next_step", dict(flight_mode=7, max_duration_seconds=1.0), 64, 100),
    "waypoints_yaw_targets_same_step": ("waypoints", "philox", "same_step", dict(use_yaw_targets=True, goal_reach_distance=0.6, goal_reach_angle=3.0), 96, 100),
    "waypoints_mode4_quaternion": ("waypoints", "philox", "next_step", dict(flight_mode=4, angle_representation="quaternion", max_duration_seconds=1.0), 64, 100),
    "fixedwing_waypoints": ("fixedwing_waypoints", "philox", "next_step", dict(), 300, 100),
}


@pytest.mark.parametrize("case", sorted(GENERIC_CASES))
def test_rollout_generic_kernel_resident(case, monkeypatch):
    """env_kernel's roll_steps (state-resident pf_rollout on the generic kernel) against k x (pf_sample_actions + pf_env_step) on the
    same kernel, bit for bit: through in-loop NEXT_STEP / SAME_STEP resets (final_obs / final_info slots), reached waypoints and yaw
    targets, the cascaded modes' controller memories, crashes with the contact solve, in two launches that must continue each other,
    with a ragged last wave."""
    task, noise, autoreset, kw, k, min_done = GENERIC_CASES[case]
    monkeypatch.setenv("PF_DISABLE_FAST", "1")
    n, seed = 1000, 77
    a = _engine(task, n, noise, autoreset, seed, lane_offset=128, **kw)
    b = _engine(task, n, noise, autoreset, seed, lane_offset=128, **kw)
    assert a.lib.pf_ctx_is_specialised(a._ctx) == 0
    a.env_reset(); b.env_reset()
    assert torch.equal(a.state, b.state)
    chunks = []
    for c in range(2):
        obs, rew, term, trunc, acts = a.rollout(k // 2, step_index0=c * (k // 2))
        chunks.append([x.clone() for x in (obs, rew, term, trunc, acts)] +
                      [a._traj["final_obs"].clone() if a.final_obs is not None else None,
                       a._traj["final_info"].clone() if a.final_info is not None else None])
    act = torch.empty(n, 4, device="cuda:0")
    n_done = 0
    for s in range(k):
        c, j = divmod(s, k // 2)
        b.sample_actions(act, s)
        o, r, t, tr = b.env_step(act)
        assert torch.equal(act, chunks[c][4][j]), f"step {s}: sampled action"
        assert torch.equal(o, chunks[c][0][j]), f"step {s}: obs max diff {(o - chunks[c][0][j]).abs().max().item()}"
        assert torch.equal(r, chunks[c][1][j]), f"step {s}: reward"
        assert torch.equal(t, chunks[c][2][j]) and torch.equal(tr, chunks[c][3][j]), f"step {s}: flags"
        done = t | tr
        n_done += int(done.sum())
        if autoreset == "same_step" and done.any():
            assert torch.equal(b.final_obs[done], chunks[c][5][j][done]), f"step {s}: final_obs"
            assert torch.equal(b.final_info[done], chunks[c][6][j][done]), f"step {s}: final_info"
    assert torch.equal(a.state, b.state)
    print(f"{case}: {n_done} episode ends inside the rollouts")
    assert n_done > min_done


def test_rollout_generic_kernel_shared_world(monkeypatch):
    """... and the PettingZoo task in a shared world on the generic kernel: exchange arrays, pair stage and the per-call flags, resident
    over a given action sequence."""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    monkeypatch.setenv("PF_DISABLE_FAST", "1")
    n, k, A = 256, 50, 4
    eng = []
    for _ in range(2):
        P = build_params("quadx", "ma_hover", noise="philox", autoreset="off", seed=3, agents_per_world=A, flight_dome_size=3.0,
                         world_options=dict(contact_response=True))
        e = BatchEngine(P, n, device="cuda:0")
        assert e.lib.pf_ctx_is_specialised(e._ctx) == 0
        pos = torch.tensor([[-0.15, 0.0, 1.0], [0.15, 0.0, 1.02], [0.0, 0.3, 1.0], [0.0, -0.3, 0.6]], device="cuda:0").repeat(n // A, 1)
        e.state[12, :, 0:3] = pos
        e.state[12, :, 3] = 0.0; e.state[13, :, 0] = 0.0; e.state[13, :, 1] = 0.0; e.state[13, :, 2] = 1.0
        e.env_reset()
        eng.append(e)
    a, b = eng
    assert torch.equal(a.state, b.state)
    rng = np.random.default_rng(1)
    seq = torch.tensor(rng.uniform([-1, -1, -1, 0.1], [1, 1, 1, 0.7], size=(k, n, 4)), dtype=torch.float32, device="cuda:0")
    obs, rew, term, trunc, _ = a.rollout(k, actions=seq)
    hits = 0
    for s in range(k):
        o, r, t, tr = b.env_step(seq[s].contiguous())
        assert torch.equal(o, obs[s]) and torch.equal(r, rew[s]) and torch.equal(t, term[s]) and torch.equal(tr, trunc[s]), s
        hits += int(t.sum())
    assert torch.equal(a.state, b.state)
    assert hits > 0


def test_rollout_dogfight_generic_aircraft(monkeypatch):
    """The dogfight's resident rollout on the GENERIC aircraft model (dogfight_env_kernel<A, DfGenericVeh, ROLLOUT>): equal to k x
    pf_env_step through crashes, wrecks and the pair stage; six-wide actions as a given sequence."""
    from pyflyt_amd import PyFlytAmdError, build_params
    from pyflyt_amd.engine import BatchEngine

    monkeypatch.setenv("PF_DISABLE_FAST", "1")

    def make(**kw):
        P = build_params("fixedwing", "dogfight", noise="philox", autoreset="off", seed=5, angle_representation="euler",
                         vehicle_options=dict(drone_model="acrowing"), world_options=dict(world_scale=5.0), dogfight=dict(sample_spawn=True, **kw))
        e = BatchEngine(P, 4 * 50, device="cuda:0")
        assert e.lib.pf_ctx_is_specialised(e._ctx) == 0
        e.env_reset()
        return e

    a, b = make(), make()
    k = 100
    obs, rew, term, trunc, acts = a.rollout(k, step_index0=7)
    ref = torch.empty(4 * 50, 4, device="cuda:0")
    for s in range(k):
        b.sample_actions(ref, 7 + s)
        assert torch.equal(acts[s], ref), s
        o, r, t, u = b.env_step(ref)
        assert torch.equal(obs[s], o) and torch.equal(rew[s], r) and torch.equal(term[s], t) and torch.equal(trunc[s], u), s
    assert torch.equal(a.state, b.state)
    assert bool(term.any())  # aircraft went down inside the launch
    # six-wide actions (assisted_flight off): not sampled on device, resident over a given sequence
    c, d = make(assisted_flight=False), make(assisted_flight=False)
    with pytest.raises(PyFlytAmdError):
        c.rollout(3)
    rng = np.random.default_rng(2)
    seq = torch.tensor(rng.uniform(-1.0, 1.0, size=(20, 4 * 50, 6)), dtype=torch.float32, device="cuda:0")
    obs6, rew6, term6, trunc6, _ = c.rollout(20, actions=seq)
    for s in range(20):
        o, r, t, u = d.env_step(seq[s].contiguous())
        assert torch.equal(obs6[s], o) and torch.equal(rew6[s], r) and torch.equal(term6[s], t) and torch.equal(trunc6[s], u), s
    assert torch.equal(c.state, d.state)


def test_rollout_dogfight():
    """The PettingZoo loop of tests/test_pz_envs.py:71-93 in one call for the dogfight task: k env steps of every world, equal to
    k x pf_env_step on the same sampled actions."""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    def make():
        P = build_params("fixedwing", "dogfight", noise="philox", autoreset="off", seed=5, angle_representation="euler",
                         vehicle_options=dict(drone_model="acrowing"), world_options=dict(world_scale=5.0), dogfight=dict(sample_spawn=True))
        e = BatchEngine(P, 4 * 50, device="cuda:0")
        e.env_reset()
        return e

    a, b = make(), make()
    k = 10
    obs, rew, term, trunc, acts = a.rollout(k, step_index0=7)
    ref = torch.empty(4 * 50, 4, device="cuda:0")
    for s in range(k):
        b.sample_actions(ref, 7 + s)  # the resident kernel samples with pf_sample_actions' keys
        assert torch.equal(acts[s], ref), s
        o, r, t, u = b.env_step(acts[s].contiguous())
        assert torch.equal(obs[s], o) and torch.equal(rew[s], r) and torch.equal(term[s], t) and torch.equal(trunc[s], u), s
    assert torch.equal(a.state, b.state)
    # ... through a longer flight (aircraft reach the ground: contact solve, wrecks, the pair stage's exchange arrays reused step
    # after step inside one launch), over a GIVEN action sequence, and without keeping the sampled actions
    k2 = 120
    seq = torch.empty(k2, 4 * 50, 4, device="cuda:0")
    for s in range(k2):
        b.sample_actions(seq[s], 1000 + s)
    obs2, rew2, term2, trunc2, _ = a.rollout(k2, actions=seq)
    for s in range(k2):
        o, r, t, u = b.env_step(seq[s].contiguous())
        assert torch.equal(obs2[s], o) and torch.equal(rew2[s], r) and torch.equal(term2[s], t) and torch.equal(trunc2[s], u), s
    assert torch.equal(a.state, b.state)
    obs3, *_rest, none = a.rollout(5, step_index0=5000, store_actions=False)
    assert none is None
    for s in range(5):
        b.sample_actions(ref, 5000 + s)
        assert torch.equal(obs3[s], b.env_step(ref)[0]), s
