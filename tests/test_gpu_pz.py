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
