"""The calm-wave path of the specialised QuadX kernel (quadx_fast.hpp: waves none of whose lanes can reach the floor during the
env step, or during the Aviary step at hand, run ticks instantiated without the contact response's call site) changes nothing:
bit-identical to the same context with it switched off (PF_NO_CALM_PATH, read at context creation), through crashes, resets and
the lanes in between -- in the one-wave-per-SIMD instantiation (WPS = 1, chosen for batches of at most one wave per SIMD) and in
the two-wave one (PF_NO_LEAN_KERNEL). The two instantiations solve a floor contact with differently arranged arithmetic (WPS = 1:
in registers, quad_floor_solve; WPS = 2: the general solver out of line): against each other they are bit-identical everywhere
except in the observation of the step that reports the collision -- the episode ends there and nothing carries over."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lean", [True, False])
@pytest.mark.parametrize("task,mode", [("hover", 0), ("waypoints", 0), ("ma_hover", 0), ("hover", 6), ("hover", 7)])
def test_calm_path_is_bit_identical(monkeypatch, task, mode, lean):
    from pyflyt_amd import _lib as L
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    n, steps = 4096 + 37, 160
    kw = dict(flight_mode=mode) if mode else {}

    def make(calm, lean):
        for var, on in (("PF_NO_CALM_PATH", calm), ("PF_NO_LEAN_KERNEL", lean)):
            if on:
                monkeypatch.delenv(var, raising=False)
            else:
                monkeypatch.setenv(var, "1")
        # (the PettingZoo task has no auto-reset: its crashed drones stay on the floor, their waves leave the calm path and come back)
        eng = BatchEngine(build_params("quadx", task, noise="philox", autoreset="off" if task == "ma_hover" else "next_step", seed=11, **kw), n, device="cuda:0")
        assert eng.lib.pf_ctx_is_specialised(eng._ctx) != 0
        if task == "ma_hover":  # the agents' spawn poses live in the state's side block (pz_envs/ma_quadx_hover.py writes them; bench.py: make_engine)
            side = torch.zeros(n, 12, device="cuda:0")
            side[:, 2], side[:, 6] = 1.0, 1.0  # (0, 0, 1), level: quaternion (0, 0, 0, 1)
            eng.state[12:15] = side.view(n, 3, 4).permute(1, 0, 2)
        return eng

    a, b, c = make(True, lean), make(False, lean), make(False, not lean)
    oa, ob, oc = a.env_reset().clone(), b.env_reset().clone(), c.env_reset().clone()
    assert torch.equal(oa, ob) and torch.equal(oa, oc)
    act = torch.empty(n, 4, device="cuda:0")
    sink = torch.arange(n, device="cuda:0") % 5 == 0  # a fifth of the lanes is told to drop: floor contacts in many waves, and waves without
    collided, worst = 0, 0.0
    for k in range(steps):
        a.sample_actions(act, k)
        if mode == 0:
            act[sink, 3] = -1.0  # thrust command at the bottom of the box
        elif mode == 6:
            act[sink, 3] = -3.0  # (vx, vy, vr, vz): descend
        else:
            act[sink, 3] = 0.0   # (x, y, r, z): go to the floor
        ra, rb, rc = a.env_step(act), b.env_step(act), c.env_step(act)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (task, mode, k)
        assert torch.equal(a.state, b.state), (task, mode, k)
        hit = (a.flags() & L.F_INFO_COLLISION) != 0
        if task == "ma_hover":  # (no reset: after its first floor contact a lane's state differs between the two instantiations' solves for good --
            collided += int(hit.sum())  # the calm path against no calm path, above, is this task's whole statement)
            continue
        assert torch.equal(hit, (c.flags() & L.F_INFO_COLLISION) != 0)
        for x, y in zip(ra[2:], rc[2:]):   # terminated, truncated: identical
            assert torch.equal(x, y), (task, mode, k)
        # every lane that did not just crash: bit-identical observation and reward
        assert torch.equal(ra[0][~hit], rc[0][~hit]) and torch.equal(ra[1][~hit], rc[1][~hit]), (task, mode, k)
        if bool(hit.any()):  # (the dense reward terms are evaluated on the terminal state too)
            worst = max(worst, float((ra[0][hit] - rc[0][hit]).abs().max()), float((ra[1][hit] - rc[1][hit]).abs().max()))
        collided += int(hit.sum())
    assert collided > 20, collided  # (the floor was in play)
    assert task == "ma_hover" or worst < 5e-3, worst      # the two instantiations' contact solves agree to the impact tolerance in that terminal observation


def test_fixedwing_instantiations_are_bit_identical(monkeypatch):
    """The Fixedwing-Waypoints kernel's two instantiations -- one wave per SIMD (512 registers, the constant table in vector registers,
    fetched through LDS, both surface pairs evaluated side by side) and two (256 registers, the table's rows by scalar loads in the
    tick; PF_NO_LEAN_KERNEL, what batches beyond one wave per SIMD get) -- run the same arithmetic per element: bit-identical
    observations, rewards, flags and state, through resets and out-of-bounds endings."""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    n, steps = 4096 + 37, 120

    def make(lean):
        if lean:
            monkeypatch.delenv("PF_NO_LEAN_KERNEL", raising=False)
        else:
            monkeypatch.setenv("PF_NO_LEAN_KERNEL", "1")
        eng = BatchEngine(build_params("fixedwing", "waypoints", noise="philox", autoreset="next_step", seed=5), n, device="cuda:0")
        assert eng.lib.pf_ctx_is_specialised(eng._ctx) != 0
        return eng

    a, b = make(True), make(False)
    assert torch.equal(a.env_reset(), b.env_reset())
    act = torch.empty(n, 4, device="cuda:0")
    ended = 0
    for k in range(steps):
        a.sample_actions(act, k)
        ra, rb = a.env_step(act), b.env_step(act)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), k
        assert torch.equal(a.state, b.state), k
        ended += int((ra[2] | ra[3]).sum())
    assert ended > 50, ended  # (episodes ended and restarted along the way)
    # ... and state-resident: sixty more steps in one launch on either side
    ta, tb = a.rollout(60), b.rollout(60)
    for x, y in zip(ta, tb):
        assert torch.equal(x, y)
    assert torch.equal(a.state, b.state)
