"""The calm-wave path of the specialised QuadX kernel (quadx_fast.hpp: waves none of whose lanes can reach the floor during the
env step run ticks instantiated without the contact response's call site) and its one-wave-per-SIMD instantiation (WPS = 1: the
solve inlined, chosen for batches of at most one wave per SIMD) change nothing: bit-identical to the same context with both
switched off (PF_NO_CALM_PATH, PF_NO_LEAN_KERNEL, read at context creation), through crashes, resets and the lanes in between."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lean", [True, False])
@pytest.mark.parametrize("task,mode", [("hover", 0), ("waypoints", 0), ("hover", 6), ("hover", 7)])
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
        eng = BatchEngine(build_params("quadx", task, noise="philox", autoreset="next_step", seed=11, **kw), n, device="cuda:0")
        assert eng.lib.pf_ctx_is_specialised(eng._ctx) != 0
        return eng

    a, b = make(True, lean), make(False, False)
    oa, ob = a.env_reset().clone(), b.env_reset().clone()
    assert torch.equal(oa, ob)
    act = torch.empty(n, 4, device="cuda:0")
    sink = torch.arange(n, device="cuda:0") % 5 == 0  # a fifth of the lanes is told to drop: floor contacts in many waves, and waves without
    collided = 0
    for k in range(steps):
        a.sample_actions(act, k)
        if mode == 0:
            act[sink, 3] = -1.0  # thrust command at the bottom of the box
        elif mode == 6:
            act[sink, 3] = -3.0  # (vx, vy, vr, vz): descend
        else:
            act[sink, 3] = 0.0   # (x, y, r, z): go to the floor
        ra, rb = a.env_step(act), b.env_step(act)
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (task, mode, k)
        assert torch.equal(a.state, b.state), (task, mode, k)
        collided += int(((a.flags() & L.F_INFO_COLLISION) != 0).sum())
    assert collided > 20, collided  # (the floor was in play)
