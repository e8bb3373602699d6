"""The spares of the specialised QuadX kernel (pyflyt_amd/csrc/quadx_fast.hpp: QuadSpare): for the Hover / Waypoints tasks the random
part of a lane's NEXT episode -- the settled spawn state, the waypoints -- is keyed by the event counter at the lane's previous
reset, prepared ahead for all lanes of a wave every kSpareEvery env steps (state groups 7-11), and copied by the reset. What a
reset produces must not depend on WHEN its draws were turned into numbers: a context whose spares are invalidated before every
step (every reset then generates on the spot, as the kernels did before round 5 and as the generic kernel, the cascaded flight
modes and the oracle still do) steps bit-identically to one that uses them; so does the generic kernel's key bookkeeping."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
VALID = np.int32(-2**31)  # bit 31 of the key word


@pytest.mark.parametrize("task", ["hover", "waypoints"])
@pytest.mark.parametrize("autoreset", ["next_step", "same_step"])
def test_a_reset_from_a_spare_is_the_reset_without_one(task, autoreset):
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    n, steps = 64 * 9 + 17, 120
    mk = lambda: BatchEngine(build_params("quadx", task, noise="philox", autoreset=autoreset, seed=3), n, device="cuda:0")  # noqa: E731
    a, b = mk(), mk()
    assert a.lib.pf_ctx_is_specialised(a._ctx) != 0
    oa, ob = a.env_reset().clone(), b.env_reset().clone()
    assert torch.equal(oa, ob)
    kw = a.state[7, :, 3].view(torch.int32)
    assert bool((kw < 0).all())  # an explicit reset leaves every lane with a prepared spare
    act = torch.empty(n, 4, device="cuda:0")
    used = ends = 0
    keep = [g for g in range(a.state.shape[0]) if g < 7 or g > 11]  # everything but the spare groups
    for k in range(steps):
        a.sample_actions(act, k)
        b.state[7, :, 3].view(torch.int32).bitwise_and_(0x7FFFFFFF)  # b: no lane has a spare -> every reset generates from its key
        done = (a.flags() & 3) != 0
        used += int(((a.state[7, :, 3].view(torch.int32) < 0) & done).sum())
        ra, rb = a.env_step(act), b.env_step(act)
        ends += int((ra[2] | ra[3]).sum())
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), (task, autoreset, k)
        assert torch.equal(a.state[keep], b.state[keep]), (task, autoreset, k)
        # the key itself (bits 0-30) is the same on both sides
        assert torch.equal(a.state[7, :, 3].view(torch.int32) & 0x7FFFFFFF, b.state[7, :, 3].view(torch.int32) & 0x7FFFFFFF)
    assert ends > 5 * n // 2
    if autoreset == "next_step":  # (the lanes that restarted did so from a prepared spare: episodes outlast the refill period)
        assert used >= 0.98 * (ends - int(((a.flags() & 3) != 0).sum())), (used, ends)


def test_the_generic_kernel_keeps_the_key_and_agrees_with_the_specialised_one(monkeypatch):
    """PF_DISABLE_FAST: the generic env kernel carries the same key word (no spares) -- observations of both kernels agree to fp32
    rounding through resets, i.e. both drew the same settle noise and the same waypoints for every episode."""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    n, steps = 256, 90
    P = build_params("quadx", "waypoints", noise="philox", autoreset="next_step", seed=8)
    a = BatchEngine(P, n, device="cuda:0")
    monkeypatch.setenv("PF_DISABLE_FAST", "1")
    g = BatchEngine(P, n, device="cuda:0")
    assert a.lib.pf_ctx_is_specialised(a._ctx) != 0 and g.lib.pf_ctx_is_specialised(g._ctx) == 0
    assert float((a.env_reset() - g.env_reset()).abs().max()) < 1e-5
    act = torch.empty(n, 4, device="cuda:0")
    ends = 0
    ok = torch.ones(n, dtype=torch.bool, device="cuda:0")
    for k in range(steps):
        a.sample_actions(act, k)
        oa, ra, ta, ua = a.env_step(act)
        og, rg, tg, ug = g.env_step(act)
        same = (ta == tg) & (ua == ug)
        ok &= same  # (a lane whose episode end moved by a step between the two fp32 kernels leaves the comparison)
        fresh = ok & (((a.flags() & 3) == 0) & (a.ints()[:, 0] == 0))  # lanes that restarted in this call: their reset observation
        if bool(fresh.any()):
            assert float((oa[fresh] - og[fresh]).abs().max()) < 1e-4, k  # same settle noise, same targets
            ends += int(fresh.sum())
        assert torch.equal(a.state[7, ok, 3].view(torch.int32) & 0x7FFFFFFF, g.state[7, ok, 3].view(torch.int32) & 0x7FFFFFFF), k
    assert ends > n and float(ok.float().mean()) > 0.97
