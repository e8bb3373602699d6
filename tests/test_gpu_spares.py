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


@pytest.mark.parametrize("task", ["hover", "waypoints"])
@pytest.mark.parametrize("kernel", ["specialised", "generic"])
def test_consecutive_resets_are_different_episodes_and_equal_the_oracles(monkeypatch, task, kernel):
    """Round 5's key (the counter AT the previous reset: 0 again at a fresh lane's first reset) gave every lane the same settle noise
    and the same waypoints for its first and second episodes, on the device and in the oracle alike, so that no parity test could see
    it. The key is now what the previous reset LEFT BEHIND: five consecutive explicit resets -- with 0, 1, 2 ... env steps
    in between -- are five different episodes for every lane, the key words grow, and each reset observation equals the oracle's."""
    from oracle import oracle as O
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    if kernel == "generic":
        monkeypatch.setenv("PF_DISABLE_FAST", "1")
    n = 64 * 3 + 5
    eng = BatchEngine(build_params("quadx", task, noise="philox", autoreset="off", seed=3), n, device="cuda:0")
    assert (eng.lib.pf_ctx_is_specialised(eng._ctx) != 0) == (kernel == "specialised")
    orc = O.OracleBatch(O.make_params("hover" if task == "hover" else "quadx_waypoints", noise_mode=O.NOISE_PHILOX, seed=3), n)
    act = torch.empty(n, 4, device="cuda:0")
    seen, keys = [], []
    for r in range(5):
        got = eng.env_reset().cpu().numpy().astype(np.float64)
        ref = orc.reset()
        assert np.abs(got - ref).max() < 2e-5, (r, np.abs(got - ref).max())
        seen.append(got)
        keys.append((eng.state[7, :, 3].view(torch.int32) & 0x7FFFFFFF).cpu().numpy().astype(np.int64))
        assert (keys[-1] == np.array([orc.lanes[i].reset_key for i in range(n)])).all()
        for k in range(r):
            eng.sample_actions(act, 10 * r + k)
            act[:, :3] *= 0.05
            act[:, 3] = 0.4
            eng.env_step(act)
            orc.step(act.cpu().numpy())
    for a in range(5):
        for b in range(a + 1, 5):
            assert (np.abs(seen[a] - seen[b]).max(axis=1) > 1e-7).all(), (a, b)
    assert all((keys[r + 1] > keys[r]).all() for r in range(4)) and (keys[0] == 1).all()


def test_a_full_reset_does_not_trust_foreign_spares():
    """A state buffer that another context has used carries that context's spares with bit 31 set (other seed: other settle noise,
    other waypoints). pf_env_reset with a NULL mask ignores what it finds and prepares fresh ones: a context handed such a state
    steps exactly like a context that started from zeros."""
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    n = 64 * 5 + 9
    mk = lambda seed: BatchEngine(build_params("quadx", "waypoints", noise="philox", autoreset="next_step", seed=seed), n, device="cuda:0")  # noqa: E731
    other, a, b = mk(99), mk(4), mk(4)
    other.env_reset()
    act = torch.empty(n, 4, device="cuda:0")
    for k in range(20):
        other.env_step(other.sample_actions(act, k))
    assert bool((other.state[7, :, 3].view(torch.int32) < 0).any())
    a.state.copy_(other.state)              # a: handed the other context's state, valid bits and all
    a.state[6].zero_()                      # (the event counters of a fresh lane: the keys then agree with b's)
    a.state[7, :, 3].view(torch.int32).bitwise_and_(VALID.item())  # the key 0 of a fresh lane, bit 31 still set
    assert torch.equal(a.env_reset(), b.env_reset())
    ends = 0
    for k in range(100):
        b.sample_actions(act, k)
        ra, rb = a.env_step(act), b.env_step(act)
        assert all(torch.equal(x, y) for x, y in zip(ra, rb)), k
        ends += int((ra[2] | ra[3]).sum())
    assert ends > n  # (through the next episodes' resets as well: they consumed the fresh spares)
