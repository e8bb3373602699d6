"""profiling tool: soak -- 50k graph-replayed env steps per env kind, finite outputs, counters advance, episode statistics sane."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
for veh, task, n in (("quadx", "hover", 65536), ("quadx", "waypoints", 65536), ("fixedwing", "waypoints", 65536)):
    P = build_params(veh, task, noise="philox", autoreset="next_step", seed=1)
    eng = BatchEngine(P, n)
    ring = [torch.empty(n, 4, device="cuda") for _ in range(100)]
    for i, a in enumerate(ring): eng.sample_actions(a, i)
    eng.env_reset()
    st = torch.cuda.Stream()
    done = torch.zeros((), device="cuda"); rsum = torch.zeros((), device="cuda")
    with torch.cuda.stream(st):
        for i in range(5): eng.env_step(ring[i])
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(100):
                eng.env_step(ring[i])
                done += (eng.terminated | eng.truncated).float().mean()
                rsum += eng.reward.mean()
        t0 = time.perf_counter()
        for _ in range(500): g.replay()
        st.synchronize()
        dt = time.perf_counter() - t0
    K = 50000
    assert torch.isfinite(eng.obs).all() and torch.isfinite(eng.state[:5]).all()
    ints = eng.ints()
    print(f"{veh}/{task}: {K} steps in {dt:.2f} s, done/step {float(done)/K:.4f}, mean reward {float(rsum)/K:.3f}, "
          f"min event counter {int(ints[:,2].min())}, max |p| {float(eng.state[0][:, :3].abs().max()):.1f}")
    eng.close()
