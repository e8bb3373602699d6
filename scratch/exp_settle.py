"""scratch: how much of the hover step is the in-kernel reset? (settle_steps 10 vs 0, and episodes that never end)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
n = 65536
def run(settle, gentle):
    P = build_params("quadx", "hover", noise="philox", autoreset="next_step")
    P.settle_steps = settle
    if gentle: P.max_steps = 10**9; P.dome = 1e9
    eng = BatchEngine(P, n)
    ring = [torch.empty(n, 4, device="cuda") for _ in range(100)]
    for i, a in enumerate(ring):
        eng.sample_actions(a, i)
        if gentle: a[:, :3] *= 0.02; a[:, 3] = 0.36
    eng.env_reset()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for i in range(10): eng.env_step(ring[i])
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for i in range(100): eng.env_step(ring[i])
        for _ in range(3): g.replay()
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        st.synchronize()
        dt = time.perf_counter() - t0
    frac = float((eng.terminated | eng.truncated).float().mean())
    print(f"settle_steps={settle} gentle={gentle}: {dt / 2000 * 1e6:.2f} us/step, done fraction per step {frac:.3f}")
run(10, False); run(0, False); run(10, True)
