"""scratch: does splitting the 65536-lane batch into S sub-batches on S free-running streams hide the
load/store bursts? Each sub-batch replays its own 100-step hipGraph; wall time for K steps of all."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
N = int(os.environ.get("N", "65536")); K = 2000; G = 100
for S in (1, 2, 4, 8):
    n = N // S
    engs, rings, graphs, streams = [], [], [], []
    for s in range(S):
        P = build_params("quadx", "hover", noise="philox", autoreset="next_step", seed=0)
        e = BatchEngine(P, n, device="cuda:0", lane_offset=s * n)
        ring = [torch.empty(n, 4, device="cuda") for _ in range(G)]
        for i, a in enumerate(ring): e.sample_actions(a, i)
        e.env_reset()
        engs.append(e); rings.append(ring)
    torch.cuda.synchronize()
    for s in range(S):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for i in range(10): engs[s].env_step(rings[s][i])
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for i in range(G): engs[s].env_step(rings[s][i])
        graphs.append(g); streams.append(st)
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for r in range(K // G):
            for s in range(S):
                with torch.cuda.stream(streams[s]):
                    graphs[s].replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"N={N} streams={S}: {dt / K * 1e6:.2f} us per whole-batch step, {N * K / dt / 1e9:.2f} G steps/s")
    del engs, graphs
