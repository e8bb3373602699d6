"""scratch experiment driver: time pf_env_step variants (not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine

def timeit(tag, n=65536, steps=500, env=("quadx","hover"), **kw):
    settle = kw.pop("settle", None)
    P = build_params(env[0], env[1], **kw)
    if settle is not None: P.settle_steps = settle
    eng = BatchEngine(P, n)
    g = 50
    ring = [torch.empty(n,4,device="cuda") for _ in range(g)]
    for i,a in enumerate(ring): eng.sample_actions(a, i)
    eng.env_reset()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(g): eng.env_step(ring[i])
        s.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for i in range(g): eng.env_step(ring[i])
        gr.replay(); s.synchronize()
        e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(steps//g): gr.replay()
        e1.record(s); s.synchronize()
    us = e0.elapsed_time(e1)*1e3/steps
    done = (eng.terminated|eng.truncated).float().mean().item()
    print(f"{tag:40s} n={n:7d} {us:8.2f} us/step  {n/us:8.1f} M steps/s  done-frac {done:.4f}", flush=True)

if __name__ == "__main__":
    timeit("hover philox next_step", noise="philox", autoreset="next_step")
    timeit("hover philox next_step settle=0", noise="philox", autoreset="next_step", settle=0)
    timeit("hover noise-off next_step settle=0", noise="off", autoreset="next_step", settle=0)
    timeit("hover noise-off next_step", noise="off", autoreset="next_step")
    for n in (4096, 16384, 262144, 1048576):
        timeit("hover philox next_step", n=n, noise="philox", autoreset="next_step")
    timeit("quadx wp philox next_step", env=("quadx","waypoints"), noise="philox", autoreset="next_step")
    timeit("fixedwing wp philox next_step", env=("fixedwing","waypoints"), noise="philox", autoreset="next_step")
