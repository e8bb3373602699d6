"""profiling tool: host time per enqueued env step (no synchronisation in the loop) through BatchEngine.env_step, a prepared step,
and (where the library has it) pf_env_step_ring -- against the kernel's 12 us, is the launch loop the bottleneck?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
n = 65536
eng = BatchEngine(build_params("quadx", "hover", noise="philox", autoreset="next_step", seed=0), n, device="cuda:0")
ring = [torch.empty(n, 4, device="cuda:0") for _ in range(100)]
for i, a in enumerate(ring): eng.sample_actions(a, i)
eng.env_reset()
s = torch.cuda.current_stream().cuda_stream
launch = [eng.prepare_step(a) for a in ring]
for name, fn in (("env_step", lambda i: eng.env_step(ring[i % 100])), ("prepared step", lambda i: launch[i % 100](s))):
    for k in (20, 200):
        for i in range(30): fn(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k): fn(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{name}: {k} launches enqueued in {(t1 - t0) / k * 1e6:.2f} us each (host), done after {(t2 - t0) / k * 1e6:.2f} us each")
if hasattr(eng, "step_ring"):
    for k in (20, 200):
        eng.step_ring(ring, 0, 30); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.step_ring(ring, 0, k); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"step_ring: {k} launches enqueued in {(t1 - t0) / k * 1e6:.2f} us each (host), done after {(t2 - t0) / k * 1e6:.2f} us each")
