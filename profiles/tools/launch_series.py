"""profiling tool: per-launch durations of the env-step kernel over a long run (HIP events around every launch), summarised in
blocks -- where in an episode cohort's life the slow launches (contact-solver tails, mass resets) fall."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine

n = int(os.environ.get("N", "65536")); steps = int(os.environ.get("STEPS", "1600")); blk = int(os.environ.get("BLOCK", "50"))
P = build_params("quadx", os.environ.get("TASK", "hover"), noise="philox", autoreset="next_step")
eng = BatchEngine(P, n)
ring = [torch.empty(n, 4, device="cuda") for _ in range(16)]
for i, a in enumerate(ring): eng.sample_actions(a, i)
eng.env_reset()
launch = [eng.prepare_step(a) for a in ring]
s = torch.cuda.current_stream()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
for _ in range(50): launch[0](s.cuda_stream)
torch.cuda.synchronize()
eng.env_reset()
ev[0].record()
for i in range(steps):
    launch[i % 16](s.cuda_stream); ev[i + 1].record()
torch.cuda.synchronize()
d = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(steps)])
print(f"{steps} launches: mean {d.mean():.2f} us, median {np.median(d):.2f}, p90 {np.percentile(d, 90):.2f}, max {d.max():.1f} (event-to-event: includes the launch gap)")
for b in range(0, steps, blk):
    x = d[b:b + blk]
    print(f"steps {b:5d}-{b + blk - 1:5d}: mean {x.mean():6.2f}  median {np.median(x):6.2f}  max {x.max():6.1f}  >1.4x median: {(x > 1.4 * np.median(d)).sum():3d}")
