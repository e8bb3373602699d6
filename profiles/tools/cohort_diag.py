"""profiling tool: the age structure of the lanes' episodes (QuadX-Hover, uniform random actions) after a rollout vs after per-step
launches with the bench's 16-entry action ring -- does the bench's setup leave the population in its steady state?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflyt_amd import build_params, _lib
from pyflyt_amd.engine import BatchEngine
n = 65536
def hist(eng, tag):
    sc = eng.ints()[:, 0].cpu().numpy()
    q = np.percentile(sc, [10, 25, 50, 75, 90])
    print(f"{tag}: episode age (env steps) mean {sc.mean():.1f}, percentiles 10/25/50/75/90 = {q}, younger than 30: {(sc < 30).mean():.3f}")
eng = BatchEngine(build_params("quadx", "hover", noise="philox", autoreset="next_step", seed=0), n, device="cuda:0")
ring = [torch.empty(n, 4, device="cuda:0") for _ in range(16)]
for i, a in enumerate(ring): eng.sample_actions(a, i)
eng.env_reset()
o, r, te, tr, _ = eng.rollout(400, step_index0=1 << 20)
print("rollout: terminated per step (every 25th):", te.sum(1).cpu().numpy()[::25], "truncated:", tr.sum(1).cpu().numpy()[::25])
hist(eng, "after the 400-step rollout")
col = []
for i in range(400):
    o, r, te, tr = eng.env_step(ring[i % 16])
    col.append((int(te.sum()), int(tr.sum())))
print("per-step launches (ring actions): terminated per step (every 25th):", [c[0] for c in col][::25], "truncated", [c[1] for c in col][::25])
hist(eng, "after 400 per-step launches")
fresh = torch.empty(n, 4, device="cuda:0")
col = []
for i in range(400):
    eng.sample_actions(fresh, 5000 + i)
    o, r, te, tr = eng.env_step(fresh)
    col.append((int(te.sum()), int(tr.sum())))
print("per-step launches (fresh actions): terminated per step (every 25th):", [c[0] for c in col][::25], "truncated", [c[1] for c in col][::25])
hist(eng, "after 400 per-step launches with fresh actions")
