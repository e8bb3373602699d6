#!/bin/bash
# generic-kernel resident rollout: tests + timing against one launch per step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04h
timeout 900 python -m pytest tests/test_gpu_rollout.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04h/rollout_tests.txt
cat gpurun_out/r04h/rollout_tests.txt
timeout 300 python - <<'PY' 2>&1 | tee gpurun_out/r04h/generic_rollout_timing.txt
import os, time, torch
os.environ["PF_DISABLE_FAST"] = "1"
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
for veh, task, kw in (("quadx", "hover", {}), ("quadx", "waypoints", {}), ("quadx", "hover", dict(flight_mode=7)), ("fixedwing", "waypoints", {})):
    P = build_params(veh, task, noise="philox", autoreset="next_step", seed=1, **kw)
    e = BatchEngine(P, 65536, device="cuda:0")
    e.env_reset()
    act = torch.empty(65536, 4, device="cuda:0")
    e.rollout(100, step_index0=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.rollout(200, step_index0=100)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for s in range(200):
        e.sample_actions(act, 300 + s)
        e.env_step(act)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{veh} {task} {kw}: generic kernel, 65536 lanes: resident rollout {(t1 - t0) / 200 * 1e6:.1f} us per step, sample + step launches {(t2 - t1) / 200 * 1e6:.1f} us per step")
PY
