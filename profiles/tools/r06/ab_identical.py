#!/usr/bin/env python3
"""Are two builds of the library bit-identical on an env? Runs `STEPS` random-action env steps of ENV at N lanes with each library
(one subprocess per library: PF_LIB_PATH is read at import) and compares a running checksum of every step's observations, rewards
and flags and of the final state. Usage: ab_identical.py libA.so libB.so [vehicle:task ...]"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch

    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine

    veh, task = sys.argv[2].split(":")
    n, steps = int(os.environ.get("N", "65536")), int(os.environ.get("STEPS", "400"))
    eng = BatchEngine(build_params(veh, task, noise="philox", autoreset="next_step", seed=5), n, device="cuda:0")
    eng.env_reset()
    act = torch.empty(n, 4, device="cuda:0")
    h = hashlib.sha256()
    ends = 0
    for k in range(steps):
        eng.sample_actions(act, k)
        if veh == "quadx":
            act[:, 3] *= 0.5  # (low thrust: most episodes end on the floor, through the contact solve)
        o, r, t, u = eng.env_step(act)
        ends += int((t | u).sum())
        for x in (o, r, t, u):
            h.update(x.cpu().numpy().tobytes())
    h.update(eng.state[:7].cpu().numpy().tobytes())
    print(h.hexdigest(), ends)
    sys.exit(0)

libs, envs = sys.argv[1:3], (sys.argv[3:] or ["quadx:waypoints"])
ok = True
for e in envs:
    out = []
    for lib in libs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", e], capture_output=True, text=True,
                           env=dict(os.environ, PF_LIB_PATH=os.path.abspath(lib)))
        out.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else f"FAILED: {r.stderr[-400:]}")
    same = out[0] == out[1] and not out[0].startswith("FAILED")
    ok &= same
    print(f"{e}: {'IDENTICAL' if same else 'DIFFERENT'}  {out[0][:16]}.. / {out[1][:16]}..  episode ends: {out[0].split()[-1]}")
sys.exit(0 if ok else 1)
