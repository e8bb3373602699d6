import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
n = 65536
for task in ("hover", "waypoints"):
    eng = BatchEngine(build_params("quadx", task, noise="philox", autoreset="next_step", seed=0), n, device="cuda:0")
    ring = [torch.empty(n, 4, device="cuda:0") for _ in range(100)]
    for i, a in enumerate(ring): eng.sample_actions(a, i)
    eng.env_reset()
    kw = eng.state[7, :, 3].view(torch.int32)
    print(task, "after reset: valid fraction", float((kw < 0).float().mean()))
    for k in range(64):
        term_before = (eng.flags() & 3) != 0
        kwb = eng.state[7, :, 3].view(torch.int32).clone()
        eng.env_step(ring[k % 100])
        kw = eng.state[7, :, 3].view(torch.int32)
        if k % 4 == 3 or k in (14, 15, 16):
            print(k, "valid fraction %.3f" % float((kw < 0).float().mean()), "lanes resetting this step with a valid spare: %d of %d" % (int(((kwb < 0) & term_before).sum()), int(term_before.sum())))
