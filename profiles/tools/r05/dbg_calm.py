import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
n, steps = 4096 + 37, 160
def make(calm):
    if calm: os.environ.pop("PF_NO_CALM_PATH", None)
    else: os.environ["PF_NO_CALM_PATH"] = "1"
    return BatchEngine(build_params("quadx", "waypoints", noise="philox", autoreset="next_step", seed=11), n, device="cuda:0")
a, b = make(True), make(False)
a.env_reset(); b.env_reset()
act = torch.empty(n, 4, device="cuda:0")
sink = torch.arange(n, device="cuda:0") % 5 == 0
for k in range(steps):
    a.sample_actions(act, k); act[sink, 3] = -1.0
    a.env_step(act); b.env_step(act)
    if not torch.equal(a.state, b.state):
        d = (a.state != b.state)
        g = d.any(dim=2).any(dim=1).nonzero().flatten().tolist()
        lanes = d.any(dim=2).any(dim=0).nonzero().flatten().tolist()
        print("step", k, "groups", g, "lanes", lanes[:10], len(lanes))
        l = lanes[0]
        for gg in g: print(gg, a.state[gg, l].tolist(), b.state[gg, l].tolist(), a.state[gg, l].view(torch.int32).tolist())
        print("ints a", a.state[6, l].view(torch.int32).tolist(), "b", b.state[6, l].view(torch.int32).tolist())
        break
else:
    print("equal throughout")
