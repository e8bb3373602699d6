"""profiling tool: run a few hundred eager env steps for rocprofv3 (not part of the product)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
env = sys.argv[1] if len(sys.argv) > 1 else "hover"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
veh, task = {"hover": ("quadx","hover"), "qwp": ("quadx","waypoints"), "fwp": ("fixedwing","waypoints")}[env]
eng = BatchEngine(build_params(veh, task, noise="philox", autoreset="next_step"), n)
ring = [torch.empty(n,4,device="cuda") for _ in range(16)]
for i,a in enumerate(ring): eng.sample_actions(a, i)
eng.env_reset()
for i in range(steps): eng.env_step(ring[i%16])
torch.cuda.synchronize()
print("done")
