"""profiling tool: dogfight step time vs the population's state (flying / on the ground / wreck at rest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pyflyt_amd import _lib as PL
n = int(os.environ.get("N", "65536"))
if os.environ.get("FREEZE"):
    from pyflyt_amd import build_params
    from pyflyt_amd.engine import BatchEngine
    eng = BatchEngine(build_params("fixedwing", "dogfight", noise="philox", autoreset="off", seed=0, angle_representation="euler", vehicle_options=dict(drone_model="acrowing"),
                                   world_options=dict(world_scale=5.0), dogfight=dict(freeze_wrecks=True)), n, device="cuda:0")
else:
    eng = bench.make_engine("dogfight", n, torch.device("cuda:0"), 0, "philox")
ring = [torch.empty(n, 4, device="cuda:0") for _ in range(50)]
for i, a in enumerate(ring):
    eng.sample_actions(a, i); a.mul_(float(os.environ.get("AMP", "0.15"))); a[:, 3] += 0.4
eng.env_reset(); torch.cuda.synchronize()
for blk in range(14):
    if os.environ.get("RESET_AT") and blk * 50 == int(os.environ["RESET_AT"]):
        eng.env_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(50):
        eng.env_step(ring[i])
    e1.record(); torch.cuda.synchronize()
    f = eng.flags(); side = eng.state[6, :, 3].view(torch.int32); z = eng.state[0, :, 2]
    print(f"steps {50*(blk+1):4d}: {e0.elapsed_time(e1)/50*1e3:8.1f} us/step  contact {int((f & PL.F_CONTACT).ne(0).sum()):6d}  inactive {int((side & 8).ne(0).sum()):6d}"
          f"  alive {int((side & 1).ne(0).sum()):6d}  z<2 {int((z < 2).sum()):6d}  mean z {float(z.mean()):.1f}"
          f"  nonfinite {int((f & PL.F_NONFINITE).ne(0).sum())}  oob {int((side & 64).ne(0).sum())}  max|xy| {float(eng.state[0, :, :2].abs().max()):.0f}  min z {float(z.min()):.0f}"
          f"  max|w| {float(eng.state[3, :, :2].abs().max()):.0f}")
