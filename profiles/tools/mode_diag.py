"""profiling tool: eager env-step time of a QuadX-Hover flight mode against the population's state (episode ends, contacts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyflyt_amd import build_params, _lib as PL
from pyflyt_amd.engine import BatchEngine
n = int(os.environ.get("N", "65536")); mode = int(os.environ.get("MODE", "-1"))
P = build_params("quadx", os.environ.get("TASK", "hover"), noise="philox", autoreset="next_step", seed=0, flight_mode=mode)
eng = BatchEngine(P, n, device="cuda:0")
print("mode", mode, "specialised", eng.lib.pf_ctx_is_specialised(eng._ctx))
ring = [torch.empty(n, 4, device="cuda:0") for _ in range(16)]
for i, a in enumerate(ring):
    eng.sample_actions(a, i)
eng.env_reset(); torch.cuda.synchronize()
for blk in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ends = 0
    e0.record()
    for i in range(25):
        o, r, t, tr = eng.env_step(ring[i % 16])
    e1.record(); torch.cuda.synchronize()
    f = eng.flags()
    print(f"steps {25*(blk+1):4d}: {e0.elapsed_time(e1)/25*1e3:7.1f} us/step  done(last) {int((t|tr).sum()):6d}  contact {int((f & PL.F_CONTACT).ne(0).sum()):6d}  nonfinite {int((f & PL.F_NONFINITE).ne(0).sum())}"
          f"  z<0.1 {int((eng.state[0,:,2] < 0.1).sum())}  max|w| {float(eng.state[3,:,:2].abs().max()):.0f}  max|v| {float(eng.state[2,:,:3].abs().max()):.0f}")
