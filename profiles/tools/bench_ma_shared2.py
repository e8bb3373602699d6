"""profiling tool: where the shared-world MA hover step goes: contact response on / off, generic kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
from pyflyt_amd.params import quat_from_euler
n, A = 65536, 4
start_pos = np.array([[-1.0, -1.0, 1.0], [1.0, -1.0, 1.0], [-1.0, 1.0, 1.0], [1.0, 1.0, 1.0]])
for cr in (True, False):
    P = build_params("quadx", "ma_hover", noise="philox", autoreset="off", seed=0, agents_per_world=A, start_pos=start_pos[0], world_options=dict(contact_response=cr))
    e = BatchEngine(P, n)
    pose = np.concatenate([start_pos, np.tile(quat_from_euler((0, 0, 0)), (A, 1))], axis=1)
    side = np.zeros((n, 12), dtype=np.float32); side[:, :7] = np.tile(pose, (n // A, 1))
    e.state[12:15] = torch.tensor(side, device="cuda").view(n, 3, 4).permute(1, 0, 2)
    e.env_reset()
    acts = [torch.empty(n, 4, device="cuda") for _ in range(16)]
    for i, a in enumerate(acts): e.sample_actions(a, i)
    torch.cuda.synchronize()
    for blk in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(25):
            e.env_step(acts[k % 16])
        e1.record(); torch.cuda.synchronize()
        f = e.flags()
        print(f"contact_response={cr} steps {25*(blk+1):3d}: {e0.elapsed_time(e1)/25*1e3:7.1f} us/step, in contact {int((f & 4).ne(0).sum())}, z<0.1 {int((e.state[0,:,2] < 0.1).sum())}")
    e.close()
