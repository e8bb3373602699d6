"""scratch: MA hover at 65 536 agents: independent lanes (specialised kernel) vs one shared world per env copy (generic kernel)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyflyt_amd.pz_envs import MAQuadXHoverEnv
n = 65536
for shared in (False, True):
    env = MAQuadXHoverEnv(num_envs=n // 4, seed=0, shared_world=shared)
    env.reset(seed=0)
    e = env.engine
    acts = [torch.empty(n, 4, device="cuda") for _ in range(16)]
    for i, a in enumerate(acts): e.sample_actions(a, i)
    torch.cuda.synchronize()
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(200):
            e.env_step(acts[k % 16])
            if k % 25 == 24:
                done = (e.terminated | e.truncated).view(-1, 4).any(dim=1, keepdim=True).expand(-1, 4).reshape(-1)  # whole worlds
                e.env_reset(mask=done)
        e1.record(); torch.cuda.synchronize()
    print(f"ma_hover shared_world={shared}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us/step (kernel {e.lib.pf_ctx_is_specialised(e._ctx)})")
    env.close()
