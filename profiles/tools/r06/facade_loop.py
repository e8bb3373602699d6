#!/usr/bin/env python3
"""The drop-in surface in a closed loop, for `rocprofv3 --kernel-trace --stats`: make_vec("PyFlyt/QuadX-Hover-v4", 65536), a linear
policy on the device (torch.addmm + torch.clamp into a fixed action tensor) and env.step(), 2 000 steps eager -- the kernel trace shows
what step() launches: ONE kernel (pf::quadx_m0_env_kernel), next to the policy's two. Prints the wall-clock rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch  # noqa: E402

from pyflyt_amd.gym_envs import make_vec  # noqa: E402

n, steps = 65536, 2000
env = make_vec("PyFlyt/QuadX-Hover-v4", n, seed=0)
obs, infos = env.reset(seed=0)
lo = torch.tensor(env.single_action_space.low, device=env.device)
hi = torch.tensor(env.single_action_space.high, device=env.device)
W = torch.zeros(obs.shape[1], 4, device=env.device)
W[0, 0] = W[1, 1] = W[2, 2] = -0.2
W[3, 0] = W[4, 1] = -4.0
W[12, 3], W[9, 3] = -0.3, -0.2
b = torch.tensor([0.0, 0.0, 0.0, 0.3772 + 0.3], device=env.device)
act = torch.zeros(n, 4, device=env.device)
for _ in range(200):
    torch.addmm(b, obs, W, out=act)
    torch.clamp(act, min=lo, max=hi, out=act)
    obs, rew, term, trunc, infos = env.step(act)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    torch.addmm(b, obs, W, out=act)
    torch.clamp(act, min=lo, max=hi, out=act)
    obs, rew, term, trunc, infos = env.step(act)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"closed loop, eager: {dt / steps * 1e6:.2f} us per step ({n * steps / dt:.3e} env-steps/s), infos never read; "
      f"collisions in the last step: {int(infos['collision'].sum())} (read once, here)")
