"""The reference's examples/core/03_control.py, batched: N quadrotors in position control (flight mode 7),
each in its own world, flown to per-drone setpoints through the core `Aviary` surface."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyflyt_amd.core import Aviary

n = 1024
start_pos = np.tile(np.array([[0.0, 0.0, 1.0]]), (n, 1))
start_orn = np.zeros((n, 3))
env = Aviary(start_pos=start_pos, start_orn=start_orn, drone_type="quadx", seed=0)
env.set_mode(7)                                  # x, y, yaw, z
rng = np.random.default_rng(0)
targets = np.concatenate([rng.uniform(-1, 1, (n, 2)), rng.uniform(-1, 1, (n, 1)), rng.uniform(1, 2, (n, 1))], axis=1)
env.set_all_setpoints(targets)
for _ in range(1000):                            # 1000 / 120 Hz ~ 8 s
    env.step()
pos = env.all_states[:, 3].cpu().numpy()         # [N, 3] lin_pos rows
err = np.linalg.norm(pos - targets[:, [0, 1, 3]], axis=1)
print(f"{n} drones after {env.elapsed_time:.1f} s: median distance to target {np.median(err):.3f} m, max {err.max():.3f} m")
env.disconnect()
