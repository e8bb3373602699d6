"""solve_stats.py -- evidence script (CPU only, not a test): how many sweeps the contact solve runs in the fp64 oracle, by contact
count -- random-action rollouts of the QuadX env tasks (per 65 536-lane launch), and bodies at rest on the floor. Quoted in
DESIGN.md section 3 (what contact_iters = 50 costs and where the residual exit ends the solve)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import oracle as O  # noqa: E402

lib = O.lib()
buf = (C.c_longlong * 15)()


def stats():
    lib.orc_debug_solve_stats(buf, 1)
    return np.array(list(buf)).reshape(5, 3)


stats()
for env in ("quadx_waypoints", "hover"):
    n, steps = 4096, 300
    ob = O.OracleBatch(O.make_params(env, noise_mode=O.NOISE_PHILOX, seed=0), n)
    ob.reset()
    rng = np.random.default_rng(0)
    low, high = np.array([-np.pi] * 3 + [0.0]), np.array([np.pi] * 3 + [0.8])
    for k in range(150):
        ob.step(rng.uniform(low, high, size=(n, 4)).astype(np.float32), autoreset=1)
    stats()
    for k in range(steps):
        ob.step(rng.uniform(low, high, size=(n, 4)).astype(np.float32), autoreset=1)
    a = stats()
    print(f"{env}: solves per 65 536-lane launch by contact count 1, 2, 3, 4+: {(a[1:, 0] / steps * 65536 / n).round(3).tolist()}; mean sweeps "
          f"{(a[1:, 1] / np.maximum(a[1:, 0], 1)).round(1).tolist()}; ran into contact_iters: {a[1:, 2].tolist()}")
for model, z0, kw in (("quadx", 0.1, {}), ("acrowing", 0.4, dict(start_vel=[0, 0, 0]))):
    P = O.make_params(model, noise_mode=O.NOISE_OFF, start_pos=[0, 0, z0], start_rpy=[0.02, 0.01, 0.3], **kw)
    L = O.Lane()
    lib.orc_aviary_reset(C.byref(P), C.byref(L), 0)
    lib.orc_set_mode(C.byref(P), C.byref(L), 0 if model != "quadx" else -1)
    for j in range(8):
        L.setpoint[j] = 0.0
    for k in range(400):
        lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
    stats()
    for k in range(100):
        lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
    a = stats()
    print(f"{model} at rest on the floor: {int(a[:, 0].sum())} solves, mean sweeps {a[:, 1].sum() / max(1, a[:, 0].sum()):.1f}, ran into contact_iters: {int(a[:, 2].sum())}")
