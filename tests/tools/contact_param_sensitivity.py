"""contact_param_sensitivity.py -- evidence script (CPU only, not a test): how much each [BULLET-FROM-MEMORY] contact parameter
(pyflyt_amd/params.py: WORLD) moves a landing when it is set to the alternative a PyBullet capture might turn up.
64 tilted cf2x drops with the motors off per setting, fp64 oracle; against the defaults: the largest distance between the trajectories,
between the resting poses, and by how many Aviary steps the first contact REPORT moves.   python tests/tools/contact_param_sensitivity.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

ALTERNATIVES = [
    ("contact_iters 50 -> 10 (Bullet's own default)", dict(world_contact_iters=10)),
    ("contact_residual_threshold 1e-7 -> 0 (never ends early)", dict(world_contact_residual_threshold=0.0)),
    ("contact_slop 1e-5 -> 1e-3", dict(world_contact_slop=1e-3)),
    ("contact_slop 1e-5 -> 0", dict(world_contact_slop=0.0)),
    ("contact_manifold_points 4 -> 8 (every vertex)", dict(world_contact_manifold_points=8)),
    ("contact_margin 0 -> 0.02 (speculative rows 2 cm ahead)", dict(world_contact_margin=0.02)),
    ("contact_report_distance 0 -> 0.02", dict(world_contact_report_distance=0.02)),
    ("contact_break_distance 0.02 -> 0 (no persistence)", dict(world_contact_break_distance=0.0)),
    ("contact_erp 0.2 -> 0.8", dict(world_contact_erp=0.8)),
    ("contact_friction 0.5 -> 1.0", dict(world_contact_friction=1.0)),
    ("contact_restitution 0 -> 0.2", dict(world_contact_restitution=0.2)),
]


def drops(over, n=64, steps=240, seed=5):
    lib = O.lib()
    rng = np.random.default_rng(seed)
    tr, first = [], []
    for i in range(n):
        pos = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.25, 0.45)]
        rpy = [rng.uniform(-0.6, 0.6), rng.uniform(-0.6, 0.6), rng.uniform(-3, 3)]
        P = O.make_params("quadx", noise_mode=O.NOISE_OFF, start_pos=pos, start_rpy=rpy, **over)
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
        lib.orc_set_mode(C.byref(P), C.byref(L), -1)
        for j in range(8):
            L.setpoint[j] = 0.0
        t, f = [], -1
        for k in range(steps):
            lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
            t.append(list(L.p) + list(L.rpy))
            if f < 0 and L.contact_step:
                f = k
        tr.append(t); first.append(f)
    return np.array(tr), np.array(first)


def main():
    base, f0 = drops({})
    print(f"{'setting':58s} trajectory      resting pose          first report")
    for label, over in ALTERNATIVES:
        a, f = drops(over)
        d = np.abs(a - base)
        d[..., 5] = np.minimum(d[..., 5], 2 * np.pi - d[..., 5])
        print(f"{label:58s} {d[..., :3].max():.1e} m   {d[:, -1, :3].max():.1e} m {d[:, -1, 3:5].max():.1e} rad   "
              f"{int((f - f0).min()):+d} .. {int((f - f0).max()):+d} steps")


def task_level():
    """... and at task level: QuadX-Hover under LOW-THRUST random actions (every episode ends on the floor; with the action space's
    own uniform draws the drones leave the dome long before they could touch it and no setting changes a single flag), 4 096 lanes x
    300 steps, auto-reset -- episode count, mean episode length and mean step reward per setting"""
    print("\nQuadX-Hover, low-thrust random actions (every episode ends on the floor), 4 096 lanes x 300 steps:")
    rng0 = np.random.default_rng(0)
    acts = [rng0.uniform([-0.5] * 3 + [0.0], [0.5] * 3 + [0.25], size=(4096, 4)).astype(np.float32) for _ in range(300)]
    for label, over in [("defaults", {})] + ALTERNATIVES:
        ob = O.OracleBatch(O.make_params("hover", noise_mode=O.NOISE_PHILOX, seed=0, **over), 4096)
        ob.reset()
        ends, rew, collided = 0, 0.0, 0
        for a in acts:
            _, r, t, u, _ = ob.step(a, autoreset=1)
            ends += int((t | u).sum()); rew += float(r.sum())
        print(f"  {label:58s} {ends:6d} episode ends, {4096 * 300 / max(1, ends):6.2f} steps per episode, mean reward {rew / (4096 * 300):+.4f}")


if __name__ == "__main__":
    main()
    task_level()
