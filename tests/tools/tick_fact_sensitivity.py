"""tick_fact_sensitivity.py -- evidence script (CPU only, not a test): what the [BULLET-FROM-MEMORY] facts of the FREE-BODY tick are worth.
Every task runs through this tick in every lane and every physics step, so unlike the contact facts these would show everywhere. The
fp64 oracle replays flight with each fact switched to its alternative:
  use_gyro_term 1 -> 0        btMultiBody::m_useGyroTerm: the links' own omega x (I omega) (orc_world.use_gyro_term)
  max_coord_vel 100 -> 1e9    btMultiBody's per-coordinate velocity clamp (orc_world.max_coord_vel)
  no pi/4 cap                 the exponential map's ANGULAR_MOTION_THRESHOLD (patched copy of the oracle)
  plain quaternion Euler step q += dt/2 w q, normalised, instead of the exponential map (patched copy)
  no m omega x v              the composite body's point-mass term of Bullet's articulated-body algorithm (patched copy: only bodies
                              with an offset centre of mass have it -- the aeroplane)
Printed per vehicle: Aviary-level distance after 1 s and 4 s of flight under random setpoints (state vector), and task-level episode
statistics of the env with uniformly random actions.   python tests/tools/tick_fact_sensitivity.py"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

PATCHES = [
    ("  if (fAngle * dt > 0.25 * PI) fAngle = 0.25 * PI / dt; /* ANGULAR_MOTION_THRESHOLD */",
     "  if (!(g_fact & 1) && fAngle * dt > 0.25 * PI) fAngle = 0.25 * PI / dt; /* ANGULAR_MOTION_THRESHOLD */"),
    ("  double cw = cos(fAngle * dt * 0.5);\n",
     "  double cw = cos(fAngle * dt * 0.5);\n  if (g_fact & 2) { ax[0] = 0.5 * dt * w[0]; ax[1] = 0.5 * dt * w[1]; ax[2] = 0.5 * dt * w[2]; cw = 1.0; }\n"),
    ("  for (int i = 0; i < 3; ++i) a[i] = a[i] - t1[i] - t3[i];\n",
     "  for (int i = 0; i < 3; ++i) a[i] = a[i] - t1[i] - ((g_fact & 4) ? 0.0 : t3[i]);\n"),
]


def patched_library():
    tmp = tempfile.mkdtemp(prefix="orc_tick_")
    s = open(os.path.join(ROOT, "oracle", "uav_oracle.c")).read()
    for old, new in PATCHES:
        assert s.count(old) == 1, old
        s = s.replace(old, new)
    i = s.index("\n", s.rindex("#include")) + 1
    s = s[:i] + "static int g_fact = 0;\nvoid probe_fact(int m) { g_fact = m; }\n" + s[i:]
    open(os.path.join(tmp, "uav_oracle.c"), "w").write(s)
    open(os.path.join(tmp, "uav_oracle.h"), "w").write(open(os.path.join(ROOT, "oracle", "uav_oracle.h")).read())
    lib = os.path.join(tmp, "libuav_oracle.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-ffp-contract=off", "-w", "-fopenmp", "-shared", "-o", lib, os.path.join(tmp, "uav_oracle.c"), "-lm"])
    return lib


CASES = [("shipped", 0, {}), ("use_gyro_term 1 -> 0", 0, dict(world_use_gyro_term=0)), ("max_coord_vel 100 -> 1e9", 0, dict(world_max_coord_vel=1e9)),
         ("no pi/4 cap in the exponential map", 1, {}), ("plain quaternion Euler step", 2, {}), ("no m omega x v (offset centre of mass)", 4, {})]


def flights(lib, model, mode, over, n=32, steps=480, seed=3):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kw = dict(start_pos=[0.0, 0.0, 10.0 if model == "fixedwing" else 20.0])  # (high enough that no flight reaches the floor)
        P = O.make_params(model, noise_mode=O.NOISE_OFF, **kw, **over)
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
        lib.orc_set_mode(C.byref(P), C.byref(L), mode)
        tr = []
        for k in range(steps):
            if k % 30 == 0:
                sp = ([*rng.uniform(-1.0, 1.0, size=3), rng.uniform(0.3, 0.45)] if model != "fixedwing" else [*rng.uniform(-0.6, 0.6, size=3), rng.uniform(0.3, 0.9)])
                for j, x in enumerate(sp):
                    L.setpoint[j] = x
            lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
            tr.append(list(L.w_b) + list(L.rpy) + list(L.v_b) + list(L.p))
        out.append(tr)
    return np.array(out)


def main():
    O._LIB_PATH = patched_library()
    O.build = lambda force=False: O._LIB_PATH
    lib = O.lib()
    for model, mode, env in (("quadx", 0, "hover"), ("fixedwing", 0, "fixedwing_waypoints")):
        print(f"{model}: 32 flights under random setpoints, distance from the shipped tick after 1 s / 4 s (largest |difference| over the 12 state entries); "
              f"{env} env, 2 048 lanes x 200 steps of uniformly random actions")
        base = None
        for label, fact, over in CASES:
            lib.probe_fact(fact)
            tr = flights(lib, model, mode, over)
            if base is None:
                base = tr
            d = np.abs(tr - base)
            d[..., 3:6] = np.minimum(d[..., 3:6], 2 * np.pi - d[..., 3:6])
            ob = O.OracleBatch(O.make_params(env, noise_mode=O.NOISE_PHILOX, seed=0, **over), 2048)
            ob.reset()
            rng = np.random.default_rng(1)
            lo, hi = (np.array([-np.pi] * 3 + [0.0]), np.array([np.pi] * 3 + [0.8])) if model == "quadx" else (-np.ones(4), np.ones(4))
            ends, rew = 0, 0.0
            for k in range(200):
                _, r, t, u, _ = ob.step(rng.uniform(lo, hi, size=(2048, 4)).astype(np.float32), autoreset=1)
                ends += int((t | u).sum()); rew += float(r.sum())
            print(f"  {label:42s} {d[:, 119].max():.1e} / {d[:, -1].max():.1e}    {ends:6d} episode ends, mean step reward {rew / (2048 * 200):+.4f}")
    lib.probe_fact(0)


if __name__ == "__main__":
    main()
