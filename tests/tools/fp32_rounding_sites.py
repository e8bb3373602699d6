"""fp32_rounding_sites.py -- evidence script (CPU only, not a test): WHICH single-precision roundings carry the cascaded modes' drift?
The fp64 oracle, patched in a temporary copy so that selected quantities are rounded to float32 where they are produced, replays
tests/golden/env_quadx_waypoints_mode7.npz; the worst distance from the fixture is printed per set of rounding sites:
  state   the rigid-body state after every tick (p, q, v, w: what an fp32 kernel carries between ticks)
  derived update_state's outputs (body-frame rates and velocities, Euler angles: what the controllers read)
  pid     every PID's error, integral, stored error and output
  motors  the motor states and thrusts
(An fp32 build of the whole oracle -- tests/tools/fp32_mode7_fixture.py -- is 7.9e-4 off.) Quoted in DESIGN.md section 3.
  python tests/tools/fp32_rounding_sites.py"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

EDITS = [
    # (site bit, anchor, replacement)
    (1, "  for (int i = 0; i < 4; ++i) q[i] = nq[i] * inv;\n}",
        "  for (int i = 0; i < 4; ++i) q[i] = nq[i] * inv;\n  if (g_sites & 1) { for (int i = 0; i < 3; ++i) { p[i] = RF(p[i]); v[i] = RF(v[i]); w[i] = RF(w[i]); } for (int i = 0; i < 4; ++i) q[i] = RF(q[i]); }\n}"),
    (2, "  orc_euler_from_quat(L->q, L->rpy);\n",
        "  orc_euler_from_quat(L->q, L->rpy);\n  if (g_sites & 2) for (int i = 0; i < 3; ++i) { L->v_b[i] = RF(L->v_b[i]); L->w_b[i] = RF(L->w_b[i]); L->rpy[i] = RF(L->rpy[i]); }\n"),
    (4, "    double error = setpoint[i] - state[i];\n", "    double error = setpoint[i] - state[i];\n    if (g_sites & 4) error = RF(error);\n"),
    (4, "    E[i] = error;\n", "    E[i] = error;\n    if (g_sites & 4) { I[i] = RF(I[i]); derivative = RF(derivative); }\n"),
    (4, "  for (int i = 0; i < n; ++i) out[i] = tmp[i];\n", "  for (int i = 0; i < n; ++i) out[i] = (g_sites & 4) ? RF(tmp[i]) : tmp[i];\n"),
    (8, "    double rpm = throttle[i] * P->max_rpm[i];\n", "    if (g_sites & 8) throttle[i] = RF(throttle[i]);\n    double rpm = throttle[i] * P->max_rpm[i];\n"),
]


def patched_library():
    tmp = tempfile.mkdtemp(prefix="orc_rnd_")
    s = open(os.path.join(ROOT, "oracle", "uav_oracle.c")).read()
    for _, old, new in EDITS:
        assert s.count(old) == 1, old
        s = s.replace(old, new)
    head = "static int g_sites = 0;\nvoid probe_sites(int m) { g_sites = m; }\n#define RF(x) ((double)(float)(x))\n"
    i = s.rindex("#include")
    i = s.index("\n", i) + 1  # (behind the last #include)
    s = s[:i] + head + s[i:]
    open(os.path.join(tmp, "uav_oracle.c"), "w").write(s)
    open(os.path.join(tmp, "uav_oracle.h"), "w").write(open(os.path.join(ROOT, "oracle", "uav_oracle.h")).read())
    lib = os.path.join(tmp, "libuav_oracle.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-ffp-contract=off", "-w", "-fopenmp", "-shared", "-o", lib, os.path.join(tmp, "uav_oracle.c"), "-lm"])
    return lib


def replay(lib, name="env_quadx_waypoints_mode7", env="quadx_waypoints", over=dict(flight_mode=7, goal_reach_distance=0.4)):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    P = O.make_params(env, noise_mode=O.NOISE_INJECT, **over)
    D = lib.orc_obs_dim(C.byref(P))
    L = O.Lane()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    resets, ri = set(int(k) for k in g["reset_before"]), 0

    def do_reset():
        nonlocal ri
        lib.orc_env_reset(C.byref(P), C.byref(L), 0, dp(np.ascontiguousarray(g["reset_xi"][ri])), dp(np.ascontiguousarray(g["reset_u"][ri])))
        ri += 1

    do_reset()
    worst = 0.0
    for k in range(len(g["action"])):
        if k in resets:
            do_reset()
        lib.orc_env_step(C.byref(P), C.byref(L), dp(np.ascontiguousarray(g["action"][k])), dp(np.ascontiguousarray(np.nan_to_num(g["xi"][k]))))
        obs = np.frombuffer(L.obs, dtype=np.float64, count=D)
        worst = max(worst, float(np.abs(obs - g["obs"][k]).max() / max(1.0, np.linalg.norm(g["obs"][k][:13]))))
    return worst


def main():
    O._LIB_PATH = patched_library()
    O.build = lambda force=False: O._LIB_PATH
    lib = O.lib()
    names = {1: "state", 2: "derived", 4: "pid", 8: "motors"}
    for mask in (0, 1, 2, 4, 8, 3, 5, 6, 7, 15):
        lib.probe_sites(mask)
        label = " + ".join(n for b, n in names.items() if mask & b) or "none (fp64)"
        print(f"rounded to float32: {label:32s} worst distance from the fixture {replay(lib):.2e}")


if __name__ == "__main__":
    main()
