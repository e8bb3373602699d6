"""row_order_probe.py -- evidence script (CPU only, not a test): how much does the ORDER of the contact rows within a sweep matter?
The model sweeps contact by contact (normal, friction x, friction y of one point, then the next point); Bullet's multibody solver is
remembered to sweep all normal rows first and all friction rows after them [BULLET-FROM-MEMORY] -- listed in DESIGN.md section 3 under
"not restated". A temporary copy of the oracle with that order replays the landing drops next to the shipped one; printed: the largest
distance between the two trajectories (position, attitude) and between the resting poses.   python tests/tools/row_order_probe.py"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OLD = """    for (int c = 0; c < n; ++c) {
      for (int d = 0; d < 3; ++d) {
        double rxd[3], ang[3], axr[3], u[3];"""
NEW = """    for (int pass_ = 0; pass_ < (g_normals_first ? 2 : 1); ++pass_)
    for (int c = 0; c < n; ++c) {
      for (int d = 0; d < 3; ++d) {
        if (g_normals_first && ((pass_ == 0) != (d == 0))) continue; /* pass 0: the normal rows, pass 1: the friction rows */
        double rxd[3], ang[3], axr[3], u[3];"""


def patched_library():
    tmp = tempfile.mkdtemp(prefix="orc_rows_")
    s = open(os.path.join(ROOT, "oracle", "uav_oracle.c")).read()
    assert s.count(OLD) == 1
    s = s.replace(OLD, NEW)
    i = s.index("\n", s.rindex("#include")) + 1
    s = s[:i] + "static int g_normals_first = 0;\nvoid probe_order(int m) { g_normals_first = m; }\n" + s[i:]
    open(os.path.join(tmp, "uav_oracle.c"), "w").write(s)
    open(os.path.join(tmp, "uav_oracle.h"), "w").write(open(os.path.join(ROOT, "oracle", "uav_oracle.h")).read())
    lib = os.path.join(tmp, "libuav_oracle.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-ffp-contract=off", "-w", "-fopenmp", "-shared", "-o", lib, os.path.join(tmp, "uav_oracle.c"), "-lm"])
    return lib


def drops(lib, model, z0, tilt, steps, n=64, seed=5):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        pos = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(z0, z0 + 0.2)]
        rpy = [rng.uniform(-tilt, tilt), rng.uniform(-tilt, tilt), rng.uniform(-3, 3)]
        kw = dict(start_vel=[0.0, 0.0, 0.0]) if model == "fixedwing" else {}
        P = O.make_params(model, noise_mode=O.NOISE_OFF, start_pos=pos, start_rpy=rpy, **kw)
        L = O.Lane()
        lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
        lib.orc_set_mode(C.byref(P), C.byref(L), 0 if model == "fixedwing" else -1)
        for j in range(8):
            L.setpoint[j] = 0.0
        tr = []
        for k in range(steps):
            lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
            tr.append(list(L.p) + list(L.rpy))
        out.append(tr)
    return np.array(out)  # [n, steps, 6]


def main():
    O._LIB_PATH = patched_library()
    O.build = lambda force=False: O._LIB_PATH
    lib = O.lib()
    for model, z0, tilt, steps in (("quadx", 0.25, 0.6, 240), ("primitive_drone", 0.45, 0.6, 400), ("fixedwing", 0.6, 0.3, 400)):
        lib.probe_order(0)
        a = drops(lib, model, z0, tilt, steps)
        lib.probe_order(1)
        b = drops(lib, model, z0, tilt, steps)
        d = np.abs(a - b)
        d[..., 5] = np.minimum(d[..., 5], 2 * np.pi - d[..., 5])  # (yaw wraps)
        print(f"{model}: normals-first against point-by-point over {a.shape[0]} tilted drops of {steps} Aviary steps: trajectories apart by at most "
              f"{d[..., :3].max():.2e} m and {d[..., 3:].max():.2e} rad; final poses by {d[:, -1, :3].max():.2e} m, {d[:, -1, 3:5].max():.2e} rad (roll, pitch), "
              f"{d[:, -1, 5].max():.2e} rad (yaw)")


if __name__ == "__main__":
    main()
