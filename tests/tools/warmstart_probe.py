"""warmstart_probe.py -- evidence script (CPU only, not a test): what carrying a persisting contact's NORMAL impulse over from the
previous tick (the warm start of Bullet's rigid-body solver: factor 0.85, friction rows from zero; switched off -- `if (0)` -- in
btMultiBodyConstraintSolver::setupMultiBodyContactConstraint, the solver PyFlyt's URDF bodies go through [BULLET-FROM-MEMORY], which
is why the model starts every solve from zero) would do to the number of sweeps a body at rest needs. Patches a temporary copy of the oracle's source
(contact ids out of the vertex scan, a one-body cache in the solve) -- the shipped oracle, the device code and the fixtures do not
warm-start. Quoted in DESIGN.md section 3.   python tests/tools/warmstart_probe.py"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def patched_library():
    tmp = tempfile.mkdtemp(prefix="orc_ws_")
    s = open(os.path.join(ROOT, "oracle", "uav_oracle.c")).read()
    edits = [
        ("static int contact_points_reach(const orc_params* P, const double p[3], const double q[4], double pts[][3], double depth[], const double margin) {\n  int n = 0;",
         "static int g_ids[ORC_MAX_CONTACTS];\nstatic double g_ws = 0.0;\nstatic int g_ws_friction = 0, g_last_sweeps = 0;\n"
         "void probe_set(double f, int fr) { g_ws = f; g_ws_friction = fr; }\nint probe_sweeps(void) { return g_last_sweeps; }\n"
         "static int contact_points_reach(const orc_params* P, const double p[3], const double q[4], double pts[][3], double depth[], const double margin) {\n  int n = 0;"),
        ("        depth[n] = -wpt[2];\n        ++n;", "        depth[n] = -wpt[2];\n        g_ids[n] = k * 16 + i;\n        ++n;"),
        ("  int sweeps_run = 0;\n",
         "  int sweeps_run = 0;\n"
         "  static int pn = 0, pids[ORC_MAX_CONTACTS];\n  static double plam[ORC_MAX_CONTACTS][3];\n  int ids[ORC_MAX_CONTACTS];\n  memcpy(ids, g_ids, sizeof(ids));\n"
         "  if (!B->persisted) pn = 0;\n"
         "  for (int c = 0; c < n; ++c)\n    for (int j = 0; j < pn; ++j)\n      if (pids[j] == ids[c])\n        for (int d = 0; d < (g_ws_friction ? 3 : 1); ++d) {\n"
         "          double rxd[3], ang[3];\n          cross3(arm[c], dir[d], rxd);\n          matvec(Iw, rxd, ang);\n          const double dl = g_ws * plam[j][d];\n          lam[c][d] = dl;\n"
         "          for (int i = 0; i < 3; ++i) { vc[i] += im * dl * dir[d][i]; w[i] += dl * ang[i]; }\n        }\n"),
        ("  if (g_solve_stats_on) {\n", "  pn = n;\n  memcpy(pids, ids, sizeof(ids));\n  memcpy(plam, lam, sizeof(double) * 3 * n);\n  g_last_sweeps = sweeps_run;\n  if (g_solve_stats_on) {\n"),
    ]
    for old, new in edits:
        assert s.count(old) == 1, old
        s = s.replace(old, new)
    open(os.path.join(tmp, "uav_oracle.c"), "w").write(s)
    open(os.path.join(tmp, "uav_oracle.h"), "w").write(open(os.path.join(ROOT, "oracle", "uav_oracle.h")).read())
    lib = os.path.join(tmp, "libuav_oracle.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-ffp-contract=off", "-w", "-fopenmp", "-shared", "-o", lib, os.path.join(tmp, "uav_oracle.c"), "-lm"])
    return lib


def main():
    O._LIB_PATH = patched_library()
    O.build = lambda force=False: O._LIB_PATH
    lib = O.lib()
    lib.probe_set.argtypes = [C.c_double, C.c_int]
    for model, z0, kw in (("quadx", 0.1, {}), ("acrowing", 0.4, dict(start_vel=[0, 0, 0]))):
        for factor, friction in ((0.0, 0), (0.85, 0), (0.85, 1), (1.0, 0), (1.0, 1)):
            lib.probe_set(factor, friction)
            P = O.make_params(model, noise_mode=O.NOISE_OFF, start_pos=[0, 0, z0], start_rpy=[0.02, 0.01, 0.3], **kw)
            L = O.Lane()
            lib.orc_aviary_reset(C.byref(P), C.byref(L), 0)
            lib.orc_set_mode(C.byref(P), C.byref(L), 0 if model != "quadx" else -1)
            for j in range(8):
                L.setpoint[j] = 0.0
            for k in range(400):  # comes down and settles
                lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
            sweeps = []
            for k in range(100):
                lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
                sweeps.append(lib.probe_sweeps())
            print(f"{model} at rest, warm-start factor {factor}{' (friction rows too)' if friction else ''}: {sum(sweeps) / len(sweeps):.1f} sweeps per solve")


if __name__ == "__main__":
    main()
