"""fp32_mode7_fixture.py -- evidence script (CPU only, not a test): an fp32 twin of the fp64 oracle (double -> float, nothing else)
replays tests/golden/env_quadx_waypoints_mode7.npz -- the position-controlled waypoint chase -- and prints its worst distance from the
fixture next to the fp64 oracle's (measured: 7.9e-4 against 2.3e-12). See tests/test_gpu_golden.py: ENV_RTOL."""
import sys, os, re, subprocess, tempfile, types, ctypes as C
import numpy as np
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from oracle import oracle as O64
def build_f32(keep_double=()):
    tmp = tempfile.mkdtemp(prefix="orc_f32_")
    for name in ("uav_oracle.c", "uav_oracle.h"):
        src = open(os.path.join(ROOT, "oracle", name)).read()
        src = re.sub(r"\bdouble\b", "float", src)
        for pat, rep in keep_double:
            src = src.replace(pat, rep)
        open(os.path.join(tmp, name), "w").write(src)
    lib = os.path.join(tmp, "libuav_oracle.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-ffp-contract=off", "-w", "-fopenmp", "-shared", "-o", lib, os.path.join(tmp, "uav_oracle.c"), "-lm"])
    code = open(os.path.join(ROOT, "oracle", "oracle.py")).read().replace("c_double", "c_float").replace("float64", "float32")
    mod = types.ModuleType("oracle_f32"); mod.__file__ = os.path.join(tmp, "oracle.py")
    exec(compile(code, mod.__file__, "exec"), mod.__dict__)
    mod._LIB_PATH = lib; mod.build = lambda force=False: lib
    return mod
def run(O, name="env_quadx_waypoints_mode7", env="quadx_waypoints", over={"flight_mode": 7, "goal_reach_distance": 0.4}):
    g = np.load(f"{ROOT}/tests/golden/{name}.npz")
    f = np.float32 if O is not O64 else np.float64
    P = O.make_params(env, noise_mode=O.NOISE_INJECT, **over)
    lib = O.lib(); D = lib.orc_obs_dim(C.byref(P)); L = O.Lane()
    ptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_float if f is np.float32 else C.c_double))
    resets = set(int(k) for k in g["reset_before"]); ri = 0
    def do_reset():
        nonlocal ri
        xr = np.ascontiguousarray(g["reset_xi"][ri], dtype=f); u = np.ascontiguousarray(g["reset_u"][ri], dtype=f)
        lib.orc_env_reset(C.byref(P), C.byref(L), 0, ptr(xr), ptr(u)); ri += 1
    do_reset(); worst = 0.0; wk=-1
    for k in range(len(g["action"])):
        if k in resets: do_reset()
        a = np.ascontiguousarray(g["action"][k], dtype=f); xi = np.ascontiguousarray(np.nan_to_num(g["xi"][k]), dtype=f)
        lib.orc_env_step(C.byref(P), C.byref(L), ptr(a), ptr(xi))
        obs = np.frombuffer(L.obs, dtype=f, count=D).astype(np.float64)
        ref = g["obs"][k]
        e = np.abs(obs - ref).max() / max(1.0, np.linalg.norm(ref[:13]))
        if e > worst: worst, wk = e, k
    return worst, wk
print("fp64 oracle:", run(O64))
O32 = build_f32()
print("fp32 twin  :", run(O32))
