"""fp32_sensitivity.py -- evidence script (CPU only, not a test).

Builds a float32 twin of the fp64 oracle (same source with `double` -> `float`) and measures how far
plain fp32 arithmetic drifts from fp64 per QuadX flight mode on the Aviary-level parity scenario of
tests/test_gpu_aviary.py. It shows that the drift seen on the GPU in the cascaded modes that use
the z PIDs (cf2x.yaml:43-54: z_vel kd = 0.05 at 120 Hz is a derivative gain of 6 per tick) is a
property of the reference's controller in fp32, not of the kernels. Results are quoted in DESIGN.md.

usage: python tests/tools/fp32_sensitivity.py [quadx|primitive_drone]
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O64  # noqa: E402
from test_gpu_aviary import sample_setpoint  # noqa: E402


def build_f32():
    tmp = tempfile.mkdtemp(prefix="orc_f32_")
    import re

    for name in ("uav_oracle.c", "uav_oracle.h"):
        src = open(os.path.join(ROOT, "oracle", name)).read()
        open(os.path.join(tmp, name), "w").write(re.sub(r"\bdouble\b", "float", src))
    lib = os.path.join(tmp, "libuav_oracle.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-ffp-contract=off", "-w", "-shared", "-o", lib,
                           os.path.join(tmp, "uav_oracle.c"), "-lm"])
    code = open(os.path.join(ROOT, "oracle", "oracle.py")).read().replace("c_double", "c_float").replace("float64", "float32")
    mod = types.ModuleType("oracle_f32")
    mod.__file__ = os.path.join(tmp, "oracle.py")
    exec(compile(code, mod.__file__, "exec"), mod.__dict__)
    mod._LIB_PATH = lib
    mod.build = lambda force=False: lib
    return mod


def main():
    O32 = build_f32()
    model = sys.argv[1] if len(sys.argv) > 1 else "quadx"  # "quadx" (cf2x) | "primitive_drone"
    print(f"model {model}")
    for mode in range(-1, 8):
        n, steps, seed = 64, 120, 40 + mode
        rng = np.random.default_rng(seed)
        start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(1.5, 2.5, size=(n, 1))], axis=1)
        start_pos = start_pos.astype(np.float32).astype(np.float64)
        start_orn = rng.uniform(-0.15, 0.15, size=(n, 3)) * np.array([1, 1, 5.0])
        sets = []
        for O in (O64, O32):
            lib = O.lib()
            Ps, Ls = [], []
            for i in range(n):
                P = O.make_params(model, noise_mode=O.NOISE_PHILOX, seed=seed, start_pos=start_pos[i], start_rpy=start_orn[i])
                L = O.Lane()
                lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
                lib.orc_set_mode(C.byref(P), C.byref(L), mode)
                Ps.append(P); Ls.append(L)
            sets.append((lib, Ps, Ls))
        rng2 = np.random.default_rng(1)
        ok = np.ones(n, bool)
        out = []
        for k in range(steps):
            if k % 20 == 5:
                sp = sample_setpoint(rng2, n, "quadx", mode).astype(np.float32)
                for _, _, Ls in sets:
                    for i, L in enumerate(Ls):
                        for j in range(4):
                            L.setpoint[j] = float(sp[i, j])
            sts = []
            for lib, Ps, Ls in sets:
                for P, L in zip(Ps, Ls):
                    lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
                    L.rng_ctr += 1
                sts.append(np.array([list(L.w_b) + list(L.rpy) + list(L.v_b) + list(L.p) + list(L.throttle) for L in Ls], dtype=np.float64))
            e = np.abs(sts[0] - sts[1]).max(axis=1)
            ok &= e < 1e-4
            if k in (24, 59, 119):
                out.append(f"step {k + 1}: median {np.median(e):.1e} max {e.max():.1e} >1e-4: {1 - ok.mean():.2f}")
        print(f"mode {mode:2d} fp32-oracle vs fp64-oracle | " + " | ".join(out))


if __name__ == "__main__":
    main()
