"""fp32_contact_sensitivity.py -- evidence script (CPU only, not a test).

A float32 twin of the fp64 oracle (fp32_sensitivity.build_f32) run through the landing scenario of
tests/test_gpu_aviary.py::test_landing_parity: how far does PLAIN fp32 arithmetic drift from fp64 through the impact
transient of the contact solver (10 projected Gauss-Seidel sweeps over up to 48 vertex contacts, clamps at the friction
cone and at zero normal impulse)? The numbers set the tolerance of the GPU landing tests: they are a property of the
model in fp32, not of the kernels.

usage: python tests/tools/fp32_contact_sensitivity.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fp32_sensitivity import O64, build_f32  # noqa: E402


def main():
    O32 = build_f32()
    for drone, model, z0, tilt, steps, extra in [("quadx", "quadx", 0.25, 0.6, 240, {}), ("quadx", "primitive_drone", 0.45, 0.6, 300, {}),
                                                 ("rocket", "rocket", 2.5, 0.02, 400, dict(starting_fuel_ratio=0.0))]:
        n, seed = 128, 77
        rng = np.random.default_rng(seed)
        start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(z0, z0 + 0.2, size=(n, 1))], axis=1).astype(np.float32).astype(np.float64)
        start_orn = np.concatenate([rng.uniform(-tilt, tilt, size=(n, 2)), rng.uniform(-3, 3, size=(n, 1))], axis=1)
        mode = 0 if drone != "quadx" else -1
        sets = []
        for O in (O64, O32):
            lib = O.lib()
            Ps, Ls = [], []
            for i in range(n):
                P = O.make_params(model, noise_mode=O.NOISE_OFF, start_pos=start_pos[i], start_rpy=start_orn[i], **extra)
                L = O.Lane()
                lib.orc_aviary_reset(C.byref(P), C.byref(L), i)
                lib.orc_set_mode(C.byref(P), C.byref(L), mode)
                for j in range(8):
                    L.setpoint[j] = 0.0
                Ps.append(P); Ls.append(L)
            sets.append((lib, Ps, Ls))
        first = np.full(n, -1)
        worst_by_offset = {}
        for k in range(steps):
            sts = []
            for lib, Ps, Ls in sets:
                for P, L in zip(Ps, Ls):
                    lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0)
                sts.append(np.array([[list(L.w_b), list(L.rpy), list(L.v_b), list(L.p)] for L in Ls], dtype=np.float64))
            contact = np.array([bool(L.contact_step) for L in sets[0][2]])
            first[(first < 0) & contact] = k
            scale = np.maximum(1.0, np.linalg.norm(sts[0], axis=2, keepdims=True))
            e = (np.abs(sts[0] - sts[1]) / scale).reshape(n, -1).max(1)
            for i in range(n):
                off = "before" if first[i] < 0 else min(k - first[i], 40)
                worst_by_offset.setdefault(off, []).append(e[i])
        final = np.abs(sts[0] - sts[1])
        line = ", ".join(f"{o}: max {np.max(worst_by_offset[o]):.1e} / >1e-4: {np.mean(np.array(worst_by_offset[o]) > 1e-4):.2f}"
                         for o in ["before", 0, 1, 2, 3, 5, 10, 20, 40] if o in worst_by_offset)
        print(f"{model}: relative error by Aviary steps after the first contact -- {line}")
        print(f"   final pose |dz| max {final[:, 3, 2].max():.1e}, |d roll,pitch| max {final[:, 1, :2].max():.1e}")


if __name__ == "__main__":
    main()
