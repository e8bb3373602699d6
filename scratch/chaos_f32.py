import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle as O64
import oracle_f32 as O32
O32._HERE = os.path.dirname(os.path.abspath(__file__)); O32._LIB_PATH = os.path.join(O32._HERE, "libuav_oracle_f32.so")
from test_gpu_aviary import sample_setpoint
for mode in range(0, 8):
    n, steps, seed = 64, 120, 40 + mode
    rng = np.random.default_rng(seed)
    start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(1.5, 2.5, size=(n, 1))], axis=1).astype(np.float32).astype(np.float64)
    start_orn = rng.uniform(-0.15, 0.15, size=(n, 3)) * np.array([1, 1, 5.0])
    sets = []
    for O in (O64, O32):
        lib = O.lib(); Ps, Ls = [], []
        for i in range(n):
            P = O.make_params("quadx", noise_mode=O.NOISE_PHILOX, seed=seed, start_pos=start_pos[i], start_rpy=start_orn[i]); L = O.Lane()
            lib.orc_aviary_reset(C.byref(P), C.byref(L), i); lib.orc_set_mode(C.byref(P), C.byref(L), mode); Ps.append(P); Ls.append(L)
        sets.append((lib, Ps, Ls))
    rng2 = np.random.default_rng(1); out = []
    ok = np.ones(n, bool)
    for k in range(steps):
        if k % 20 == 5:
            sp = sample_setpoint(rng2, n, "quadx", mode).astype(np.float32)
            for lib, Ps, Ls in sets:
                for i, L in enumerate(Ls):
                    for j in range(4): L.setpoint[j] = float(sp[i, j])
        sts = []
        for lib, Ps, Ls in sets:
            for P, L in zip(Ps, Ls):
                lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0); L.rng_ctr += 1
            sts.append(np.array([list(L.w_b)+list(L.rpy)+list(L.v_b)+list(L.p)+list(L.throttle) for L in Ls], dtype=np.float64))
        e = np.abs(sts[0]-sts[1]).max(axis=1); ok &= e < 1e-4
        if k in (29, 59, 119): out.append(f"step {k+1}: median {np.median(e):.1e} max {e.max():.1e} dropped {1-ok.mean():.2f}")
    print("mode", mode, " | ".join(out))
