import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle as O
from test_gpu_aviary import sample_setpoint
lib = O.lib()
for mode in range(0, 8):
    n, steps, seed = 64, 120, 40 + mode
    rng = np.random.default_rng(seed)
    start_pos = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), rng.uniform(1.5, 2.5, size=(n, 1))], axis=1)
    start_orn = rng.uniform(-0.15, 0.15, size=(n, 3)) * np.array([1, 1, 5.0])
    sets = []
    for eps in (0.0, 6e-8):
        Ps, Ls = [], []
        for i in range(n):
            P = O.make_params("quadx", noise_mode=O.NOISE_PHILOX, seed=seed, start_pos=start_pos[i]*(1+eps), start_rpy=start_orn[i]); L = O.Lane()
            lib.orc_aviary_reset(C.byref(P), C.byref(L), i); lib.orc_set_mode(C.byref(P), C.byref(L), mode); L.v[2] += eps*2; L.w[0] += eps; L.throttle[0] += eps; Ps.append(P); Ls.append(L)
        sets.append((Ps, Ls))
    rng2 = np.random.default_rng(1)
    out = []
    for k in range(steps):
        if k % 20 == 5:
            sp = sample_setpoint(rng2, n, "quadx", mode).astype(np.float32)
            for Ps, Ls in sets:
                for i, L in enumerate(Ls):
                    for j in range(4): L.setpoint[j] = float(sp[i, j])
        sts = []
        for Ps, Ls in sets:
            for P, L in zip(Ps, Ls):
                lib.orc_aviary_step(C.byref(P), C.byref(L), None, 0, 0); L.rng_ctr += 1
            sts.append(np.array([list(L.w_b)+list(L.rpy)+list(L.v_b)+list(L.p)+list(L.throttle) for L in Ls]))
        e = np.abs(sts[0]-sts[1]).max(axis=1)
        if k in (9, 29, 59, 119): out.append(f"step {k+1}: median {np.median(e):.1e} max {e.max():.1e} frac>1e-4 {np.mean(e>1e-4):.2f}")
    print("mode", mode, " | ".join(out))
