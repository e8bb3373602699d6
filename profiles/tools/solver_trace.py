"""profiling tool: the contact solver's call statistics from the -DPF_PHASE_TRACE variant library (PF_LIB_PATH): calls per launch,
shader-clock cycles spent in the setup (vertex generation + records) and in the sweeps, contacts and sweeps per call.
WHAT=landed (16 384 quadrotors resting on the floor, Aviary level) | hover (the headline env) | ma (shared-world MA hover)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflyt_amd import build_params, _lib

what = os.environ.get("WHAT", "landed")
L = _lib.lib()


def read():
    buf = (C.c_ulonglong * 8)()
    assert L.pf_debug_solver_trace(buf) == 0
    return np.array(list(buf), dtype=np.float64)


def report(tag, launches, t):
    calls = max(t[0], 1.0)
    print(f"{tag}: {t[0] / launches:.1f} solver calls per launch; per call (shader clocks, inside the function): count pass {t[1] / calls:.0f}, records {t[6] / calls:.0f}, "
          f"scan + bookkeeping {t[7] / calls:.0f}, sweeps {t[2] / calls:.0f}; "
          f"max contacts {t[3] / calls:.2f}, lanes with contacts {t[4] / calls:.2f}, sweeps run {t[5] / calls:.2f}")


if what == "landed":
    from pyflyt_amd.core import Aviary
    N = 16384
    for kind, opts, z0 in (("quadx", None, 0.1), ("fixedwing", dict(drone_model="acrowing", starting_velocity=(0.0, 0.0, 0.0)), 0.4)):
        pos = np.zeros((N, 3)); pos[:, 2] = z0
        av = Aviary(pos, np.zeros((N, 3)), drone_type=kind, drone_options=opts, seed=0)
        av.set_mode(0 if kind == "fixedwing" else -1)
        for _ in range(400):
            av.step()
        torch.cuda.synchronize(); read()
        for _ in range(50):
            av.step()
        torch.cuda.synchronize()
        report(f"landed {kind} (per Aviary step = 2 ticks, {N // 64} waves)", 50, read())
elif what == "hover":
    from pyflyt_amd.engine import BatchEngine
    n = 65536
    eng = BatchEngine(build_params("quadx", "hover", noise="philox", autoreset="next_step", seed=0), n, device="cuda:0")
    ring = [torch.empty(n, 4, device="cuda:0") for _ in range(16)]
    for i, a in enumerate(ring):
        eng.sample_actions(a, i)
    eng.env_reset()
    for i in range(200):
        eng.env_step(ring[i % 16])
    torch.cuda.synchronize(); read()
    for i in range(100):
        eng.env_step(ring[i % 16])
    torch.cuda.synchronize()
    report("hover 65536 (per env step = 6 ticks, 1024 waves)", 100, read())
