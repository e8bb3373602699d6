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
    ring = [torch.empty(n, 4, device="cuda:0") for _ in range(100)]  # (not a short ring: WHAT=rates)
    for i, a in enumerate(ring):
        eng.sample_actions(a, i)
    eng.env_reset()
    for i in range(200):
        eng.env_step(ring[i % 100])
    torch.cuda.synchronize(); read()
    for i in range(100):
        eng.env_step(ring[i % 100])
    torch.cuda.synchronize()
    report("hover 65536 (per env step = 6 ticks, 1024 waves)", 100, read())
elif what == "after_rollout":
    # the per-step launches right after a rollout: solver calls per launch, and how many the busiest wave made (the launch waits
    # for that wave)
    from pyflyt_amd.engine import BatchEngine
    n = 65536
    eng = BatchEngine(build_params("quadx", "hover", noise="philox", autoreset="next_step", seed=0), n, device="cuda:0")
    ring = [torch.empty(n, 4, device="cuda:0") for _ in range(16)]
    for i, a in enumerate(ring):
        eng.sample_actions(a, i)
    eng.env_reset()
    eng.rollout(400, step_index0=1 << 20)
    torch.cuda.synchronize(); read()
    out = []
    for i in range(60):
        eng.env_step(ring[i % 16])
        torch.cuda.synchronize()
        t = read()
        f = eng.flags().cpu().numpy()
        out.append((int(t[0]), int(t[5]), int(((f & _lib.F_INFO_COLLISION) != 0).sum())))
    print("per-step launches after a 400-step rollout: (solver calls, sweeps summed, lanes reporting a collision)")
    print(" ".join(f"{a}/{b}/{c}" for a, b, c in out))
elif what == "bench_window":
    # the launches bench.py times in the driver's shape (--steps 20 --warmup 5), after its setup: how many solver calls do they hold?
    from pyflyt_amd.engine import BatchEngine
    n = 65536
    eng = BatchEngine(build_params("quadx", "hover", noise="philox", autoreset="next_step", seed=0), n, device="cuda:0")
    ring = [torch.empty(n, 4, device="cuda:0") for _ in range(16)]
    for i, a in enumerate(ring):
        eng.sample_actions(a, i)
    eng.env_reset()
    eng.rollout(400, step_index0=1 << 20)
    for i in range(21):
        eng.env_step(ring[i % 16])
    eng.rollout(300, step_index0=(1 << 20) + 400)
    torch.cuda.synchronize(); read()
    out = []
    for i in range(25):
        eng.env_step(ring[i % 16])
        torch.cuda.synchronize()
        t = read()
        out.append((int(t[0]), int(t[5]), int(t[3])))
    print("warm-up 5 + timed 20 launches: (solver calls, sweeps summed, contacts summed)")
    print(" ".join(f"{a}/{b}/{c}" for a, b, c in out))
elif what == "rates":
    # contact-solver calls per launch under different action processes (uniform draws over the action box): fresh every step,
    # or a ring of R draws per lane repeated -- the bench's graphs replay a ring
    from pyflyt_amd.engine import BatchEngine
    n = 65536
    veh, task = os.environ.get("VEH", "quadx"), os.environ.get("TASK", "hover")
    print(f"-- {veh} {task}, {n} lanes")
    for R in [int(x) for x in os.environ.get("RINGS", "0,16,20,100,400").split(",")]:
        eng = BatchEngine(build_params(veh, task, noise="philox", autoreset="next_step", seed=0), n, device="cuda:0")
        ring = [torch.empty(n, 4, device="cuda:0") for _ in range(max(R, 1))]
        for i, a in enumerate(ring):
            eng.sample_actions(a, i)
        eng.env_reset()
        rates = []
        for blk in range(6):
            torch.cuda.synchronize(); read()
            for i in range(200):
                k = blk * 200 + i
                if R == 0:
                    eng.sample_actions(ring[0], 1000 + k)
                eng.env_step(ring[k % R] if R else ring[0])
            torch.cuda.synchronize()
            t = read()
            rates.append(t[0] / 200.0)
        report(f"  last block, ring {R}", 200, t)
        print(f"action ring of {R or 'fresh draws every step'}: solver calls per launch over blocks of 200 steps: " + " ".join(f"{r:.2f}" for r in rates))
elif what == "perlaunch":
    # launch by launch: the solver's calls, sweeps and contact counts (launches with exactly one call show the distribution per call)
    from pyflyt_amd.engine import BatchEngine
    n = 65536
    task = os.environ.get("TASK", "waypoints")
    eng = BatchEngine(build_params("quadx", task, noise="philox", autoreset="next_step", seed=0), n, device="cuda:0")
    ring = [torch.empty(n, 4, device="cuda:0") for _ in range(100)]
    for i, a in enumerate(ring):
        eng.sample_actions(a, i)
    eng.env_reset()
    for i in range(200):
        eng.env_step(ring[i % 100])
    torch.cuda.synchronize(); read()
    single = []
    for i in range(int(os.environ.get("LAUNCHES", "600"))):
        eng.env_step(ring[i % 100])
        torch.cuda.synchronize()
        t = read()
        if t[0] == 1:
            single.append((int(t[3]), int(t[5]), int(t[2]), int(t[6])))
    a = np.array(single)
    print(f"-- quadx {task}: {len(a)} launches with exactly one solver call")
    for nc in range(1, 5):
        m = a[:, 0] == nc
        if m.any():
            sw = a[m, 1]
            print(f"  {nc} contact(s): {int(m.sum())} calls; sweeps: median {np.median(sw):.0f}, p90 {np.percentile(sw, 90):.0f}, max {sw.max()}, ran into the cap: {int((sw >= 50).sum())}; "
                  f"clocks in the sweeps: median {np.median(a[m, 2]):.0f}, per sweep {np.median(a[m, 2] / np.maximum(sw, 1)):.0f}; records {np.median(a[m, 3]):.0f}")
elif what == "calm":
    # how many waves keep the contact response's call site in their tick loop (some lane within reach of the floor this env step)
    from pyflyt_amd.engine import BatchEngine
    n = 65536
    eng = BatchEngine(build_params("quadx", os.environ.get("TASK", "hover"), noise="philox", autoreset="next_step", seed=0), n, device="cuda:0")
    ring = [torch.empty(n, 4, device="cuda:0") for _ in range(100)]
    for i, a in enumerate(ring):
        eng.sample_actions(a, i)
    eng.env_reset()
    print(f"-- quadx {os.environ.get('TASK', 'hover')}")
    buf = (C.c_ulonglong * 2)()
    for blk in range(4):
        for i in range(100):
            eng.env_step(ring[i])
        torch.cuda.synchronize()
        assert L.pf_debug_calm_trace(buf) == 0
        print(f"steps {blk * 100}-{blk * 100 + 99}: {buf[0] / 100:.2f} of {(n + 63) // 64} waves per launch are not calm over the env step; "
              f"{buf[1] / 100:.2f} Aviary steps per launch run with the contact response's call site")
