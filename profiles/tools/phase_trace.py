"""profiling tool: per-wave phase timeline of the one-step-per-launch QuadX kernel, from the -DPF_PHASE_TRACE variant library
(PF_LIB_PATH must point at it). Prints, per phase boundary, the median / p10 / p90 over the waves of the shader-clock time since
the wave's entry, the spread of the waves' entry and exit times over the launch (100 MHz wall clock), and the clock the two imply."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflyt_amd import build_params, _lib
from pyflyt_amd.engine import BatchEngine

n = int(os.environ.get("N", "65536"))
task = os.environ.get("TASK", "hover")
wo = dict(contact_response=False) if os.environ.get("CR", "1") == "0" else None
veh = os.environ.get("VEH", "quadx")
P = build_params(veh, task, noise="philox", autoreset="next_step", seed=0, world_options=wo, flight_mode=int(os.environ.get("MODE", "0")))  # (MODE: a --modes variant library)
eng = BatchEngine(P, n, device="cuda:0")
ring = [torch.empty(n, 4, device="cuda:0") for _ in range(100)]  # (a ring that repeats within an episode is a different action process: solver_trace.py WHAT=rates)
for i, a in enumerate(ring):
    eng.sample_actions(a, i)
eng.env_reset()
for i in range(200):
    eng.env_step(ring[i % 100])
torch.cuda.synchronize()
L = _lib.lib()
K = 13
waves = min(4096, (n + 63) // 64)
names = ["entry", "int group arrived", "Philox done", "state unpacked + derive", "resets done", "Aviary steps done", "(pre obs)", "obs row in LDS",
         "obs stores issued", "state stores issued", "stores acknowledged"]
if veh == "fixedwing":
    assert L.pf_debug_solver_trace((C.c_ulonglong * 8)()) == 0  # (clears the tick counters)
acc = []
for rep in range(20):
    eng.env_step(ring[(200 + rep) % 100])
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (waves * K))()
    rc = L.pf_debug_phase_trace(buf, waves * K)
    assert rc == 0, rc
    acc.append(np.frombuffer(buf, dtype=np.uint64).reshape(waves, K).astype(np.int64).copy())
T = np.stack(acc)  # [rep][wave][stamp]
rel = T[:, :, :11] - T[:, :, :1]
rt0, rt1 = T[:, :, 11], T[:, :, 12]
clk_mhz = np.median((T[:, :, 10] - T[:, :, 0]) / np.maximum(rt1 - rt0, 1)) * 100.0
print(f"vehicle {veh} task {task} lanes {n} contact_response {bool(eng.params.contact_response)}; shader clock ~{clk_mhz:.0f} MHz (s_memtime / s_memrealtime)")
prev = 0.0
for i, nm in enumerate(names):
    v = rel[:, :, i].reshape(-1) / clk_mhz  # us
    print(f"  {i:2d} {nm:26s} median {np.median(v):7.3f} us  p10 {np.percentile(v, 10):7.3f}  p90 {np.percentile(v, 90):7.3f}   (+{np.median(v) - prev:6.3f})")
    prev = np.median(v)
start = (rt0 - rt0.min(axis=1, keepdims=True)) / 100.0
end = (rt1 - rt0.min(axis=1, keepdims=True)) / 100.0
print(f"  wave entry after the first wave's: median {np.median(start):.2f} us, p90 {np.percentile(start, 90):.2f}, max {start.max(axis=1).mean():.2f}")
print(f"  wave exit  after the first wave's entry: median {np.median(end):.2f} us, p90 {np.percentile(end, 90):.2f}, max {end.max(axis=1).mean():.2f}")
# the launch lasts as long as its slowest wave: where does THAT wave spend its time?
life = rel[:, :, 10] / clk_mhz
slow = life.argmax(axis=1)
inc = np.diff(rel[np.arange(rel.shape[0]), slow, :] / clk_mhz, axis=1)  # [rep][phase increments]
print(f"  slowest wave of each launch: life median {np.median(life.max(axis=1)):.2f} us (all waves: median {np.median(life):.2f}, p99 {np.percentile(life, 99):.2f}); its phase increments, median over launches:")
print("   " + ", ".join(f"{names[i + 1]} +{np.median(inc[:, i]):.2f}" for i in range(10)))
over = (life > np.median(life) + 1.0).sum(axis=1)
print(f"  waves more than 1 us slower than the median wave, per launch: mean {over.mean():.1f}, max {over.max()}")
if veh == "fixedwing":  # the tick's own split (diagnostic counters of the last 20 launches)
    buf = (C.c_ulonglong * 8)()
    assert L.pf_debug_solver_trace(buf) == 0
    ticks = max(int(buf[0]), 1)
    print(f"  per tick and wave: {buf[1] / ticks / clk_mhz:.3f} us in the five lifting surfaces, {buf[7] / ticks / clk_mhz:.3f} us in the motor, the floor test, "
          f"the rigid-body tick and derive(); {ticks / (20 * waves):.2f} ticks per wave and launch")
