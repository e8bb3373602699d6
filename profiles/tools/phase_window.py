"""profiling tool: the launches bench.py times in the driver's shape, one by one, with the -DPF_PHASE_TRACE variant library: for
every launch the span from the first wave's entry to the last wave's exit, and the phase timeline of the wave that exits last."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflyt_amd import build_params, _lib
from pyflyt_amd.engine import BatchEngine
n = 65536
wo = dict(contact_response=False) if os.environ.get("CR", "1") == "0" else None
eng = BatchEngine(build_params("quadx", "hover", noise="philox", autoreset="next_step", seed=0, world_options=wo), n, device="cuda:0")
ring = [torch.empty(n, 4, device="cuda:0") for _ in range(16)]
for i, a in enumerate(ring): eng.sample_actions(a, i)
eng.env_reset()
eng.rollout(400, step_index0=1 << 20)
for i in range(21): eng.env_step(ring[i % 16])
eng.rollout(300, step_index0=(1 << 20) + 400)
torch.cuda.synchronize()
L = _lib.lib(); K = 13; waves = 1024
names = ["entry", "ints", "philox", "unpack", "resets", "steps", "preobs", "obsLDS", "obsst", "statest", "ack"]
for rep in range(int(os.environ.get("STEPS", "30"))):
    eng.env_step(ring[rep % 16]); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (waves * K))()
    assert L.pf_debug_phase_trace(buf, waves * K) == 0
    T = np.frombuffer(buf, dtype=np.uint64).reshape(waves, K).astype(np.int64)
    rt0, rt1 = T[:, 11], T[:, 12]
    span = (rt1.max() - rt0.min()) / 100.0
    w = int(np.argmax(rt1))
    clk = np.median((T[:, 10] - T[:, 0]) / np.maximum(rt1 - rt0, 1)) * 100.0
    rel = (T[w, :11] - T[w, 0]) / clk
    med = np.median((T[:, :11] - T[:, :1]) / clk, axis=0)
    print(f"launch {rep:2d}: span {span:6.2f} us; last wave {w:4d} entered at +{(rt0[w] - rt0.min()) / 100.0:5.2f} us, lived {(rt1[w] - rt0[w]) / 100.0:6.2f} us (median wave {np.median(rt1 - rt0) / 100.0:5.2f}); its phases (us since entry, [median wave]): "
          + " ".join(f"{nm} {rel[i]:.2f}[{med[i]:.2f}]" for i, nm in enumerate(names) if i in (3, 4, 5, 7, 10)))
