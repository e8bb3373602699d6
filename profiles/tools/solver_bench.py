"""profiling tool: time per physics tick with every body resting on the floor (the contact solve runs for every lane in every tick)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pyflyt_amd.core import Aviary
N = int(os.environ.get("N", "16384"))
for kind, opts, z0 in (("quadx", None, 0.1), ("fixedwing", dict(drone_model="acrowing", starting_velocity=(0.0, 0.0, 0.0)), 0.4)):
    pos = np.zeros((N, 3)); pos[:, 2] = z0
    av = Aviary(pos, np.zeros((N, 3)), drone_type=kind, drone_options=opts, seed=0)
    av.set_mode(0 if kind == "fixedwing" else -1)
    for _ in range(400):
        av.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        av.step()
    e1.record(); torch.cuda.synchronize()
    st = av.all_states
    print(f"{kind}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per tick at {N} resting bodies; mean z {float(st[:, 3, 2].mean()):.3f} contact {float(av.contact_array.float().mean()) if hasattr(av, 'contact_array') else -1:.2f}")
