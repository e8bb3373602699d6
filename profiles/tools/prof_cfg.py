"""profiling tool: eager env steps for rocprofv3 with config knobs (SETTLE, NOISE env vars)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
n = int(os.environ.get("N", "65536")); steps = int(os.environ.get("STEPS", "60"))
kw = dict(flight_mode=int(os.environ["MODE"])) if "MODE" in os.environ else {}
P = build_params(os.environ.get("VEH","quadx"), os.environ.get("TASK","hover"), noise=os.environ.get("NOISE","philox"), autoreset="next_step",
                 world_options=(dict(contact_response=os.environ["CR"] == "1") if "CR" in os.environ else None), **kw)
if "SETTLE" in os.environ: P.settle_steps = int(os.environ["SETTLE"])
eng = BatchEngine(P, n)
ring = [torch.empty(n,4,device="cuda") for _ in range(100)]  # (a ring that repeats within an episode is a different action process: solver_trace.py WHAT=rates)
for i,a in enumerate(ring): eng.sample_actions(a, i)
eng.env_reset()
for i in range(steps): eng.env_step(ring[i%100])
torch.cuda.synchronize()
