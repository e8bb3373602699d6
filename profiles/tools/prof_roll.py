"""profiling tool: pf_rollout launches for rocprofv3 (N, K, REPS, TASK, NOISE env vars)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
n = int(os.environ.get("N", "65536")); k = int(os.environ.get("K", "100")); reps = int(os.environ.get("REPS", "6"))
P = build_params("quadx", os.environ.get("TASK", "hover"), noise=os.environ.get("NOISE", "philox"), autoreset="next_step",
                 world_options=(dict(contact_response=os.environ["CR"] == "1") if "CR" in os.environ else None))
eng = BatchEngine(P, n)
eng.env_reset()
for i in range(reps):
    eng.rollout(k, step_index0=i * k)
torch.cuda.synchronize()
