"""profiling tool (run under rocprofv3 --kernel-trace): steady-state per-step launches with the caches thrashed between them by a
large memset (THRASH=1) or not -- what a training loop that runs policy inference between env steps would see."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pyflyt_amd import build_params
from pyflyt_amd.engine import BatchEngine
n = 65536
eng = BatchEngine(build_params("quadx", "hover", noise="philox", autoreset="next_step", seed=0,
                               world_options=(dict(contact_response=False) if os.environ.get("CR") == "0" else None)), n, device="cuda:0")
ring = [torch.empty(n, 4, device="cuda:0") for _ in range(16)]
for i, a in enumerate(ring): eng.sample_actions(a, i)
eng.env_reset()
launch = [eng.prepare_step(a) for a in ring]
s = torch.cuda.current_stream()
big = torch.empty(int(os.environ.get("MB", "1024")) << 18, device="cuda:0")
for i in range(300): launch[i % 16](s.cuda_stream)
torch.cuda.synchronize()
for i in range(200):
    if os.environ.get("THRASH") == "1": big.zero_()
    launch[i % 16](s.cuda_stream)
torch.cuda.synchronize()
