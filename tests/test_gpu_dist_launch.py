"""The N > 1 launcher path of bench.py -- torchrun environment, per-rank shard, barrier + max-over-ranks timing, one JSON line
from rank 0 -- run as TWO processes on the one GPU of the test box (PF_BENCH_SINGLE_DEVICE=1, gloo for the host-side
collectives: RCCL refuses two ranks on one device), so that the first real multi-GPU run is not also the first run of that path."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("scaling,global_batch", [("weak", 8192), ("strong", 4096)])
def test_two_rank_bench_line(scaling, global_batch):
    env = dict(os.environ, PF_BENCH_SINGLE_DEVICE="1", PF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "60", "--warmup", "10",
           "--batch", "4096", "--scaling", scaling, "--no-cpu-baseline", "--rollout-steps", "20"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["steps"] == 60 and d["warmup"] == 10
    assert d["config"]["global_batch"] == global_batch and d["config"]["batch_per_gpu"] == (4096 if scaling == "weak" else 2048)
    assert d["value"] > 0 and d["nonfinite_lanes"] == 0 and d["roofline"]["frac"] > 0
    assert abs(d["value"] - global_batch * 60 / (d["ms_per_step"] * 1e-3 * 60)) / d["value"] < 1e-6  # whole-job aggregate
    if scaling == "strong":  # one batch cut over the GPUs: the state-resident figure is the headline, the per-step launch next to it
        assert "pf_rollout" in d["headline"] and d["per_step_launch"]["value"] > 0 and d["value"] == d["rollout"]["value"]
    else:
        assert "per_step_launch" not in d
