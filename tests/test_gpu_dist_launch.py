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
    # `value` means the same thing in every mode: one launch per env step (the state-resident figure has its own key)
    assert "per-step" in d["value_kind"] or "per env step" in d["value_kind"]
    assert d["rollout"]["value"] > 0 and d["value"] != d["rollout"]["value"]
    assert ("state_resident_note" in d) == (scaling == "strong")
    assert d["world_size"] == 2 and d["launcher"] == "torch.distributed.run environment" and d["collective_backend"] == "gloo"
    assert len(d["timed"]["per_rank_ms_per_step"]) == 2


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun environment (how the driver spells it): bench.py launches the two ranks itself
    and the line says n_gpus 2 -- round 5's bench.py printed a one-GPU line under that command."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PF_BENCH_SINGLE_DEVICE="1", PF_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "10", "--batch", "4096",
           "--no-cpu-baseline", "--rollout-steps", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0], r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and "itself" in d["launcher"]
    assert d["config"]["global_batch"] == 8192 and d["value"] > 0


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus 8` on a box with fewer devices exits non-zero and prints no result line."""
    import torch

    if torch.cuda.device_count() >= 8:
        pytest.skip("an 8-GPU box: nothing to refuse")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PF_BENCH_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0 and "refusing" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
