"""Multi-rank path on CPU (gloo, world_size 2): the sharding plan used by bench.py and the
VectorEnv façades gives, lane for lane, the results of a single-process run, because lanes are
independent and the RNG is keyed by the GLOBAL lane index; the ranks only meet for a barrier and a
max-reduction of their clocks (no data-path collective). The oracle stands in for the kernels here
(test infrastructure); the identical property is asserted on the GPU in test_gpu_api.py."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from pyflyt_amd.dist import max_over_ranks, strong_shard, weak_shard  # noqa: E402


def test_shard_plans():
    s = [weak_shard(65536, r, 8) for r in range(8)]
    assert [x.lane_offset for x in s] == [r * 65536 for r in range(8)] and s[0].global_lanes == 524288
    u = [strong_shard(1000, r, 3, unit=4) for r in range(3)]  # shared worlds of 4 lanes never straddle ranks
    assert sum(x.lanes for x in u) == 1000 and all(x.lanes % 4 == 0 and x.lane_offset % 4 == 0 for x in u)
    assert [x.lane_offset for x in u] == [0, u[0].lanes, u[0].lanes + u[1].lanes]
    with pytest.raises(ValueError):
        strong_shard(1002, 0, 3, unit=4)
    t = [strong_shard(1000, r, 3) for r in range(3)]
    assert [x.lanes for x in t] == [334, 333, 333] and [x.lane_offset for x in t] == [0, 334, 667]
    assert sum(x.lanes for x in t) == 1000
    with pytest.raises(ValueError):
        weak_shard(0, 0, 1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_global, steps, out):
    from oracle import oracle as O

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = strong_shard(n_global, rank, world)
    P = O.make_params("hover", noise_mode=O.NOISE_PHILOX, seed=5)
    ob = O.OracleBatch(P, sh.lanes, lane0=sh.lane_offset)
    ob.reset()
    rng = np.random.default_rng(0)
    dist.barrier()
    for _ in range(steps):
        a_all = rng.uniform([-3, -3, -3, 0], [3, 3, 3, 0.8], size=(n_global, 4)).astype(np.float32)
        obs, rew, term, trunc, _ = ob.step(a_all[sh.lane_offset: sh.lane_offset + sh.lanes], autoreset=1)
    dist.barrier()
    # no data-path collective: results are gathered only to be checked
    gathered = [None] * world
    dist.all_gather_object(gathered, (sh.lane_offset, obs, rew))
    slow = max_over_ranks(1.0 + rank, dist)
    if rank == 0:
        out.put((gathered, slow))
    dist.destroy_process_group()


def test_two_rank_run_equals_single_process():
    from oracle import oracle as O

    n, steps, world = 96, 40, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, slow = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert slow == 2.0  # max over ranks of (1 + rank)
    obs = np.concatenate([g[1] for g in sorted(gathered, key=lambda g: g[0])])
    rew = np.concatenate([g[2] for g in sorted(gathered, key=lambda g: g[0])])
    ob = O.OracleBatch(O.make_params("hover", noise_mode=O.NOISE_PHILOX, seed=5), n)
    ob.reset()
    rng = np.random.default_rng(0)
    for _ in range(steps):
        a_all = rng.uniform([-3, -3, -3, 0], [3, 3, 3, 0.8], size=(n, 4)).astype(np.float32)
        ref_obs, ref_rew, *_ = ob.step(a_all, autoreset=1)
    assert np.array_equal(obs, ref_obs) and np.array_equal(rew, ref_rew)
