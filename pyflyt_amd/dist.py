"""Multi-GPU sharding: drones are independent, so the batch is cut into
contiguous per-rank slices and every rank steps its slice with its own context -- there is NO
collective on the hot path. torch.distributed (RCCL on ROCm, gloo on CPU) is used only to line
the ranks up around a timed region and to take the max of their clocks."""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    lanes: int        # lanes owned by this rank
    lane_offset: int  # global index of this rank's lane 0 (keys the counter-based RNG)
    global_lanes: int


def env_rank_world() -> tuple[int, int, int]:
    """(rank, local_rank, world) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def weak_shard(lanes_per_rank: int, rank: int, world: int) -> Shard:
    """Weak scaling: every rank owns `lanes_per_rank` lanes (BASELINE config 5: 65 536 per GPU)."""
    if not (0 <= rank < world) or lanes_per_rank <= 0:
        raise ValueError("bad shard arguments")
    # (a shared world never straddles ranks as long as lanes_per_rank is a multiple of agents_per_world: pf_ctx_create checks it)
    return Shard(rank, world, lanes_per_rank, rank * lanes_per_rank, world * lanes_per_rank)


def strong_shard(global_lanes: int, rank: int, world: int, unit: int = 1) -> Shard:
    """Strong scaling: a fixed global batch cut into contiguous, near-equal slices. `unit`: lanes that must stay together
    (the agents of a shared world: pf_params.agents_per_world) -- slices are whole multiples of it."""
    if not (0 <= rank < world) or unit < 1 or global_lanes % unit != 0 or global_lanes // unit < world:
        raise ValueError("bad shard arguments")
    base, rem = divmod(global_lanes // unit, world)
    lanes = (base + (1 if rank < rem else 0)) * unit
    off = (rank * base + min(rank, rem)) * unit
    return Shard(rank, world, lanes, off, global_lanes)


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """The slowest rank's clock (the job's time); a no-op without a process group."""
    if dist is None or not dist.is_initialized():
        return float(seconds)
    import torch

    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])
