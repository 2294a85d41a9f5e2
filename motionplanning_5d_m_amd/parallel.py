"""Multi-GPU sharding of a problem batch: one process per GPU, no exchange while solving.

The reference's only parallel construct is `parfor` over independent RRT seeds followed by a
gather and a min (Lib/functions/s_Parallel_rrt.m:14-28).  Here the batch dimension is split
contiguously over the ranks; every rank solves its shard with its own `CFSBatch`; one
`all_gather` (RCCL over xGMI with backend "nccl", gloo on CPU) returns the converged
trajectories, inputs, statuses and iteration counts to every rank.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(B: int, rank: int, world: int):
    """Contiguous split of range(B) over `world` ranks; the first B % world ranks get one extra."""
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_results(local: dict, B: int, group=None) -> dict:
    """all_gather of per-problem result tensors (leading dim = local shard) into full-batch tensors
    on every rank.  Shards may be ragged (B not divisible by the world size): tensors are padded to
    the largest shard for the collective and trimmed afterwards."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0] for r in range(world)]
    mx = max(sizes)
    out = {}
    for name, t in local.items():
        assert t.shape[0] == sizes[rank], (name, t.shape, sizes[rank])
        pad = t if t.shape[0] == mx else torch.cat([t, t.new_zeros((mx - t.shape[0],) + tuple(t.shape[1:]))])
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad.contiguous(), group=group)
        out[name] = torch.cat([p[: sizes[r]] for r, p in enumerate(parts)])
    return out


def best_of(cost: torch.Tensor, status: torch.Tensor) -> int:
    """Index of the cheapest successfully solved problem (the analogue of `min(routeL)` in
    s_Parallel_rrt.m:27); -1 if none succeeded."""
    ok = status < 2
    if not bool(ok.any()):
        return -1
    c = torch.where(ok, cost, torch.full_like(cost, float("inf")))
    return int(torch.argmin(c).item())
