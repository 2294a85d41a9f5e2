"""Multi-GPU sharding of a problem batch: one process per GPU, no exchange while solving.

The reference's only parallel construct is `parfor` over independent RRT seeds followed by a
gather and a min (Lib/functions/s_Parallel_rrt.m:14-28).  Here the batch dimension is split
contiguously over the ranks; every rank solves its shard with its own `CFSBatch`; one
`all_gather` (RCCL over xGMI with backend "nccl", gloo on CPU) returns the converged
trajectories, inputs, statuses and iteration counts to every rank, packed into a single collective.
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
    on every rank.  Shards may be ragged (B not divisible by the world size): records are padded to
    the largest shard for the collective and trimmed afterwards.

    All tensors travel in ONE collective: per problem they are packed into a float64 record (int32
    statuses and counts are exact in fp64), so a step costs one all_gather latency on xGMI instead of one
    per field (~3.7 KB per problem: u, x_, cost, status, iter_O)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0] for r in range(world)]
    mx, n = max(sizes), sizes[rank]
    names = list(local)
    shapes, dtypes, widths = {}, {}, {}
    cols = []
    for name in names:
        t = local[name]
        assert t.shape[0] == n, (name, t.shape, n)
        if t.is_floating_point():
            assert t.dtype == torch.float64, (name, t.dtype)
        else:
            assert t.dtype in (torch.int32, torch.int16, torch.uint8, torch.int8, torch.bool), (name, t.dtype)   # exact in fp64
        shapes[name], dtypes[name] = tuple(t.shape[1:]), t.dtype
        flat = t.reshape(n, -1).to(torch.float64)
        widths[name] = flat.shape[1]
        cols.append(flat)
    rec = torch.cat(cols, dim=1) if cols else torch.zeros((n, 0), dtype=torch.float64)
    if n < mx:
        rec = torch.cat([rec, rec.new_zeros((mx - n, rec.shape[1]))])
    rec = rec.contiguous()
    allrec = rec.new_empty((world * mx, rec.shape[1]))
    dist.all_gather_into_tensor(allrec, rec, group=group)
    full = torch.cat([allrec[r * mx: r * mx + sizes[r]] for r in range(world)])
    out, c0 = {}, 0
    for name in names:
        w = widths[name]
        out[name] = full[:, c0:c0 + w].to(dtypes[name]).reshape((full.shape[0],) + shapes[name]).contiguous()
        c0 += w
    return out


def best_of(cost: torch.Tensor, status: torch.Tensor) -> int:
    """Index of the cheapest successfully solved problem (the analogue of `min(routeL)` in
    s_Parallel_rrt.m:27); -1 if none succeeded."""
    ok = status < 2
    if not bool(ok.any()):
        return -1
    c = torch.where(ok, cost, torch.full_like(cost, float("inf")))
    return int(torch.argmin(c).item())


def solve_sharded(solve_local, inputs: dict, B: int, group=None, keep=("u", "x_", "status", "iter_O", "cost")) -> dict:
    """The whole multi-GPU path for a FIXED batch of B problems (strong scaling: BASELINE configs 4 and 5 are worded
    "sharded 8 x MI355X"): every rank takes the contiguous shard `shard_bounds(B, rank, world)` of the full-batch `inputs`
    (arrays / tensors whose leading dimension is B), solves it with `solve_local(shard_inputs, lo, hi)` on its own GPU
    (no collective while solving), and ONE all_gather returns the full-batch results to every rank; `best` is the index of
    the cheapest solved problem -- s_Parallel_rrt.m:16-28's `parfor` + gather + `min`.

    `solve_local` returns a dict of per-problem tensors for the shard (at least the names in `keep`; `cost` = final cost).
    On one rank (no process group) this is just the local solve.  Returns the gathered dict plus `best` and `bounds`."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = shard_bounds(B, rank, world)
    shard = {k: (v[lo:hi] if v is not None else None) for k, v in inputs.items()}
    local = solve_local(shard, lo, hi)
    local = {k: local[k] for k in keep}
    for k, v in local.items():
        assert v.shape[0] == hi - lo, (k, tuple(v.shape), lo, hi)
    full = gather_results(local, B, group=group)
    full = dict(full)
    full["best"] = best_of(full["cost"], full["status"])
    full["bounds"] = (lo, hi)
    return full


def cfs_solve_local(slv, device, noise_key="noise"):
    """`solve_local` for `solve_sharded` over a `solvers.CFSBatch` living on this rank's GPU: host or device inputs in,
    device tensors out (cfs_solve_batch_device on the current stream)."""
    def run(shard, lo, hi):
        t = lambda a: a if isinstance(a, torch.Tensor) and a.is_cuda else torch.as_tensor(a, dtype=torch.float64).to(device).contiguous()  # noqa: E731
        nz = shard.get(noise_key)
        out = slv.solve_device(t(shard["x_init"]), t(shard["xR1"]), t(shard["ff"]), t(shard["caug"]), t(shard["obs"]),
                               noise=None if nz is None else t(nz))
        n_it = (out.iter_O - 1).clamp(min=1).long()
        cost = out.cost_all.gather(1, (n_it - 1).unsqueeze(1)).squeeze(1)        # eval.cost_new of every problem
        return dict(u=out.u, x_=out.x_, status=out.status, iter_O=out.iter_O, cost=cost, total_iter=out.total_iter)
    return run
