"""The multi-GPU path with the REAL solve: parallel.solve_sharded (contiguous shards -> cfs_solve_batch_device on each rank ->
ONE packed all_gather -> best seed) with two ranks that share the one GPU of the test box (gloo collective on CPU copies,
exactly as `CFS_BENCH_BACKEND=gloo CFS_BENCH_DEVICE=0 python bench.py --gpus 2` rehearses the bench).  Problems never
interact, so the gathered result must be bit-identical to a one-process solve of the whole batch -- even and ragged shards.
On the driver's 8-GPU node the same code runs with backend "nccl" (RCCL over xGMI), one rank per GPU."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, B, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import motionplanning_5d_m_amd as pkg
        from motionplanning_5d_m_amd import parallel, workloads
        dev = torch.device("cuda", 0)
        s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=B)      # the same full batch on every rank
        lo, hi = parallel.shard_bounds(B, rank, world)
        slv = pkg.CFSBatch(s, bt.nobs, bt.margin_psg, mode="PSGCFS", max_batch=hi - lo, device=0)
        run = parallel.cfs_solve_local(slv, dev)

        def solve_local(shard, lo_, hi_):
            r = run(shard, lo_, hi_)
            torch.cuda.synchronize()
            return {k: v.cpu() for k, v in r.items()}                     # gloo: the collective runs on host copies

        full = parallel.solve_sharded(solve_local, dict(x_init=bt.x_init, xR1=bt.xR1, ff=bt.ff, caug=bt.caug, obs=bt.obs, noise=bt.noise), B)
        slv.close()
        q.put((rank, {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in full.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [96, 61])            # even and ragged shards
def test_two_ranks_sharing_the_gpu_match_one_process(gpu, B):
    from motionplanning_5d_m_amd import workloads
    s, bt = workloads.config3(lambda rb, th, ob: gpu.dist_arm(rb, th, ob)[0], B=B)
    one = gpu.CFSBatch(s, bt.nobs, bt.margin_psg, mode="PSGCFS", max_batch=B)
    want = one.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs, noise=bt.noise)
    one.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + B
    procs = [ctx.Process(target=_worker, args=(r, 2, B, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    n_it = np.maximum(want.iter_O - 1, 1)
    cost = want.cost_all[np.arange(B), n_it - 1]
    for r in (0, 1):
        f = res[r]
        np.testing.assert_array_equal(f["x_"], want.x_)
        np.testing.assert_array_equal(f["u"], want.u)
        np.testing.assert_array_equal(f["status"], want.status)
        np.testing.assert_array_equal(f["iter_O"], want.iter_O)
        np.testing.assert_array_equal(f["cost"], cost)
    ok = want.status < 2
    best = int(np.argmin(np.where(ok, cost, np.inf))) if ok.any() else -1
    assert res[0]["best"] == res[1]["best"] == best
    assert res[0]["bounds"] == (0, (B + 1) // 2) and res[1]["bounds"] == ((B + 1) // 2, B)
