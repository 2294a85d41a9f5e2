#!/usr/bin/env python3
"""bench.py -- CFS iterations/s on BASELINE config 3 (5-DoF M200i, 30 waypoints, 8 obstacles, batch 1024).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode CFS|PSGCFS] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one complete batched solve (all outer iterations: 20 for PSGCFS_FANUC, the solver the batch-1024
config of BASELINE.json names; convergence-terminated <= 20 for CFS_FANUC, reported under "other_mode") of the
1024 problems resident on this rank's GPU; inputs are in HBM before the timed region starts.  One
unit = one CFS outer iteration of one problem (get_con + QP + rollout + cost/stop test; BASELINE.md
section 3).  With N GPUs every rank solves its own 1024 problems (weak scaling, seed + rank) and the
converged trajectories are all-gathered (RCCL) inside the timed region.  Rank 0 prints ONE JSON line.

`roofline`: the dominant kernel is the fused solve kernel (cfs_solve_fused_kernel).  Its duration is
measured live with HIP events recorded by the library on the launch stream over the timed steps;
`achieved` = algorithmic bytes (585 120 B per problem-iteration at this config)
x units per launch / that duration; `traffic` = HBM bytes per launch from the rocprofv3 PMC passes
committed under profiles/ (null when no such file is present).
`cpu_baseline`: the CPU oracle (oracle/cfs_oracle.c, a restatement -- kind "port") on the host cores of
this box, on the same 1024-problem batch (one pass), rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import motionplanning_5d_m_amd as pkg  # noqa: E402
from motionplanning_5d_m_amd import parallel, workloads  # noqa: E402

BATCH = 1024
ALGO_BYTES_PER_UNIT = 8 * (300 + 150 + 36000 + 240 + 36000 + 150 + 300)   # SURVEY.md 8(d), config 3 = 585 120
HBM_PEAK_GBS = 8000.0                                                       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="PSGCFS", choices=["CFS", "PSGCFS"],
                    help="headline solver: PSGCFS_FANUC is the solver BASELINE.json's batch-1024 config names; the other "
                         "one is measured too (shorter) and reported under 'other_mode'")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=16,
                    help="independent solves in flight (one handle + HIP stream each); 1 = strictly serial steps")
    ap.add_argument("--no-other-mode", action="store_true", help="measure only --mode (profiling runs)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (not used by the driver): CFS_BENCH_BACKEND=gloo runs the collective on CPU copies,
    # CFS_BENCH_DEVICE=0 lets several ranks share one GPU on a one-GPU box
    backend = os.environ.get("CFS_BENCH_BACKEND", "nccl")
    if "CFS_BENCH_DEVICE" in os.environ:
        local = int(os.environ["CFS_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = args.batch
    head = measure(args, args.mode, args.steps, args.warmup, world, rank, local, dev, backend, B, headline=True)
    other = "CFS" if args.mode == "PSGCFS" else "PSGCFS"
    oth = None
    if not args.no_other_mode:
        oth = measure(args, other, max(4, args.steps // 2), max(1, args.warmup // 2), world, rank, local, dev, backend, B, headline=False)
    if rank == 0 and oth is None:
        print(json.dumps(head), flush=True)
    elif rank == 0:
        head["other_mode"] = {k: oth[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup")}
        head["other_mode"]["config"] = {k: oth["config"][k] for k in ("solver", "iterations_per_step_rank0", "status_counts_rank0",
                                                                        "concurrent_solves", "ms_single_solve_alone")}
        head["other_mode"]["roofline"] = {k: oth["roofline"][k] for k in ("achieved", "frac", "kernel_ms_per_launch", "achieved_all_streams")}
        print(json.dumps(head), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


_WL = {}


def _workload(B, rank):
    """config 3 for this rank, generated once (both solvers are measured on the same batch)"""
    if (B, rank) not in _WL:
        _WL[(B, rank)] = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=B, seed=20260101 + rank)
    return _WL[(B, rank)]


def measure(args, mode, steps, warmup, world, rank, local, dev, backend, B, headline):
    # ---- synthetic inputs, generated with the GPU distance entry point, then resident in HBM
    pkg.lib().cfs_set_device(local)
    s, bt = _workload(B, rank)
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    # Steps are independent solves of the same resident batch.  A single solve ends with a long tail (one
    # workgroup per problem; the hardest problem of the batch runs ~4x longer than the average CU load), so
    # `--streams S` keeps S solves in flight, each with its own handle (workspace) and HIP stream: the next
    # solve's workgroups fill the CUs the previous one has already drained.
    S = max(1, args.streams if steps >= 8 else min(args.streams, 4))   # a handful of steps cannot fill 16 queues: 4 measured best at K = 5
    slvs = [pkg.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=B, device=local) for _ in range(S)]
    slv = slvs[0]
    t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()  # noqa: E731
    x_init, xR1, ff, caug, obs = t(bt.x_init), t(bt.xR1), t(bt.ff), t(bt.caug), t(bt.obs)
    noise = t(bt.noise) if mode == "PSGCFS" else None
    outs = [sl.alloc_outputs(B, dev) for sl in slvs]
    out = outs[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]

    def step(i):
        k = i % S
        with torch.cuda.stream(streams[k]):
            slvs[k].solve_device(x_init, xR1, ff, caug, obs, noise=noise, out=outs[k], stream=streams[k].cuda_stream)
            if world > 1:   # the path's one exchange: all-gather of the converged trajectories (s_Parallel_rrt.m:16-28)
                o = outs[k]
                loc = dict(u=o.u, x_=o.x_, status=o.status, iter_O=o.iter_O, cost=o.cost_all[:, -1].contiguous())
                if backend != "nccl":
                    loc = {kk: v.cpu() for kk, v in loc.items()}
                parallel.gather_results(loc, B * world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    warmup = max(warmup, S)            # every handle / stream runs at least once before the clock starts (first launches page in their scratch)
    for i in range(warmup):
        step(i)
    fence()
    # latency of one solve alone (stream 0, nothing else in flight)
    t0 = time.perf_counter()
    step(0)
    fence()
    latency_ms = (time.perf_counter() - t0) * 1e3
    for sl in slvs:
        sl.profile(True)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    fused_ms = gemm_ms = 0.0
    nsolves = 0
    for sl in slvs:
        a, b_, n = sl.profile_read()
        fused_ms += a; gemm_ms += b_; nsolves += n
        sl.profile(False)

    units_step = int((out.iter_O - 1).sum().item())          # outer iterations executed in one solve of this rank
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    usum = torch.tensor([float(units_step)], dtype=torch.float64, device=dev)
    if world > 1:
        if backend != "nccl":
            tmax, usum = tmax.cpu(), usum.cpu()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(usum, op=dist.ReduceOp.SUM)
    dt_max, units_all = float(tmax.item()), float(usum.item())

    if rank == 0:
        status = np.bincount(out.status.cpu().numpy(), minlength=4)
        kern_ms = fused_ms / max(nsolves, 1)
        achieved = ALGO_BYTES_PER_UNIT * units_step / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(f"hbm_bytes_per_launch_{mode}")
            except Exception:
                traffic = None
        res = {
            "metric": "CFS iterations/sec, 5-DoF 30-wp 8-obs batch-1024; l_inf wp err vs quadprog",
            "value": units_all * steps / dt_max,
            "unit": "CFS iterations/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt_max / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config3: M200i 5-DoF, H=30, 8 vertical line obstacles, batch %d per GPU, %s (%s), "
                                   "rng default_rng(20260101+rank), cost matrices main_FANUC.m:64-127" %
                                   (B, mode, "convergence-terminated <= 20 outer iterations" if mode == "CFS" else "20 outer iterations"),
                       "batch_per_gpu": B, "horizon": 30, "njoint": 5, "nobs": 8, "solver": mode,
                       "concurrent_solves": S, "ms_single_solve_alone": latency_ms,
                       "iterations_per_step_rank0": units_step,
                       "status_counts_rank0": {"converged": int(status[0]), "max_iter": int(status[1]),
                                               "qp_infeasible": int(status[2]), "numeric": int(status[3])}},
            "roofline": {"bound": "hbm", "kernel": "cfs_solve_fused_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms_per_launch": kern_ms, "gemm_ms_per_launch": gemm_ms / max(nsolves, 1),
                         "algorithmic_bytes_per_unit": ALGO_BYTES_PER_UNIT, "units_per_launch": units_step,
                         "achieved_all_streams": ALGO_BYTES_PER_UNIT * units_step * steps / dt_max / 1e9,
                         "note": "achieved = per-launch figure (launches of different steps overlap on %d streams; "
                                 "achieved_all_streams = bytes of all launches / wall time); true limiter is fp64 VALU + LDS "
                                 "latency of the sequential active-set steps, not HBM (DESIGN.md)" % S},
        }
        if world == 1 and headline and not args.no_cpu_baseline:
            res["cpu_baseline"], res["accuracy"] = cpu_baseline(s, bt, mode, margin, out)
        for sl in slvs:
            sl.close()
        return res
    for sl in slvs:
        sl.close()
    return None


def cpu_baseline(s, bt, mode, margin, out):
    """The oracle (checker, never the product path) timed on this box's host cores on the same batch."""
    from oracle import oracle as O
    cores = min(O.max_threads(), os.cpu_count() or 1)
    t0 = time.perf_counter()
    w = O.optimizer_batch(O.robotproperty2("M200i"), mode, s.H, 5, bt.x_init, bt.xR1, s.QQ, bt.ff, bt.caug, s.Aaug, s.Baug,
                          s.lim, s.MAX_input, bt.obs, margin, s.epsilon_O, s.MAX_O_ITER, s.alpha,
                          noise=bt.noise if mode == "PSGCFS" else None, nthreads=cores)
    dt = time.perf_counter() - t0
    units = int((w.iter_O - 1).sum())
    st, it, x = out.status.cpu().numpy(), out.iter_O.cpu().numpy(), out.x_.cpu().numpy()
    same = (st == w.status) & (it == w.iter_O)
    ok = same & (st < 2)
    err = np.abs(x - w.x_).max(axis=1)[ok]
    base = {"value": units / dt, "unit": "CFS iterations/s", "cores": cores, "kind": "port",
            "sample": "one pass over the same %d-problem batch (%d outer iterations, %.1f s wall), OpenMP over problems" % (bt.B, units, dt)}
    acc = {"vs": "CPU oracle (quadprog itself is closed source: parity unpinned, DESIGN.md)",
           "status_and_iteration_agreement": float(same.mean()), "problems_compared": int(ok.sum()),
           "linf_rad_median": float(np.median(err)) if err.size else None,
           "linf_rad_p99": float(np.quantile(err, 0.99)) if err.size else None,
           "linf_rad_max": float(err.max()) if err.size else None,
           "frac_below_1e-5_rad": float((err < 1e-5).mean()) if err.size else None}
    return base, acc


if __name__ == "__main__":
    main()
