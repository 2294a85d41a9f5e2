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

What is timed.  W warm-up steps (exactly the number asked for), then `--blocks` (default 5) blocks of EXACTLY K steps,
each block bracketed by barrier + torch.cuda.synchronize() on both sides; `ms_per_step` / `value` are those of the MEDIAN
block, the spread is in `blocks`.  Steps are independent solves of the same resident batch, `--streams` of them in
flight (one handle + HIP stream each): `value` is that throughput.  `value_single_launch` is the config-faithful figure
of ONE batch-1024 launch with nothing else on the GPU (median of 5): units / `config.ms_single_solve_alone`.

`roofline` (dominant kernel: cfs_solve_fused_kernel).  `achieved` = algorithmic bytes (585 120 B per
problem-iteration, the dense-constraint-matrix convention) x units per launch / the kernel's average duration measured
live with HIP events on the launch stream over the timed blocks (launches of different steps overlap, so this is a
shared-occupancy duration); `frac_single_launch` uses the duration of one launch alone.  The kernel never forms the dense
constraint matrix, so HBM is NOT what bounds it: `traffic` is the measured HBM bytes per launch (rocprofv3 --pmc
FETCH_SIZE / WRITE_SIZE passes committed under profiles/, serial launches -- `traffic_source` says which file; it is not
re-measured in this run), `hbm_actual_frac` = traffic / single-launch duration / 8 TB/s, and `valu_f64_frac` = fp64 vector
flops per launch (same PMC source) / single-launch duration / 78.6 TFLOP/s.
`cpu_baseline`: the CPU oracle (oracle/cfs_oracle.c, a restatement -- kind "port") on the host cores of this box, rank 0,
N=1 only: OpenMP over a 256-problem sample of the same batch (3 warm-ups + median of 10 passes) and, under
`single_thread`, batch-1 solves on one thread (3 warm-ups + median of 10 problems) -- BASELINE.md section 4.
"""
import argparse
import json
import os
import statistics
import sys
import time

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); launches that share a queue run one
# after the other.  Independent solves are kept in flight on their own streams below, so give each its own queue (read by
# the runtime when it initialises, i.e. before torch touches the GPU).  Measured, CFS 8 in flight: 4 queues 2.5-3.1 ms per
# solve depending on which streams collide, 16 queues 2.05 ms.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

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
FP64_VALU_PEAK_TFLOPS = 78.6                                                # MI355X fp64 vector peak (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of exactly --steps steps each (median reported)")
    ap.add_argument("--mode", default="PSGCFS", choices=["CFS", "PSGCFS"],
                    help="headline solver: PSGCFS_FANUC is the solver BASELINE.json's batch-1024 config names; the other "
                         "one is measured too (shorter) and reported under 'other_mode'")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=0,
                    help="independent solves in flight (one handle + HIP stream each); 1 = strictly serial steps; "
                         "0 = per solver: 2 for PSGCFS, 8 for CFS")
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
        head["other_mode"] = {k: oth[k] for k in ("value", "value_single_launch", "unit", "ms_per_step", "steps", "warmup", "blocks")}
        head["other_mode"]["config"] = {k: oth["config"][k] for k in ("solver", "iterations_per_step_rank0", "status_counts_rank0",
                                                                        "concurrent_solves", "ms_single_solve_alone")}
        head["other_mode"]["roofline"] = {k: oth["roofline"][k] for k in ("achieved", "frac", "frac_single_launch", "kernel_ms_per_launch",
                                                                            "kernel_ms_single_launch", "achieved_all_streams", "traffic",
                                                                            "hbm_actual_frac", "valu_f64_frac")}
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


def _pmc(mode):
    """(HBM bytes per launch, fp64 vector flops per launch, file) from the committed rocprofv3 PMC passes, or Nones"""
    f = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(f):
        return None, None, None
    try:
        d = json.load(open(f))
        return d.get(f"hbm_bytes_per_launch_{mode}"), d.get(f"fp64_valu_flops_per_launch_{mode}"), "profiles/pmc_latest.json (round %s)" % d.get("round")
    except Exception:
        return None, None, None


def measure(args, mode, steps, warmup, world, rank, local, dev, backend, B, headline):
    # ---- synthetic inputs, generated with the GPU distance entry point, then resident in HBM
    pkg.lib().cfs_set_device(local)
    s, bt = _workload(B, rank)
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    # Steps are independent solves of the same resident batch.  A single solve ends with a long tail (one
    # workgroup per problem; the hardest problem of the batch is a serial chain), so `--streams S` keeps S solves in
    # flight, each with its own handle (workspace) and HIP stream: the next solve's workgroups fill the CUs the
    # previous one has already drained.  Handles are created from the cost weights (cfs_problem_create_from_weights).
    # How many: a launch ends with its longest chain, and the next launch fills the slots that fall idle behind it.  Measured
    # on config 3 (ms per solve | ms per launch): PSGCFS 2 in flight 1.47 | 2.9, 3: 1.43 | 4.2, 16: 1.50 | 5.1 -- two saturate
    # the chip and keep every launch short (6 in flight: 1.40 | 8.4); CFS (16 hardware queues) 4: 2.6, 5: 2.2, 8: 2.05, 12: 2.02,
    # 16: 2.4 -- its 9 ms chain needs more launches behind it.
    want = args.streams if args.streams > 0 else (2 if mode == "PSGCFS" else 8)
    S = max(1, min(want, steps))
    slvs = [pkg.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=B, device=local) for _ in range(S)]
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

    for i in range(warmup):            # exactly the warm-up asked for; handles beyond it take their first launch in a timed block
        step(i)
    fence()
    # ---- timed blocks of exactly `steps` steps ------------------------------------------------------------------------
    for sl in slvs:
        sl.profile(True)
    block_s = []
    fused_ms = gemm_ms = 0.0
    nsolves = 0
    for _ in range(max(1, args.blocks)):
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        fence()
        block_s.append(time.perf_counter() - t0)
        for sl in slvs:              # outside the timed region; the events go back to the handle's pool, so only the first
            a, b_, n = sl.profile_read()   # block ever creates any
            fused_ms += a; gemm_ms += b_; nsolves += n
    # ---- one launch alone (nothing else in flight): wall latency and kernel duration, median of 5 -----------------------
    lat_ms, lone_ms = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        step(0)
        fence()
        lat_ms.append((time.perf_counter() - t0) * 1e3)
        lone_ms.append(slvs[0].profile_read()[0])
    for sl in slvs:
        sl.profile(False)
    latency_ms, kern_lone_ms = statistics.median(lat_ms), statistics.median(lone_ms)

    dt = statistics.median(block_s)
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
        algo = ALGO_BYTES_PER_UNIT * units_step
        achieved = algo / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        achieved_lone = algo / (kern_lone_ms * 1e-3) / 1e9 if kern_lone_ms > 0 else 0.0
        traffic, flops, src = _pmc(mode)
        res = {
            "metric": "CFS iterations/sec, 5-DoF 30-wp 8-obs batch-1024; l_inf wp err vs quadprog",
            "value": units_all * steps / dt_max,
            "value_single_launch": units_all / (latency_ms * 1e-3),
            "unit": "CFS iterations/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt_max / steps * 1e3,
            "blocks": {"n": len(block_s), "steps_each": steps, "ms_per_step_median": dt / steps * 1e3,
                       "ms_per_step_min": min(block_s) / steps * 1e3, "ms_per_step_max": max(block_s) / steps * 1e3},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "config3: M200i 5-DoF, H=30, 8 vertical line obstacles, batch %d per GPU, %s (%s), "
                                   "rng default_rng(20260101+rank), cost matrices main_FANUC.m:64-127" %
                                   (B, mode, "convergence-terminated <= 20 outer iterations" if mode == "CFS" else "20 outer iterations"),
                       "batch_per_gpu": B, "horizon": 30, "njoint": 5, "nobs": 8, "solver": mode,
                       "concurrent_solves": S, "ms_single_solve_alone": latency_ms,
                       "iterations_per_step_rank0": units_step,
                       "status_counts_rank0": {"converged": int(status[0]), "max_iter": int(status[1]),
                                               "qp_infeasible": int(status[2]), "numeric": int(status[3])},
                       "value_is": "throughput with %d independent solves of the batch in flight; value_single_launch is one "
                                   "batch-1024 launch alone on the GPU" % S},
            "roofline": {"bound": "hbm", "kernel": "cfs_solve_fused_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "frac_single_launch": achieved_lone / HBM_PEAK_GBS,
                         "hbm_actual_frac": (traffic / (kern_lone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and kern_lone_ms > 0 else None,
                         "valu_f64_frac": (flops / (kern_lone_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS) if flops and kern_lone_ms > 0 else None,
                         "traffic_source": (src + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, serial launches; "
                                            "not re-measured in this run") if src else None,
                         "kernel_ms_per_launch": kern_ms, "kernel_ms_single_launch": kern_lone_ms,
                         "gemm_ms_per_launch": gemm_ms / max(nsolves, 1),
                         "algorithmic_bytes_per_unit": ALGO_BYTES_PER_UNIT, "units_per_launch": units_step,
                         "achieved_all_streams": algo * steps / dt_max / 1e9,
                         "note": "HBM convention of the contract (dense constraint matrix, never materialised here): `frac` is the "
                                 "per-launch figure with %d launches sharing the chip, `frac_single_launch` one launch alone, "
                                 "`achieved_all_streams` all launches / wall time; the kernel's actual limiter is the latency of "
                                 "the sequential active-set steps (fp64 VALU + LDS + barriers), see hbm_actual_frac / "
                                 "valu_f64_frac and DESIGN.md" % S},
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
    """The oracle (checker, never the product path) timed on this box's host cores: a bounded sample of the same batch."""
    from oracle import oracle as O
    robot = O.robotproperty2("M200i")
    cores = min(O.max_threads(), os.cpu_count() or 1)
    nz = bt.noise if mode == "PSGCFS" else None

    def run(sl, threads):
        t0 = time.perf_counter()
        w = O.optimizer_batch(robot, mode, s.H, 5, bt.x_init[sl], bt.xR1[sl], s.QQ, bt.ff[sl], bt.caug[sl], s.Aaug, s.Baug, s.lim,
                              s.MAX_input, bt.obs[sl], margin, s.epsilon_O, s.MAX_O_ITER, s.alpha,
                              noise=None if nz is None else nz[sl], nthreads=threads)
        return w, time.perf_counter() - t0

    # (b) OpenMP over the batch on all host cores: 256-problem sample, 3 warm-ups + median of 10
    n_smp = min(256, bt.B)
    smp = slice(0, n_smp)
    for _ in range(3):
        w_s, _ = run(smp, cores)
    ts = [run(smp, cores)[1] for _ in range(10)]
    units_s = int((w_s.iter_O - 1).sum())
    # (a) one thread, batch 1: the like-for-like stand-in for the reference's interpreter path
    for b in range(3):
        run(slice(b, b + 1), 1)
    one = [run(slice(b, b + 1), 1) for b in range(3, 13)]
    rate1 = statistics.median([(int(w.iter_O[0]) - 1) / max(tt, 1e-9) for w, tt in one if int(w.iter_O[0]) > 1] or [0.0])
    base = {"value": units_s / statistics.median(ts), "unit": "CFS iterations/s", "cores": cores, "kind": "port",
            "sample": "first %d problems of the same batch (%d outer iterations per pass), OpenMP over problems, 3 warm-up passes + "
                      "median of 10 (min %.2f s, max %.2f s)" % (n_smp, units_s, min(ts), max(ts)),
            "single_thread": {"value": rate1, "unit": "CFS iterations/s", "cores": 1,
                              "sample": "batch-1 solves of problems 3..12 on one thread after 3 warm-up solves, median of 10"},
            "reference_logged": "~2 CFS iterations/s (M200i/test.xlsx row 19; unknown hardware, includes setup: non-comparable)"}
    # accuracy of the GPU answers on the whole batch, with the problems the oracle itself cannot pin set aside
    w, _ = run(slice(0, bt.B), cores)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import chaotic_problems
    chaotic, _ = chaotic_problems(O, s, bt, mode, w)
    st, it, x = out.status.cpu().numpy(), out.iter_O.cpu().numpy(), out.x_.cpu().numpy()
    same = (st == w.status) & (it == w.iter_O)
    ok = same & (st < 2)
    err_all = np.abs(x - w.x_).max(axis=1)
    err, err_nc = err_all[ok], err_all[ok & ~chaotic]
    acc = {"vs": "CPU oracle (quadprog itself is closed source: parity unpinned, DESIGN.md)",
           "status_and_iteration_agreement": float(same.mean()),
           "status_and_iteration_agreement_non_chaotic": float(same[~chaotic].mean()),
           "problems_compared": int(ok.sum()),
           "linf_rad_median": float(np.median(err)) if err.size else None,
           "linf_rad_p99": float(np.quantile(err, 0.99)) if err.size else None,
           "linf_rad_max": float(err.max()) if err.size else None,
           "frac_below_1e-5_rad": float((err < 1e-5).mean()) if err.size else None,
           "chaotic_definition": "the ORACLE's own answer moves by > 1e-6 rad (or changes status / iteration count) when its x_init is "
                                 "perturbed by N(0, 1e-12^2), 3 draws",
           "chaotic_problems": np.nonzero(chaotic)[0].tolist(),
           "non_chaotic_compared": int((ok & ~chaotic).sum()),
           "linf_rad_max_non_chaotic": float(err_nc.max()) if err_nc.size else None,
           "frac_below_1e-5_rad_non_chaotic": float((err_nc < 1e-5).mean()) if err_nc.size else None,
           "non_chaotic_misses": np.nonzero(ok & ~chaotic & (err_all >= 1e-5))[0].tolist()}
    return base, acc


if __name__ == "__main__":
    main()
