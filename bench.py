#!/usr/bin/env python3
"""bench.py -- CFS iterations/s on BASELINE config 3 (5-DoF M200i, 30 waypoints, 8 obstacles, batch 1024); configs 4 and 5 on request.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 3|4|5] [--scaling weak|strong] [--mode CFS|PSGCFS]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` with WORLD_SIZE unset starts exactly that torch.distributed.run command as a FRESH CHILD PROCESS before this
process has imported torch or touched the GPU, and exits with its return code (never an exec of a process that holds the GPU).

One "step" = one complete batched solve (all outer iterations: 20 for PSGCFS_FANUC, the solver the batch-1024 config of
BASELINE.json names; convergence-terminated <= 20 for CFS_FANUC, reported under "other_mode") of the problems resident on
this rank's GPU; inputs are in HBM before the timed region starts.  One unit = one CFS outer iteration of one problem
(get_con + QP + rollout + cost/stop test).

Multi-GPU: problems are independent, so the batch is split contiguously over the ranks
(parallel.shard_bounds), every rank solves its shard with no collective in between, and ONE packed all-gather (RCCL) of the
converged trajectories closes every step inside the timed region (parallel.gather_results; parallel.solve_sharded is the same
path as a library call).  `--scaling weak` (default): every rank has its own full batch (1024 / 4096 / 256 problems,
seed + rank); `--scaling strong`: the config's batch is FIXED and split over the ranks, as BASELINE.json words configs 4 and 5
("sharded 8 x MI355X": 4096 -> 512 and 256 -> 32 problems per GPU; DESIGN.md section 7 says what that does to the curve).
Rank 0 prints ONE JSON line.

What is timed.  W warm-up steps (exactly the number asked for), then `--blocks` (default 5) blocks of EXACTLY K steps,
each block bracketed by barrier + torch.cuda.synchronize() on both sides; `ms_per_step` / `value` are those of the MEDIAN
block, the spread is in `blocks`.  Steps are independent solves of the same resident batch, `--streams` of them in
flight (one handle + HIP stream each): `value` is that throughput.  `value_single_launch` is the config-faithful figure
of ONE batch launch with nothing else on the GPU (median of 5): units / `config.ms_single_solve_alone`.
`solved_problems_per_s` / `iterations_per_s_solved_only` count only problems that ended OK_CONVERGED / OK_MAXITER (a third of
config 3's prescribed batch has an infeasible linearisation: those proofs are work, but not useful iterations).

`roofline` (dominant kernel: cfs_solve_fused_kernel).  `achieved` = algorithmic bytes (585 120 B per
problem-iteration at config 3, the dense-constraint-matrix convention) x units per launch / the kernel's average duration
measured live with HIP events on the launch stream over the timed blocks (launches of different steps overlap, so this is a
shared-occupancy duration); `frac_single_launch` uses the duration of one launch alone.  The kernel never forms the dense
constraint matrix, so HBM is NOT what bounds it: `traffic` is the measured HBM bytes per launch (rocprofv3 --pmc
FETCH_SIZE / WRITE_SIZE passes committed under profiles/, serial launches -- `traffic_source` says which file; it is not
re-measured in this run), `hbm_actual_frac` = traffic / single-launch duration / 8 TB/s, and `valu_f64_frac` = fp64 vector
flops per launch (same PMC source) / single-launch duration / 78.6 TFLOP/s.
`cpu_baseline`: the CPU oracle (oracle/cfs_oracle.c, a restatement -- kind "port") on the host cores of this box, rank 0,
N=1 only: OpenMP over a bounded sample of the same batch (3 warm-ups + median of 10 passes at config 3) and, under
`single_thread`, batch-1 solves on one thread -- BASELINE.md section 4.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); launches that share a queue run one
# after the other.  Independent solves are kept in flight on their own streams below, so give each its own queue (read by
# the runtime when it initialises, i.e. before torch touches the GPU).  Measured, CFS 8 in flight: 4 queues 2.5-3.1 ms per
# solve depending on which streams collide, 16 queues 2.05 ms.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool: RCCL across processes needs it (already exported on the boxes)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                                                       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6                                                # MI355X fp64 vector peak (spec)
CONFIG_BATCH = {3: 1024, 4: 4096, 5: 256}
CONFIG_MODE = {3: "PSGCFS", 4: "CFS", 5: "PSGCFS"}                           # the solver BASELINE.json's config names


def algo_bytes_per_unit(H, nj, nobs):
    """SURVEY.md section 8(d), dense-contract form: 8 [H ns + nn + nobs H nn + nobs H + nobs H nn + nn + H ns]"""
    nn, ns = H * nj, 2 * nj
    return 8 * (H * ns + nn + nobs * H * nn + nobs * H + nobs * H * nn + nn + H * ns)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of exactly --steps steps each (median reported)")
    ap.add_argument("--config", type=int, default=3, choices=[3, 4, 5], help="BASELINE.json config (3 = the headline metric)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the config's batch per GPU; strong: the config's batch split over the GPUs")
    ap.add_argument("--mode", default=None, choices=["CFS", "PSGCFS"],
                    help="headline solver (default: the one the config names -- PSGCFS_FANUC for 3 and 5, CFS_FANUC for 4); at "
                         "config 3 the other one is measured too (shorter) and reported under 'other_mode'")
    ap.add_argument("--batch", type=int, default=0, help="problems in the config's batch (default 1024 / 4096 / 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=0,
                    help="independent solves in flight (one handle + HIP stream each); 1 = strictly serial steps; "
                         "0 = per workload: 2 for PSGCFS, 8 for CFS at config 3, 2 at config 4, 4 at config 5")
    ap.add_argument("--no-other-mode", action="store_true", help="measure only --mode (profiling runs)")
    ap.add_argument("--map", default="synthetic", choices=["synthetic", "reference"],
                    help="config 5: the synthetic assembly line, or the reference's own material_traveller map (tests/golden fixture)")
    return ap.parse_args()


def self_launch(args):
    """`--gpus N` without a launcher: start torch.distributed.run as a child BEFORE anything here touches the GPU."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")               # dmabuf IPC only on this pool (RCCL across processes)
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    import motionplanning_5d_m_amd as pkg

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
    pkg.lib().cfs_set_device(local)
    mode = args.mode or CONFIG_MODE[args.config]
    ctx = dict(args=args, world=world, rank=rank, local=local, dev=dev, backend=backend)
    head = measure(ctx, mode, args.steps, args.warmup, headline=True)
    oth = None
    if args.config == 3 and not args.no_other_mode:
        other = "CFS" if mode == "PSGCFS" else "PSGCFS"
        oth = measure(ctx, other, max(4, args.steps // 2), max(1, args.warmup // 2), headline=False)
    if rank == 0 and oth is not None:
        head["other_mode"] = {k: oth[k] for k in ("value", "value_single_launch", "unit", "ms_per_step", "steps", "warmup", "blocks",
                                                  "solved_problems_per_s", "iterations_per_s_solved_only")}
        head["other_mode"]["config"] = {k: oth["config"][k] for k in ("solver", "iterations_per_step_rank0", "status_counts_rank0",
                                                                        "concurrent_solves", "ms_single_solve_alone")}
        head["other_mode"]["roofline"] = {k: oth["roofline"][k] for k in ("achieved", "frac", "frac_single_launch", "kernel_ms_per_launch",
                                                                            "kernel_ms_single_launch", "achieved_all_streams", "traffic",
                                                                            "hbm_actual_frac", "valu_f64_frac")}
    if rank == 0:
        print(json.dumps(head), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


_WL = {}


def _workload(ctx, mode):
    """(family, batch, extras) of this rank: the config's full batch (weak: seed + rank) or this rank's shard of it (strong).
    Generated once per process (both solvers are measured on the same batch)."""
    import numpy as np
    import motionplanning_5d_m_amd as pkg
    from motionplanning_5d_m_amd import parallel, workloads
    args, world, rank = ctx["args"], ctx["world"], ctx["rank"]
    cfg = args.config
    Bc = args.batch or CONFIG_BATCH[cfg]
    strong = args.scaling == "strong"
    key = (cfg, Bc, strong, args.map)
    if key in _WL:
        return _WL[key]
    seed_off = 0 if strong else rank
    extras = {}
    if cfg == 3:
        s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=Bc, seed=20260101 + seed_off)
        name = "config3: M200i 5-DoF, H=30, 8 vertical line obstacles, cost matrices main_FANUC.m:64-127, rng default_rng(20260101+rank)"
    elif cfg == 4:
        route = np.load(os.path.join(ROOT, "tests", "golden", "route_wp_200i_xori.npy"))
        s, bt = workloads.config4(route, B=Bc, seed=20260104 + seed_off)
        name = ("config4: RRT*-CFS smoothing stage, H=40, the two obstacles and cost matrices of RRTstar_CFS.m:40-50,124-187, routes = "
                "jittered copies of the logged RRT route data/200i_xori.mat (RRT trees grown on the device: tests/tools/rrt_bench.py)")
    else:
        if args.map == "reference":
            s, bt, tri = workloads.config5_reference_map(B=Bc, seed=20260105 + seed_off)
            name = "config5: M200i, H=50, the reference's own map/material_traveller.STL (12 620 triangles, MapFromSTL.m transform, mm -> m) as the one obstacle"
        else:
            s, bt, tri = workloads.config5(B=Bc, seed=20260105 + seed_off)
            name = "config5: M200i, H=50, synthetic assembly-line mesh (%d triangles) as the one obstacle" % tri.shape[0]
        extras["tri"] = tri
    lo, hi = parallel.shard_bounds(Bc, rank, world) if strong else (0, Bc)
    if strong:                                           # this rank's contiguous shard of the fixed batch
        for k in ("x_init", "xR1", "ff", "caug", "obs", "noise", "x0", "xg"):
            v = getattr(bt, k, None)
            if v is not None:
                setattr(bt, k, v[lo:hi])
        bt.B = hi - lo
    extras.update(name=name, total=Bc if strong else Bc * world, lo=lo, hi=hi)
    _WL[key] = (s, bt, extras)
    return _WL[key]


def _pmc(cfg, mode):
    """(HBM bytes per launch, fp64 vector flops per launch, file) from the committed rocprofv3 PMC passes, or Nones"""
    f = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if cfg != 3 or not os.path.exists(f):
        return None, None, None
    try:
        d = json.load(open(f))
        return d.get(f"hbm_bytes_per_launch_{mode}"), d.get(f"fp64_valu_flops_per_launch_{mode}"), "profiles/pmc_latest.json (round %s)" % d.get("round")
    except Exception:
        return None, None, None


def measure(ctx, mode, steps, warmup, headline):
    import numpy as np
    import torch
    import torch.distributed as dist
    import motionplanning_5d_m_amd as pkg
    from motionplanning_5d_m_amd import parallel
    args, world, rank, local, dev, backend = (ctx[k] for k in ("args", "world", "rank", "local", "dev", "backend"))
    cfg = args.config
    # ---- synthetic inputs, generated with the GPU distance entry point, then resident in HBM
    s, bt, ex = _workload(ctx, mode)
    B = bt.x_init.shape[0]                               # problems on this rank
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    # Steps are independent solves of the same resident batch.  A single solve ends with a long tail (one
    # workgroup per problem; the hardest problem of the batch is a serial chain), so `--streams S` keeps S solves in
    # flight, each with its own handle (workspace) and HIP stream: the next solve's workgroups fill the CUs the
    # previous one has already drained.  Handles are created from the cost weights (cfs_problem_create_from_weights).
    # How many: a launch ends with its longest chain, and the next launch fills the slots that fall idle behind it.  Measured
    # on config 3 (ms per solve | ms per launch): PSGCFS 2 in flight 1.47 | 2.9, 3: 1.43 | 4.2, 16: 1.50 | 5.1 -- two saturate
    # the chip and keep every launch short (6 in flight: 1.40 | 8.4); CFS (16 hardware queues) 4: 2.6, 5: 2.2, 8: 2.05, 12: 2.02,
    # 16: 2.4 -- its 9 ms chain needs more launches behind it.  Config 4: 2; config 5 (20 x 7 small launches per solve): 4.
    default_s = {3: (2 if mode == "PSGCFS" else 8), 4: 2, 5: 4}[cfg]
    S = max(1, min(args.streams if args.streams > 0 else default_s, steps))
    slvs = [pkg.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=B, device=local) for _ in range(S)]
    mesh = None
    if cfg == 5:
        mesh = pkg.Mesh(tri=ex["tri"])
        for sl in slvs:
            sl.set_meshes([mesh])
    t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()  # noqa: E731
    x_init, xR1, ff, caug, obs = t(bt.x_init), t(bt.xR1), t(bt.ff), t(bt.caug), t(bt.obs)
    noise = t(bt.noise) if (mode == "PSGCFS" and bt.noise is not None) else None
    outs = [sl.alloc_outputs(B, dev) for sl in slvs]
    out = outs[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    B_total = ex["total"]

    def step(i):
        k = i % S
        with torch.cuda.stream(streams[k]):
            slvs[k].solve_device(x_init, xR1, ff, caug, obs, noise=noise, out=outs[k], stream=streams[k].cuda_stream)
            if world > 1:   # the path's one exchange: all-gather of the converged trajectories (s_Parallel_rrt.m:16-28)
                o = outs[k]
                loc = dict(u=o.u, x_=o.x_, status=o.status, iter_O=o.iter_O, cost=o.cost_all[:, -1].contiguous())
                if backend != "nccl":
                    loc = {kk: v.cpu() for kk, v in loc.items()}
                parallel.gather_results(loc, B_total)

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
    it_rank = (out.iter_O - 1)
    units_step = int(it_rank.sum().item())                   # outer iterations executed in one solve of this rank
    solved = out.status < 2
    vec = torch.tensor([float(units_step), float(it_rank[solved].sum().item()), float(solved.sum().item()), float(B)],
                       dtype=torch.float64, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        if backend != "nccl":
            tmax, vec = tmax.cpu(), vec.cpu()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    dt_max = float(tmax.item())
    units_all, units_solved_all, solved_all, problems_all = (float(v) for v in vec.tolist())

    res = None
    if rank == 0:
        status = np.bincount(out.status.cpu().numpy(), minlength=4)
        kern_ms = fused_ms / max(nsolves, 1)
        per_unit = algo_bytes_per_unit(s.H, 5, bt.nobs)
        algo = per_unit * units_step
        achieved = algo / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        achieved_lone = algo / (kern_lone_ms * 1e-3) / 1e9 if kern_lone_ms > 0 else 0.0
        traffic, flops, src = _pmc(cfg, mode)
        metric = {3: "CFS iterations/sec, 5-DoF 30-wp 8-obs batch-1024; l_inf wp err vs quadprog",
                  4: "CFS iterations/sec, 5-DoF 40-wp 2-obs RRT-route batch-%d (BASELINE config 4)" % B_total,
                  5: "CFS iterations/sec, 5-DoF 50-wp mesh-map batch-%d (BASELINE config 5)" % B_total}[cfg]
        res = {
            "metric": metric,
            "value": units_all * steps / dt_max,
            "value_single_launch": units_all / (latency_ms * 1e-3),
            "solved_problems_per_s": solved_all * steps / dt_max,
            "iterations_per_s_solved_only": units_solved_all * steps / dt_max,
            "unit": "CFS iterations/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt_max / steps * 1e3,
            "blocks": {"n": len(block_s), "steps_each": steps, "ms_per_step_median": dt / steps * 1e3,
                       "ms_per_step_min": min(block_s) / steps * 1e3, "ms_per_step_max": max(block_s) / steps * 1e3},
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s; %s (%s); %s" % (ex["name"], mode, "convergence-terminated <= 20 outer iterations" if mode == "CFS" else "20 outer iterations",
                                                       ("batch %d per GPU (weak scaling)" % B) if args.scaling == "weak" else
                                                       ("batch %d split over %d GPU(s): %d problems on rank 0 (strong scaling)" % (B_total, world, B))),
                       "baseline_config": cfg, "batch_per_gpu": B, "batch_total": int(problems_all), "horizon": s.H, "njoint": 5, "nobs": bt.nobs, "solver": mode,
                       "concurrent_solves": S, "ms_single_solve_alone": latency_ms,
                       "iterations_per_step_rank0": units_step,
                       "status_counts_rank0": {"converged": int(status[0]), "max_iter": int(status[1]),
                                               "qp_infeasible": int(status[2]), "numeric": int(status[3])},
                       "value_is": "throughput with %d independent solves of the batch in flight; value_single_launch is one "
                                   "batch launch alone on the GPU; solved_problems_per_s / iterations_per_s_solved_only count "
                                   "only problems that ended converged or at MAX_ITER" % S},
            "roofline": {"bound": "hbm", "kernel": "cfs_solve_fused_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "frac_single_launch": achieved_lone / HBM_PEAK_GBS,
                         "hbm_actual_frac": (traffic / (kern_lone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and kern_lone_ms > 0 else None,
                         "valu_f64_frac": (flops / (kern_lone_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS) if flops and kern_lone_ms > 0 else None,
                         "traffic_source": (src + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, serial launches; "
                                            "not re-measured in this run") if src else None,
                         "kernel_ms_per_launch": kern_ms, "kernel_ms_single_launch": kern_lone_ms,
                         "gemm_ms_per_launch": gemm_ms / max(nsolves, 1),
                         "algorithmic_bytes_per_unit": per_unit, "units_per_launch": units_step,
                         "achieved_all_streams": algo * steps / dt_max / 1e9,
                         "note": "HBM convention of the contract (dense constraint matrix, never materialised here): `frac` is the "
                                 "per-launch figure with %d launches sharing the chip, `frac_single_launch` one launch alone, "
                                 "`achieved_all_streams` all launches / wall time; the kernel's actual limiter is the latency of "
                                 "the sequential active-set steps (fp64 VALU + LDS + barriers), see hbm_actual_frac / "
                                 "valu_f64_frac and DESIGN.md%s" % (S, "; with mesh obstacles a solve is one (mesh linearisation, fused kernel) "
                                                                   "launch pair per outer iteration and kernel_ms covers the whole sequence" if cfg == 5 else "")},
        }
        if world == 1 and headline and not args.no_cpu_baseline:
            res["cpu_baseline"], res["accuracy"] = cpu_baseline(cfg, s, bt, ex, mode, margin, out)
    for sl in slvs:
        sl.close()
    if mesh is not None:
        mesh.close()
    return res


def cpu_baseline(cfg, s, bt, ex, mode, margin, out):
    """The oracle (checker, never the product path) timed on this box's host cores: a bounded sample of the same batch."""
    import numpy as np
    from oracle import oracle as O
    robot = O.robotproperty2("M200i")
    cores = min(O.max_threads(), os.cpu_count() or 1)
    nz = bt.noise if (mode == "PSGCFS" and bt.noise is not None) else None
    obs_all = bt.obs
    if cfg == 5:                                          # a mesh obstacle enters the oracle as an obs{j}.l whose first entry is NaN
        l = O.mesh_register(0, ex["tri"])
        obs_all = np.tile(np.concatenate([l[:, 0], l[:, 1]]), (bt.x_init.shape[0], 1, 1))

    def run(sl, threads):
        t0 = time.perf_counter()
        w = O.optimizer_batch(robot, mode, s.H, 5, bt.x_init[sl], bt.xR1[sl], s.QQ, bt.ff[sl], bt.caug[sl], s.Aaug, s.Baug, s.lim,
                              s.MAX_input, obs_all[sl], margin, s.epsilon_O, s.MAX_O_ITER, s.alpha,
                              noise=None if nz is None else nz[sl], nthreads=threads)
        return w, time.perf_counter() - t0

    Bn = bt.x_init.shape[0]
    if cfg != 3:
        # configs 4 / 5: one bounded pass (the mesh oracle is brute force over every triangle: ~1.5 s per problem and thread)
        n_smp = min({4: 256, 5: 16}[cfg], Bn)
        w, tt = run(slice(0, n_smp), cores)
        st, it, x = out.status.cpu().numpy()[:n_smp], out.iter_O.cpu().numpy()[:n_smp], out.x_.cpu().numpy()[:n_smp]
        same = (st == w.status) & (it == w.iter_O)
        ok = same & (st < 2)
        err = np.abs(x - w.x_).max(axis=1)[ok]
        base = {"value": int((w.iter_O - 1).sum()) / tt, "unit": "CFS iterations/s", "cores": cores, "kind": "port",
                "sample": "first %d problems of the same batch, OpenMP over problems, one pass of %.1f s" % (n_smp, tt)}
        acc = {"vs": "CPU oracle (parity unpinned, DESIGN.md)", "problems": n_smp, "status_and_iteration_agreement": float(same.mean()),
               "linf_rad_median": float(np.median(err)) if err.size else None, "linf_rad_max": float(err.max()) if err.size else None,
               "frac_below_1e-5_rad": float((err < 1e-5).mean()) if err.size else None,
               "note": "unclassified sample: the chaos classification of this shape is in tests/test_gpu_batch.py / test_gpu_configs.py"}
        return base, acc
    # (b) OpenMP over the batch on all host cores: 256-problem sample, 3 warm-ups + median of 10
    n_smp = min(256, Bn)
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
    w, _ = run(slice(0, Bn), cores)
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
                                 "perturbed by N(0, 1e-12^2), 3 draws; every outer iteration of these problems is checked one oracle step "
                                 "at a time in tests/test_gpu_chaos.py",
           "chaotic_problems": np.nonzero(chaotic)[0].tolist(),
           "non_chaotic_compared": int((ok & ~chaotic).sum()),
           "linf_rad_max_non_chaotic": float(err_nc.max()) if err_nc.size else None,
           "frac_below_1e-5_rad_non_chaotic": float((err_nc < 1e-5).mean()) if err_nc.size else None,
           "non_chaotic_misses": np.nonzero(ok & ~chaotic & (err_all >= 1e-5))[0].tolist()}
    return base, acc


if __name__ == "__main__":
    main()
