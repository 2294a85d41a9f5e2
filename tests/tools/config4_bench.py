"""BASELINE config 4 (RRT*-CFS smoothing stage, H=40, 2 obstacles, 4096 routes) on one GPU: one JSON line in bench.py's format.
The batch is B perturbed copies of the logged RRT route (tests/golden/route_wp_200i_xori.npy = data/200i_xori.mat:route_wp),
resampled on the device (cfs_build_terms_from_routes_device) and smoothed by CFS_FANUC.
usage: python tests/tools/config4_bench.py [--batch B] [--steps K] [--warmup W] [--check N] [--streams S]"""
import argparse, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # one hardware queue per stream (see bench.py)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--check", type=int, default=0)
ap.add_argument("--streams", type=int, default=2, help="independent batch solves in flight (one handle + HIP stream each): the next launch fills the tail of the previous one")
a = ap.parse_args()
route = np.load(os.path.join(ROOT, "tests", "golden", "route_wp_200i_xori.npy"))
s, bt = workloads.config4(route, B=a.batch)
dev = torch.device("cuda", 0)
t = lambda x: torch.tensor(x, dtype=torch.float64, device=dev).contiguous()  # noqa: E731
S = max(1, a.streams)
slvs = [pkg.CFSBatch(s, bt.nobs, bt.margin_cfs, mode="CFS", max_batch=a.batch, use_weights=True) for _ in range(S)]   # cfs_problem_create_from_weights:
slv = slvs[0]                                            # neither QQ nor Qaug crosses the boundary, the state-cost terms come with it
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
rng = np.random.default_rng(20260104)
routes = t((route[None] + 0.02 * rng.standard_normal((a.batch,) + route.shape)).transpose(0, 2, 1).copy())   # (B, nwp, 5), as config4 draws them
obs = t(bt.obs)
outs = [sl.alloc_outputs(a.batch, dev) for sl in slvs]
out = outs[0]
torch.cuda.synchronize()

def step(i):
    k = i % S
    with torch.cuda.stream(streams[k]):
        st_ = streams[k].cuda_stream
        x_init, xR1, ff, caug = slvs[k].build_terms_from_routes_device(routes, stream=st_)   # resampling + cost terms on the device (row f2)
        slvs[k].solve_device(x_init, xR1, ff, caug, obs, out=outs[k], stream=st_)
    return x_init, xR1, ff, caug

for i in range(max(a.warmup, S)):
    terms = step(i)
torch.cuda.synchronize()
terms = step(0)                                          # the terms the parity check below reads belong to outs[0]
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
units = int((out.iter_O - 1).sum().item())
st = np.bincount(out.status.cpu().numpy(), minlength=4)
H, nn = s.H, s.H * 5
algo = 8 * (H * 10 + nn + 2 * H * nn + 2 * H + 2 * H * nn + nn + H * 10)
res = {"metric": "CFS iterations/sec, 5-DoF 40-wp 2-obs RRT-route batch-%d" % a.batch, "value": units / dt, "unit": "CFS iterations/s", "n_gpus": 1,
       "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3, "dtype": "f64", "data": "synthetic",
       "config": {"workload": "config4: %d jittered copies of the logged RRT route, cubic resampling to H=40 on the device, CFS_FANUC, 2 obstacles" % a.batch,
                  "iterations_per_step": units, "solves_per_s": a.batch / dt, "concurrent_solves": S,
                  "status_counts": {"converged": int(st[0]), "max_iter": int(st[1]), "qp_infeasible": int(st[2]), "numeric": int(st[3])}},
       "roofline_convention": {"algorithmic_bytes_per_unit": algo, "achieved_GBs": algo * units / dt / 1e9, "frac_of_8TBs": algo * units / dt / 8e12}}
if a.check:
    from oracle import oracle as O
    n = a.check
    x_init, xR1, ff, caug = [v.cpu().numpy() for v in terms]
    t1 = time.perf_counter()
    w = O.optimizer_batch(O.robotproperty2("M200i"), "CFS", H, 5, x_init[:n], xR1[:n], s.QQ, ff[:n], caug[:n], s.Aaug, s.Baug, s.lim, s.MAX_input,
                          bt.obs[:n], bt.margin_cfs, s.epsilon_O, s.MAX_O_ITER, s.alpha)
    cpu_dt = time.perf_counter() - t1
    gs, gi, gx = out.status.cpu().numpy()[:n], out.iter_O.cpu().numpy()[:n], out.x_.cpu().numpy()[:n]
    same = (gs == w.status) & (gi == w.iter_O)
    ok = same & (gs < 2)
    err = np.abs(gx - w.x_).max(axis=1)[ok]
    res["accuracy"] = {"vs": "CPU oracle (parity unpinned)", "problems": n, "status_and_iteration_agreement": float(same.mean()),
                       "linf_rad_median": float(np.median(err)) if err.size else None, "linf_rad_p99": float(np.quantile(err, 0.99)) if err.size else None,
                       "frac_below_1e-5_rad": float((err < 1e-5).mean()) if err.size else None}
    res["cpu_baseline"] = {"value": int((w.iter_O - 1).sum()) / cpu_dt, "unit": "CFS iterations/s", "cores": O.max_threads(), "kind": "port",
                           "sample": "%d problems, %.1f s" % (n, cpu_dt)}
print(json.dumps(res))
