"""BASELINE config 5 (mesh map, 50 waypoints, 256 seeds) on one GPU: one JSON line in bench.py's format.
A step = one complete batched solve (mesh linearisation + fused solver, one outer iteration per launch pair).
usage: python tests/tools/mesh_bench.py [--mode CFS|PSGCFS] [--steps K] [--warmup W] [--check N] [--streams S]"""
import argparse, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # one hardware queue per stream (see bench.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="PSGCFS"); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--batch", type=int, default=256); ap.add_argument("--tris", type=int, default=10000)
ap.add_argument("--check", type=int, default=0, help="compare the first N problems with the CPU oracle (brute force over the mesh: slow)")
ap.add_argument("--streams", type=int, default=4, help="independent batch solves in flight (one handle + HIP stream each): a solve is 20 x 7 small launches")
a = ap.parse_args()
dev = torch.device("cuda", 0)
s, bt, tri = workloads.config5(B=a.batch, n_tri=a.tris)
mesh = pkg.Mesh(tri=tri)
margin = bt.margin_cfs if a.mode == "CFS" else bt.margin_psg
S = max(1, a.streams)
slvs = [pkg.CFSBatch(s, 1, margin, mode=a.mode, max_batch=a.batch) for _ in range(S)]
for sl in slvs:
    sl.set_meshes([mesh])
slv = slvs[0]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
t = lambda x: torch.tensor(x, dtype=torch.float64, device=dev).contiguous()  # noqa: E731
x_init, xR1, ff, caug, obs = t(bt.x_init), t(bt.xR1), t(bt.ff), t(bt.caug), t(bt.obs)
noise = t(bt.noise) if a.mode == "PSGCFS" else None
outs = [sl.alloc_outputs(a.batch, dev) for sl in slvs]
out = outs[0]
torch.cuda.synchronize()

def step(i):
    k = i % S
    slvs[k].solve_device(x_init, xR1, ff, caug, obs, noise=noise, out=outs[k], stream=streams[k].cuda_stream)

for i in range(max(a.warmup, S)):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
units = int((out.iter_O - 1).sum().item())
st = np.bincount(out.status.cpu().numpy(), minlength=4)
H, nn = s.H, s.H * 5
algo = 8 * (H * 10 + nn + 1 * H * nn + H + 1 * H * nn + nn + H * 10)
res = {"metric": "CFS iterations/sec, 5-DoF 50-wp mesh-map batch-%d" % a.batch, "value": units / dt, "unit": "CFS iterations/s",
       "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3, "dtype": "f64", "data": "synthetic",
       "config": {"workload": "config5 (synthetic assembly-line mesh, %d triangles, H=50, %s)" % (tri.shape[0], a.mode),
                  "iterations_per_step": units, "solves_per_s": a.batch / dt, "concurrent_solves": S,
                  "status_counts": {"converged": int(st[0]), "max_iter": int(st[1]), "qp_infeasible": int(st[2]), "numeric": int(st[3])}},
       "roofline_convention": {"algorithmic_bytes_per_unit": algo, "achieved_GBs": algo * units / dt / 1e9, "frac_of_8TBs": algo * units / dt / 8e12}}
if a.check:
    from oracle import oracle as O
    l = O.mesh_register(0, tri)
    n = a.check
    oobs = np.tile(np.concatenate([l[:, 0], l[:, 1]]), (n, 1, 1))
    t1 = time.perf_counter()
    w = O.optimizer_batch(O.robotproperty2("M200i"), a.mode, H, 5, bt.x_init[:n], bt.xR1[:n], s.QQ, bt.ff[:n], bt.caug[:n], s.Aaug, s.Baug,
                          s.lim, s.MAX_input, oobs, margin, s.epsilon_O, s.MAX_O_ITER, s.alpha, noise=bt.noise[:n] if a.mode == "PSGCFS" else None)
    cpu_dt = time.perf_counter() - t1
    gs, gi, gx = out.status.cpu().numpy()[:n], out.iter_O.cpu().numpy()[:n], out.x_.cpu().numpy()[:n]
    same = (gs == w.status) & (gi == w.iter_O)
    ok = same & (gs < 2)
    err = np.abs(gx - w.x_).max(axis=1)[ok]
    res["accuracy"] = {"vs": "CPU oracle, brute force over the mesh (parity unpinned)", "problems": n, "status_and_iteration_agreement": float(same.mean()),
                       "linf_rad_max": float(err.max()) if err.size else None, "linf_rad_median": float(np.median(err)) if err.size else None}
    res["cpu_baseline"] = {"value": int((w.iter_O - 1).sum()) / cpu_dt, "unit": "CFS iterations/s", "cores": O.max_threads(), "kind": "port",
                           "sample": "%d problems, %.1f s" % (n, cpu_dt)}
print(json.dumps(res))
