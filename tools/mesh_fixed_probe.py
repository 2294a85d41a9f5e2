"""Developer probe: time of the mesh linearisation pipeline against trivial meshes (fixed costs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads, mesh as M
s, bt, tri = workloads.config5(B=256)
dev = torch.device("cuda", 0)
t = lambda x: torch.tensor(x, dtype=torch.float64, device=dev).contiguous()
base = s.robot.base
cases = {"far box 12 tris": M.box_mesh(base + np.array([5, 5, 0]), base + np.array([6, 6, 1]), n=1),
         "near box 12 tris": M.box_mesh(base + np.array([0.55, -0.9, -0.33]), base + np.array([1.05, 0.9, 0.12]), n=1),
         "near box 4800 tris": M.box_mesh(base + np.array([0.55, -0.9, -0.33]), base + np.array([1.05, 0.9, 0.12]), n=20),
         "3 spheres 960 tris": np.concatenate([M.icosphere(base + np.array([0.8, y, 0.2]), 0.08, subdiv=2) for y in (-0.5, 0.0, 0.5)]),
         "1 sphere 5120 tris": M.icosphere(base + np.array([0.8, 0.0, 0.2]), 0.08, subdiv=4),
         "cylinder 240 tris": M.cylinder_mesh((base[0] + 0.8, base[1] + 0.35), 0.06, base[2] + 0.12, base[2] + 0.42, nseg=24, nring=4),
         "posts+beam": np.concatenate([M.box_mesh(base + np.array([0.50, -0.95, -0.33]), base + np.array([0.58, -0.87, 1.25]), n=10),
                                       M.box_mesh(base + np.array([0.50, 0.87, -0.33]), base + np.array([0.58, 0.95, 1.25]), n=10),
                                       M.box_mesh(base + np.array([0.50, -0.95, 1.25]), base + np.array([0.58, 0.95, 1.33]), n=10)]),
         "config5": tri}
for name, tr in cases.items():
    mesh = pkg.Mesh(tri=tr)
    slv = pkg.CFSBatch(s, 1, bt.margin_psg, mode="PSGCFS", max_batch=256)
    slv.set_meshes([mesh])
    args = (t(bt.x_init), t(bt.xR1), t(bt.ff), t(bt.caug), t(bt.obs))
    out = slv.alloc_outputs(256, dev)
    slv.solve_device(*args, noise=t(bt.noise), out=out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        slv.solve_device(*args, noise=t(bt.noise), out=out)
    torch.cuda.synchronize()
    print(f"{name:22s} {tr.shape[0]:6d} tris: {(time.perf_counter()-t0)/3*1e3:7.2f} ms per solve (20 iterations)", flush=True)
    slv.close(); mesh.close()
