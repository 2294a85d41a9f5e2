"""Developer probe: per-phase cycle stamps of the fused kernel on config 4 (4096 RRT routes, H = 40, CFS_FANUC)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
route = np.load(os.path.join(ROOT, "tests", "golden", "route_wp_200i_xori.npy"))
s, bt = workloads.config4(route, B=B)
slv = pkg.CFSBatch(s, bt.nobs, bt.margin_cfs, mode="CFS", max_batch=B)
lib = _lib.lib()
r = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs)
slv.stamps(B)
r = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs)
st = slv.stamps().astype(np.float64)
names = ["lin: base dist+minima+FD", "qp setup", "step1 scan", "w gather+roll", "d, r=Pd", "z+roll+refine", "steplen/update", "add", "drop", "post (roll,cost)", "lin: sincos+FK", "lin: shifted pairs"]
TICK = 1.0 / 21.0
its = r.iter_O - 1; steps = r.total_iter
for grp, m in (("all", np.ones(B, bool)), ("solved", r.status < 2), ("infeasible", r.status == 2)):
    t = st[m].sum(axis=0)
    print(f"[{grp}] problems {m.sum()}, outer its {its[m].sum()}, QP steps {steps[m].sum()}, total {t.sum()*1e-2/1e3*TICK:.1f} ms of WG time; per step {t[2:9].sum()*10/max(steps[m].sum(),1)/1e3*TICK:.2f} us")
    print("   " + ", ".join(f"{n} {100*v/t.sum():.1f}%" for n, v in zip(names, t)))
tot = st.sum(axis=1)
worst = np.argsort(-tot)[:5]
print("slowest:", [(int(b), int(r.status[b]), int(r.iter_O[b]), int(steps[b]), round(tot[b]*1e-5*TICK, 2)) for b in worst], "(b, status, iter_O, steps, ms)")
print("steps histogram of infeasible problems (last QP included):", np.percentile(steps[r.status == 2], [10, 50, 90, 99]).tolist())
