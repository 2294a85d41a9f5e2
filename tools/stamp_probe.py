"""Developer probe: where the fused kernel spends its cycles (per-phase s_memtime stamps of thread 0)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads, _lib
B = 1024; mode = sys.argv[1] if len(sys.argv) > 1 else "CFS"
s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=B)
margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
slv = pkg.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=B)
lib = _lib.lib()
r = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs, noise=bt.noise if mode != "CFS" else None)   # warm
slv.stamps(B)
r = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs, noise=bt.noise if mode != "CFS" else None)
st = slv.stamps().astype(np.float64)
names = ["lin: base dist+minima+FD", "qp setup", "step1 scan", "w gather+roll", "d, r=Pd", "z+roll+refine", "steplen/update", "add", "drop", "post (roll,cost)", "lin: sincos+FK", "lin: shifted pairs"]
tot = st.sum(axis=1)
its = r.iter_O - 1; steps = r.total_iter
TICK = 1.0 / 21.0   # s_memtime counts shader cycles here (MI355X_MICROARCH.md: tick = shader cycle, ~2.1 GHz), not the 100 MHz the
                    # formulas below were written for: scale their 10 ns ticks by 1/21
print("s_memtime ticks = shader cycles (~2.1 GHz); times below are approximate (clock varies with load)")
for grp, m in (("all", np.ones(B, bool)), ("solved", r.status < 2), ("infeasible", r.status == 2)):
    t = st[m].sum(axis=0)
    print(f"[{grp}] problems {m.sum()}, outer its {its[m].sum()}, QP steps {steps[m].sum()}, total {t.sum()*1e-2/1e3*TICK:.1f} ms of WG time; per step {t[2:9].sum()*10/max(steps[m].sum(),1)/1e3*TICK:.2f} us; per-iteration linearise {(t[0]+t[10]+t[11])*10/max((its[m]+ (r.status[m]==2)).sum(),1)/1e3*TICK:.2f} us, post {t[9]*10/max(its[m].sum(),1)/1e3*TICK:.2f} us")
    print("   " + ", ".join(f"{n} {100*v/t.sum():.1f}%" for n, v in zip(names, t)))
worst = np.argsort(-tot)[:5]
print("slowest problems:", [(int(b), int(r.status[b]), int(r.iter_O[b]), int(steps[b]), round(tot[b]*1e-5*TICK, 2)) for b in worst], "(b, status, iter_O, steps, ms)")
print("sum of WG time / 256 CUs = %.2f ms" % (tot.sum() * 1e-5 * TICK / 256))
np.savez(f"gpurun_out/stamps_{mode}.npz", st=st, status=r.status, iter_O=r.iter_O, steps=r.total_iter)
for b in worst[:2]:
    t = st[b]
    print(f"problem {int(b)}: steps {int(steps[b])}, per step {t[2:9].sum()*TICK/100/max(steps[b],1):.2f} us: " + ", ".join(f"{n} {v*TICK/100/max(steps[b],1):.2f}" for n, v in zip(names[2:9], t[2:9])))
