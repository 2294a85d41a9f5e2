"""Developer probe: first-outer-iteration u of the fused kernel vs the oracle (config 3 and config-4 shape), plus the
full solves; everything is saved to gpurun_out/iter1_probe.npz for offline analysis."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads
from oracle import oracle as O

out = {}
def run(tag, s, bt, mode, K):
    s2 = copy.copy(s); s2.MAX_O_ITER = K
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    B = bt.x_init.shape[0]
    slv = pkg.CFSBatch(s2, bt.nobs, margin, mode=mode, max_batch=B)
    nz = bt.noise if (mode == "PSGCFS" and bt.noise is not None) else None
    g = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs, noise=nz)
    slv.close()
    t0 = time.time()
    w = O.optimizer_batch(O.robotproperty2("M200i"), mode, s.H, 5, bt.x_init, bt.xR1, s.QQ, bt.ff, bt.caug, s.Aaug, s.Baug,
                          s.lim, s.MAX_input, bt.obs, margin, s.epsilon_O, K, s.alpha, noise=nz, nthreads=0)
    same = (g.status == w.status) & (g.iter_O == w.iter_O)
    ok = same & (g.status < 2)
    eu = np.abs(g.u - w.u).max(axis=1) / np.maximum(np.abs(w.u).max(axis=1), 1e-300)
    ex = np.abs(g.x_ - w.x_).max(axis=1)
    print(f"[{tag} {mode} K={K}] oracle {time.time()-t0:.1f}s agree {same.sum()}/{B} solved {ok.sum()} status gpu {np.bincount(g.status, minlength=4)} orc {np.bincount(w.status, minlength=4)}")
    print(f"   rel u err: median {np.median(eu[ok]):.2e} p99 {np.quantile(eu[ok], .99):.2e} max {eu[ok].max():.2e}  >1e-9: {(eu[ok] > 1e-9).sum()}  >1e-7: {(eu[ok]>1e-7).sum()}; x err max {ex[ok].max():.2e} >1e-5: {(ex[ok]>1e-5).sum()}")
    worst = np.argsort(-np.where(ok, eu, 0))[:12]
    print("   worst:", [(int(b), f"{eu[b]:.1e}", f"{ex[b]:.1e}", int(g.total_iter[b]), int(w.total_iter[b])) for b in worst])
    print("   mismatch:", [(int(b), int(g.status[b]), int(w.status[b]), int(g.iter_O[b]), int(w.iter_O[b])) for b in np.nonzero(~same)[0][:10]])
    for k, v in (("gu", g.u), ("gx", g.x_), ("gst", g.status), ("git", g.iter_O), ("gtot", g.total_iter),
                 ("wu", w.u), ("wx", w.x_), ("wst", w.status), ("wit", w.iter_O), ("wtot", w.total_iter)):
        out[f"{tag}_{mode}_{K}_{k}"] = v

s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=1024)
for mode in ("CFS", "PSGCFS"):
    run("c3", s, bt, mode, 1)
    run("c3", s, bt, mode, 20)
route = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "route_wp_200i_xori.npy"))
s4, bt4 = workloads.config4(route, B=512)
run("c4", s4, bt4, "CFS", 1)
run("c4", s4, bt4, "CFS", 20)
np.savez_compressed("gpurun_out/iter1_probe.npz", **out)
