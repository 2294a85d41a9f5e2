"""Developer probe: first outer iteration at which the device solve and the oracle part ways."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads
from oracle import oracle as O
mode = sys.argv[1]; probs = [int(a) for a in sys.argv[2:]]
s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=1024)
margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
robot = O.robotproperty2("M200i")
for b in probs:
    sl = slice(b, b + 1)
    w = O.optimizer_batch(robot, mode, s.H, 5, bt.x_init[sl], bt.xR1[sl], s.QQ, bt.ff[sl], bt.caug[sl], s.Aaug, s.Baug, s.lim,
                          s.MAX_input, bt.obs[sl], margin, s.epsilon_O, 20, s.alpha, noise=bt.noise[sl] if mode != "CFS" else None)
    print(f"problem {b}: oracle status {w.status[0]} iter_O {w.iter_O[0]}")
    for k in range(1, int(w.iter_O[0])):
        s2 = copy.copy(s); s2.MAX_O_ITER = k
        slv = pkg.CFSBatch(s2, bt.nobs, margin, mode=mode, max_batch=1)
        g = slv.solve(bt.x_init[sl], bt.xR1[sl], bt.ff[sl], bt.caug[sl], bt.obs[sl], noise=bt.noise[sl] if mode != "CFS" else None)
        wk = O.optimizer_batch(robot, mode, s.H, 5, bt.x_init[sl], bt.xR1[sl], s.QQ, bt.ff[sl], bt.caug[sl], s.Aaug, s.Baug, s.lim,
                               s.MAX_input, bt.obs[sl], margin, s.epsilon_O, k, s.alpha, noise=bt.noise[sl] if mode != "CFS" else None)
        du = np.abs(g.u - wk.u).max(); dx = np.abs(g.x_ - wk.x_).max()
        print(f"  after {k:2d} iterations: status {g.status[0]}/{wk.status[0]} linf u {du:.3e} x_ {dx:.3e} steps {g.total_iter[0]}/{wk.total_iter[0]}")
        slv.close()
        if dx > 1e-3: break
