"""Developer probe: how fast the objective gain approaches the infeasibility bound (per infeasible problem)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads, _lib
s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=1024)
slv = pkg.CFSBatch(s, bt.nobs, bt.margin_cfs, mode="CFS", max_batch=1024)
full = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs)
inf = np.nonzero((full.status == 2) & (full.iter_O == 1))[0][:24]
one = pkg.CFSBatch(s, bt.nobs, bt.margin_cfs, mode="CFS", max_batch=1)
lib = _lib.lib(); cap = 1500
for b in inf:
    one.trace(0, cap)
    sl = slice(b, b + 1)
    r = one.solve(bt.x_init[sl], bt.xR1[sl], bt.ff[sl], bt.caug[sl], bt.obs[sl])
    rec = one.trace(); n = rec.shape[0]; ratio = rec[:, 7]
    cross = [int(np.argmax(ratio > th)) if (ratio > th).any() else -1 for th in (1e-8, 1e-6, 1e-4, 1e-2, 1e-1)]
    print(f"problem {b}: steps {n}, first step with fgain/fbound > 1e-8,1e-6,1e-4,1e-2,1e-1: {cross}, final ratio {ratio[-1]:.2e}, q_end {int(rec[-1,1])}")
