import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads
s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=1024)
slv = pkg.CFSBatch(s, bt.nobs, bt.margin_cfs, mode="CFS", max_batch=1024)
for _ in range(3):
    r = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs)
print(np.bincount(r.status))
