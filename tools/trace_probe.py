"""Developer probe: trace the device active-set steps of one problem of the config-3 batch."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads, _lib
B = 1024; b = int(sys.argv[1]); mode = sys.argv[2]; cap = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=B)
margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
slv = pkg.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=1)
lib = _lib.lib()
slv.trace(0, cap)
sl = slice(b, b + 1)
r = slv.solve(bt.x_init[sl], bt.xR1[sl], bt.ff[sl], bt.caug[sl], bt.obs[sl], noise=bt.noise[sl] if mode != "CFS" else None)
rec = slv.trace(); n = rec.shape[0]
print("status", r.status, "iter_O", r.iter_O, "total_iter", r.total_iter, "records", n)
np.save(f"gpurun_out/trace_{mode}_{b}.npy", rec)
for k in list(range(min(n, 70))) + list(range(max(70, n - 25), n)):
    print("it %7d q %3d p %4d sp %.3e d/spp %.3e t1 %.3e t2 %.3e l %d" % tuple(rec[k]))
