"""Developer probe (GPU): per-problem workgroup clocks and active-set steps of config 3, both solvers -> gpurun_out/order_probe.npz
(input of the launch-order study: which cheap key predicts a problem's duration best)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads
s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=1024)
out = {}
for mode in ("CFS", "PSGCFS"):
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    slv = pkg.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=1024)
    nz = bt.noise if mode == "PSGCFS" else None
    slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs, noise=nz)
    slv.stamps(1024)
    r = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs, noise=nz)
    st = slv.stamps()
    out[mode + "_clocks"], out[mode + "_steps"], out[mode + "_status"], out[mode + "_iter"] = st.sum(axis=1), r.total_iter, r.status, r.iter_O
    slv.close()
# distances of the initial trajectories to every obstacle (what the pre-pass sees)
th = bt.x_init.reshape(1024, s.H, 10)[:, :, :5]
d = np.stack([pkg.dist_arm(s.robot, th[b], bt.obs[b])[0] for b in range(1024)])      # (B, H, nobs)
out["dist0"] = d
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "order_probe.npz"), **out)
print("saved", {k: v.shape for k, v in out.items()})
