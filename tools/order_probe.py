"""Developer probe: single-launch latency of config 3 with the identity launch order and with the longest-first order taken
from the first solve's total_iter (cfs_set_launch_order) -- the bound of what any scheduling heuristic can gain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads
B = 1024
s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=B)
dev = torch.device("cuda", 0)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()  # noqa: E731
for mode in ("CFS", "PSGCFS"):
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    sl = pkg.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=B)
    x_init, xR1, ff, caug, obs = t(bt.x_init), t(bt.xR1), t(bt.ff), t(bt.caug), t(bt.obs)
    noise = t(bt.noise) if mode == "PSGCFS" else None
    out = sl.alloc_outputs(B, dev)
    def lat(n=9):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sl.solve_device(x_init, xR1, ff, caug, obs, noise=noise, out=out)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return np.median(ts) * 1e3, np.min(ts) * 1e3
    sl.set_launch_order("identity")
    lat(3)
    base = lat()
    ti = out.total_iter.cpu().numpy().astype(np.int64)
    ref = out.x_.cpu().numpy().copy()
    res = {"identity": base}
    orders = {"longest_first": np.argsort(-ti, kind="stable"), "reverse_index": np.arange(B)[::-1].copy(),
              "shortest_first": np.argsort(ti, kind="stable")}
    for name, o in orders.items():
        sl.set_launch_order(o)
        res[name] = lat()
        assert np.array_equal(out.x_.cpu().numpy(), ref), name
    sl.set_launch_order("auto")
    res["auto"] = lat()
    assert np.array_equal(out.x_.cpu().numpy(), ref), "auto"
    print(mode, {k: "%.2f ms (min %.2f)" % v for k, v in res.items()}, "top total_iter", np.sort(ti)[-5:].tolist(), np.argsort(ti)[-5:].tolist(), flush=True)
    sl.close()
