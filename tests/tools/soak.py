"""Developer soak: N solves per handle of config 3 on several handles and streams at once (both solvers interleaved), every
output compared on the device with the first one of its handle -- any race between launches (spill pool, launch order buffers,
warm-start state) would show as a differing bit.  usage: python tests/tools/soak.py [solves per handle]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B = 1024
s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=B)
dev = torch.device("cuda", 0)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()  # noqa: E731
x_init, xR1, ff, caug, obs, noise = t(bt.x_init), t(bt.xR1), t(bt.ff), t(bt.caug), t(bt.obs), t(bt.noise)
H = []
for mode in ("PSGCFS", "CFS", "PSGCFS", "CFS", "CFS"):
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    slv = pkg.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=B)
    st = torch.cuda.Stream(device=dev)
    H.append(dict(mode=mode, slv=slv, st=st, out=slv.alloc_outputs(B, dev), ref=None, bad=0))
keys = ("u", "x_", "status", "iter_O", "total_iter", "cost_all")
for i in range(N):
    for h in H:
        with torch.cuda.stream(h["st"]):
            h["slv"].solve_device(x_init, xR1, ff, caug, obs, noise=noise if h["mode"] == "PSGCFS" else None, out=h["out"], stream=h["st"].cuda_stream)
            if h["ref"] is None:
                h["ref"] = {k: getattr(h["out"], k).clone() for k in keys}
            else:
                same = all(torch.equal(getattr(h["out"], k), h["ref"][k]) for k in keys)      # stream-ordered: compares this solve's output
                h["bad"] += (not same)
    if (i + 1) % 100 == 0:
        torch.cuda.synchronize()
        print(f"{i + 1} solves per handle, mismatches so far: {[h['bad'] for h in H]}", flush=True)
torch.cuda.synchronize()
a, b = H[0]["ref"], H[2]["ref"]
print("the two PSGCFS handles agree:", all(torch.equal(a[k], b[k]) for k in keys), "; the CFS handles:", all(torch.equal(H[1]["ref"][k], H[3]["ref"][k]) and torch.equal(H[1]["ref"][k], H[4]["ref"][k]) for k in keys))
print("mismatching solves:", sum(h["bad"] for h in H), "of", N * len(H))
