"""Developer A/B probe: `python tools/ab_probe.py out.npz [libname.so] [flag ...]` solves config 3 in both modes with the given
build of the library (default libcfs_hip.so, a file next to it) and the given cfs_debug_set_options flags (names of _lib.DBG,
`warm_max=N`), saves results and timings; `python tools/ab_probe.py cmp a.npz b.npz` reports bitwise equality.  Run each
variant in its own process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True)
        extra = "" if same or a[k].dtype.kind != "f" else "  max|diff| %.3e  rows differing %d" % (
            np.nanmax(np.abs(a[k] - b[k])), int((np.abs(a[k] - b[k]).reshape(a[k].shape[0], -1).max(axis=1) > 0).sum()))
        print(f"{k:16s} {'bit-identical' if same else 'DIFFERENT'}{extra}")
    sys.exit(0)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # one hardware queue per stream, as bench.py
import torch
from motionplanning_5d_m_amd import _lib
LIBNAME = next((a for a in sys.argv[2:] if a.endswith(".so")), "libcfs_hip.so")
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), LIBNAME)          # before the first lib() call
FLAGS = {a: True for a in sys.argv[2:] if a in _lib.DBG}
WARM = next((int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("warm_max=")), 0)
import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import workloads
B = 1024
s, bt = workloads.config3(lambda rb, th, ob: pkg.dist_arm(rb, th, ob)[0], B=B)
dev = torch.device("cuda", 0)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev).contiguous()  # noqa: E731
res = {}
for mode in ("CFS", "PSGCFS"):
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    S = 8 if mode == "CFS" else 2            # as bench.py keeps them in flight
    slvs = [pkg.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=B) for _ in range(S)]
    for sl in slvs:
        sl.debug_options(warm_max=WARM, **FLAGS)
    x_init, xR1, ff, caug, obs = t(bt.x_init), t(bt.xR1), t(bt.ff), t(bt.caug), t(bt.obs)
    noise = t(bt.noise) if mode == "PSGCFS" else None
    outs = [sl.alloc_outputs(B, dev) for sl in slvs]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    def step(i):
        k = i % S
        with torch.cuda.stream(streams[k]):
            slvs[k].solve_device(x_init, xR1, ff, caug, obs, noise=noise, out=outs[k], stream=streams[k].cuda_stream)
    for i in range(S):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(0); torch.cuda.synchronize(); lat = time.perf_counter() - t0
    n, reps = 200, []
    for rep in range(7):                      # 7 blocks of 200 solves: the run-to-run noise of a 48-solve block was ~5 %
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / n)
    dt = float(np.median(reps))
    o = outs[0]
    its = int((o.iter_O - 1).sum().item())
    print(f"{LIBNAME} {sorted(FLAGS)} warm_max={WARM} {mode}: single {lat*1e3:.2f} ms, overlapped {dt*1e3:.3f} ms/solve (median of 7 x 200; min {min(reps)*1e3:.3f}), {its/dt:.3e} it/s, "
          f"status {np.bincount(o.status.cpu().numpy(), minlength=4).tolist()}", flush=True)
    for k in ("u", "x_", "status", "iter_O", "total_iter", "cost_all"):
        res[f"{mode}_{k}"] = getattr(o, k).cpu().numpy()
    for sl in slvs:
        sl.close()
np.savez(sys.argv[1], **res)
