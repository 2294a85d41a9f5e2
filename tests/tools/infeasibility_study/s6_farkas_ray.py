import os, sys, json
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from types import SimpleNamespace
from oracle import oracle as O
from motionplanning_5d_m_amd import workloads
from helpers import oracle_obs
B = 1024
orb = O.robotproperty2("M200i")
def dist_fn(rb, th, ob):
    return np.array([[O.dist_arm(orb, t, np.stack([o[:3], o[3:]], axis=1))[0] for o in ob] for t in th])
s, bt = workloads.config3(dist_fn, B=B)
H, nj, nobs = s.H, 5, bt.nobs
nn = H * nj; per = 1 + 2 * nj
margin = bt.margin_cfs
study = {d["b"]: d for d in json.load(open("gpurun_out/study/infeasible.json"))}
z = np.load("gpurun_out/study/rows.npz"); unc = set(int(z["bs"][p]) for p in np.load("gpurun_out/study/unc.npy"))

def gi(G, g0, C, b, m, maxit=3000):
    """min 1/2 x'Gx + g0'x  s.t. C'x >= b.  Returns status, steps, first step at which the ray test fires."""
    Gi = np.linalg.inv(G)
    x = -Gi @ g0; x0 = x.copy()
    W = []; lam = np.zeros(0)
    steps = 0; fired = None; hist = []
    while True:
        sl = C.T @ x - b
        sl[W] = np.inf
        p = int(np.argmin(sl))
        if sl[p] >= -1e-11 * (1 + abs(b[p])): return 0, steps, fired, hist
        lp = 0.0
        npv = C[:, p]
        while True:
            steps += 1
            if steps > maxit: return 3, steps, fired, hist
            if W:
                N = C[:, W]; GN = Gi @ N; M = N.T @ GN
                r = np.linalg.solve(M, GN.T @ npv)
                zv = Gi @ npv - GN @ r
            else:
                r = np.zeros(0); zv = Gi @ npv
            zn = zv @ npv
            pos = np.nonzero(r > 1e-13 * (1 + np.abs(r).max() if r.size else 1))[0]
            if pos.size:
                ratios = lam[pos] / r[pos]; j = pos[int(np.argmin(ratios))]; t1 = ratios.min()
            else: t1 = np.inf; j = -1
            dep = zn <= 1e-10 * (npv @ (Gi @ npv))
            t2 = np.inf if dep else -(C[:, p] @ x - b[p]) / zn
            t = min(t1, t2)
            if not np.isfinite(t): return 2, steps, fired, hist
            if not dep: x = x + t * zv
            lam = lam - t * r; lp += t
            # ---- ray test with y = (lam, lp)
            y = np.concatenate([lam, [lp]]); idx = W + [p]
            rv = C[:, idx] @ y
            T = np.abs(rv) @ m - y @ b[idx]
            hist.append((steps, len(W), float(T), float(y @ b[idx]), float(np.abs(rv) @ m)))
            if T < 0 and fired is None: fired = steps
            if t == t2 and not dep:
                W.append(p); lam = np.append(lam, lp); break
            W.pop(int(j)); lam = np.delete(lam, int(j))

dt = s.robot.delta_t
todo = [int(a) for a in sys.argv[1:]] or sorted(b for b in unc if study[b]["k"] == 1)[:6]
for b in todo:
    s2 = SimpleNamespace(**vars(s)); s2.xR1, s2.robot = bt.xR1[b], orb
    u, x_ = np.zeros(nn), bt.x_init[b]
    A, rhs, dist, _, grad = O.get_con("M200i", s2, oracle_obs(bt, b, margin), x_, u, mode="CFS")
    keep = np.concatenate([np.arange(0, nobs * H * per, per), np.setdiff1d(np.arange(H * per), np.arange(0, H * per, per))])
    A = np.vstack([A[keep], np.eye(nn), -np.eye(nn)]); rh = np.concatenate([rhs[keep], s.MAX_input, s.MAX_input])
    st, steps, fired, hist = gi(s.QQ, bt.ff[b], -A.T, -rh, s.MAX_input)
    print("problem", b, "status", st, "steps", steps, "ray test first fires at step", fired, flush=True)
    for h in hist[:: max(1, len(hist) // 12)]: print("   step %4d  active %3d  T %.3e  y'b %.3e  |Cy|.m %.3e" % h)
