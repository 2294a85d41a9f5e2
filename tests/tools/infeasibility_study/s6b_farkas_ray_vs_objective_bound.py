import sys, os, json
sys.argv = [sys.argv[0]]
src = open(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "s6_farkas_ray.py")).read().split("def gi(")[0]
exec(src)
import concurrent.futures as cf
G_ = 0.5 * (s.QQ + s.QQ.T)
Gi = np.linalg.inv(G_)
FB = 6.11e7
def gi2(g0, C, b, m, every=8, start=16, maxit=3000):
    x = -Gi @ g0; x0 = x.copy()
    W = []; lam = np.zeros(0)
    steps = 0; fired = None; fb_stop = None; nmain = 0
    while True:
        # device-like checks at the top of the main loop
        if fb_stop is None:
            d = x - x0
            if 0.5 * d @ G_ @ d > FB: fb_stop = steps
        if fired is None and W and steps >= start and nmain % every == 0:
            rv = C[:, W] @ lam
            T = np.abs(rv) @ m - lam @ b[W]
            if T < -1e-9 * (np.abs(rv) @ m + abs(lam @ b[W])): fired = steps
        nmain += 1
        sl = C.T @ x - b
        sl[W] = np.inf
        p = int(np.argmin(sl))
        if sl[p] >= -1e-11 * (1 + abs(b[p])): return 0, steps, fired, fb_stop
        lp = 0.0; npv = C[:, p]
        while True:
            steps += 1
            if steps > maxit: return 3, steps, fired, fb_stop
            if W:
                N = C[:, W]; GN = Gi @ N; M = N.T @ GN
                r = np.linalg.solve(M, GN.T @ npv); zv = Gi @ npv - GN @ r
            else:
                r = np.zeros(0); zv = Gi @ npv
            zn = zv @ npv
            pos = np.nonzero(r > 1e-13 * (1 + np.abs(r).max() if r.size else 1))[0]
            if pos.size:
                ratios = lam[pos] / r[pos]; j = pos[int(np.argmin(ratios))]; t1 = ratios.min()
            else: t1 = np.inf; j = -1
            dep = zn <= 1e-8 * (npv @ (Gi @ npv))
            t2 = np.inf if dep else -(C[:, p] @ x - b[p]) / zn
            t = min(t1, t2)
            if not np.isfinite(t): return 2, steps, fired, fb_stop
            x = x + t * zv
            lam = lam - t * r; lp += t
            if t == t2 and not dep:
                W.append(p); lam = np.append(lam, lp); break
            W.pop(int(j)); lam = np.delete(lam, int(j))
todo = sorted(b for b in unc if study[b]["k"] == 1)
def one(b):
    s2 = SimpleNamespace(**vars(s)); s2.xR1, s2.robot = bt.xR1[b], orb
    u, x_ = np.zeros(nn), bt.x_init[b]
    A, rhs, dist, _, grad = O.get_con("M200i", s2, oracle_obs(bt, b, margin), x_, u, mode="CFS")
    keep = np.concatenate([np.arange(0, nobs * H * per, per), np.setdiff1d(np.arange(H * per), np.arange(0, H * per, per))])
    A = np.vstack([A[keep], np.eye(nn), -np.eye(nn)]); rh = np.concatenate([rhs[keep], s.MAX_input, s.MAX_input])
    out = {}
    for ev in (1, 8):
        st, steps, fired, fb = gi2(bt.ff[b], -A.T, -rh, s.MAX_input, every=ev, start=16 if ev > 1 else 0)
        end_dev = min(steps, fb if fb is not None else steps)
        out[ev] = (steps, fb, fired, end_dev, min(end_dev, fired if fired is not None else end_dev))
    return b, st, out
with cf.ThreadPoolExecutor(8) as ex:
    res = list(ex.map(one, todo))
for ev in (1, 8):
    dev = sum(r[2][ev][3] for r in res); new = sum(r[2][ev][4] for r in res)
    print("check every", ev, ": device-like steps", dev, "with the ray test", new, "saved %.1f %%" % (100.0 * (dev - new) / dev))
print([ (r[0], r[2][8]) for r in res][:12])
