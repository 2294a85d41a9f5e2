import sys, os
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, copy
from oracle import oracle as O, rrt_oracle as R
import test_rrt
robot, obs, x0, goal, rg, rs, ratial = test_rrt._setup(O)
def emul(g, rh, lim, v0, racc, dt, weights=(1.0, 0.5, 2.0)):
    nobs, H, nj = g.shape
    f = dt * (np.arange(H) + 0.5)
    lo = np.maximum(f[:, None] * (-lim - v0)[None], -racc); hi = np.minimum(f[:, None] * (lim - v0)[None], racc)
    cen, rad = 0.5 * (lo + hi), np.maximum(0.5 * (hi - lo), 0)
    scen, srad = dt * (-v0), dt * lim
    n = np.linalg.norm(g, axis=-1); ok = n > 0
    gn = np.where(ok[..., None], g / np.where(ok, n, 1)[..., None], 0); r = np.where(ok, -rh / np.where(ok, n, 1), -np.inf)
    hits = []
    for i in range(H):
        for a in range(nobs):
            if not ok[a, i]: continue
            for b2 in range(a + 1, nobs):
                for w in weights:
                    cc = gn[a, i] + w * gn[b2, i]
                    if cc @ cen[i] + np.abs(cc) @ rad[i] < r[a, i] + w * r[b2, i] - 1e-9: hits.append(("same", i, a, b2, w))
            if i + 1 < H:
                for b2 in range(nobs):
                    if not ok[b2, i + 1]: continue
                    for w in weights:
                        cb = w * gn[b2, i + 1]; cc = gn[a, i] + cb
                        lhs = cc @ cen[i] + np.abs(cc) @ rad[i] + cb @ scen + np.abs(cb) @ srad
                        if lhs < r[a, i] + w * r[b2, i + 1] - 1e-9: hits.append(("adj", i, a, b2, w, round(float(lhs - (r[a, i] + w * r[b2, i + 1])), 4)))
    return hits
for sd in (1, 5, 6, 7, 8):
    r = R.find_route(robot, obs, x0, goal, goal, rg, rs, np.zeros(5), ratial, np.random.default_rng(sd), "RRT")
    P = O.problem_RRTstar_CFS(r["route"]); s = P.sys_info
    s1 = copy.copy(s); s1.MAX_O_ITER = 1
    w1 = O.optimizer(P.ROBOT, s1, P.obs, "CFS")
    H, nj = s.H, 5; nn = H * nj; dt = 0.5
    A, rhs, dist, lid, grad = O.get_con(P.ROBOT, s, P.obs, w1.x_, w1.u, mode="CFS")
    per = 11; col = np.arange(0, A.shape[0], per)
    g = np.asarray(grad).reshape(2, H, nj)
    # the kernel's right side refers to the position offset P = Bpos u (absolute), as rhs from get_con does: rh = rhs (A u <= rhs with A = -coef x g)
    rh = rhs[col].reshape(2, H)
    coefm = np.array([[((i - k) + 0.5) * dt * dt if k <= i else 0.0 for k in range(H)] for i in range(H)])
    racc = coefm @ s.MAX_input.reshape(H, nj)
    v0 = np.asarray(s.xR1).ravel()[5:]
    hits = emul(g, rh, s.lim, v0, racc, dt)
    wq = O.optimizer(P.ROBOT, s, P.obs, "CFS")
    print("seed", sd, "oracle status", wq.status, "iter", wq.iter_O, "QP steps total", getattr(wq, "total_iter", None), "| certificate hits on QP 2:", hits[:4])
