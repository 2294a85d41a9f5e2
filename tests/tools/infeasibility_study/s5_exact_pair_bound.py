import numpy as np, re, sys
from scipy.optimize import linprog
z = np.load("gpurun_out/study/rows.npz")
bs, G, RH, racc, lim = z["bs"], z["g"], z["rh"], z["racc"], z["lim"]
nP, nobs, H, nj = G.shape
dt = 0.5
m = np.array([1, 1, np.pi, np.pi, np.pi]) * dt
idx = {int(b): p for p, b in enumerate(bs)}
pairs = []
for line in open("gpurun_out/study/minimal.txt"):
    mm = re.match(r"(\d+) \d+ \('pair', \((\d+), (\d+)\), \((\d+), (\d+)\)\)", line)
    if mm: pairs.append(tuple(int(v) for v in mm.groups()))
def joint_max(alpha, beta, i, k2, c):
    """max alpha*P_i + beta*P_k2 for joint c over its own trajectory polytope (exact LP)"""
    n = max(i, k2) + 1
    coef = lambda ii: np.array([((ii - k) + 0.5) * dt * dt if k <= ii else 0.0 for k in range(n)])
    obj = alpha * coef(i) + beta * coef(k2)
    Lv = np.tril(np.ones((n, n))) * dt
    A = np.vstack([Lv, -Lv]); b = np.full(2 * n, lim[c])
    r = linprog(-obj, A_ub=A, b_ub=b, bounds=[(-m[c], m[c])] * n, method="highs")
    return -r.fun
for (b, j1, i1, j2, i2) in pairs:
    p = idx[b]; g, rh = G[p], RH[p]
    ga, gb = g[j1, i1], g[j2, i2]
    ia, ib = 1 / np.linalg.norm(ga), 1 / np.linalg.norm(gb)
    ra, rb = -rh[j1, i1] * ia, -rh[j2, i2] * ib
    out = []
    for w in (0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 4.0):
        tot = sum(joint_max(ga[c] * ia, w * gb[c] * ib, i1, i2, c) for c in range(nj))
        out.append((w, tot - (ra + w * rb)))
    cosang = float(ga @ gb * ia * ib)
    print(b, (j1, i1), (j2, i2), "cos %.3f |ga| %.3f |gb| %.3f" % (cosang, 1 / ia, 1 / ib), " ".join("w%.2f:%+.3f" % o for o in out), flush=True)
