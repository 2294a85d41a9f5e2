import numpy as np, itertools, sys
from scipy.optimize import linprog
z = np.load("gpurun_out/study/rows.npz")
bs, G, RH, racc, lim = z["bs"], z["g"], z["rh"], z["racc"], z["lim"]
unc = np.load("gpurun_out/study/unc.npy")
nP, nobs, H, nj = G.shape
dt = 0.5; nn = H * nj
coefm = np.array([[((i - k) + 0.5) * dt * dt if k <= i else 0.0 for k in range(H)] for i in range(H)])
MAXI = np.tile(np.array([1, 1, np.pi, np.pi, np.pi]) * dt, H)
# velocity rows
L = np.kron(np.tril(np.ones((H, H))), np.eye(nj)) * dt
Av = np.vstack([L, -L]); bv = np.concatenate([np.tile(lim, H), np.tile(lim, H)])
def row(g, i):
    return -(coefm[i][:, None] * g[None, :]).reshape(-1)
def feasible(rows, g, rh):
    A = np.vstack([Av] + [row(g[j, i], i)[None] for j, i in rows]); b = np.concatenate([bv, [rh[j, i] for j, i in rows]])
    r = linprog(np.zeros(nn), A_ub=A, b_ub=b, bounds=[(-m, m) for m in MAXI], method="highs")
    return r.status == 0
from collections import Counter
kinds = Counter()
for p in unc[: int(sys.argv[1]) if len(sys.argv) > 1 else len(unc)]:
    g, rh = G[p], RH[p]
    V = [(j, i) for j in range(nobs) for i in range(H) if rh[j, i] < 0.02 and np.abs(g[j, i]).max() > 0]
    found = None
    for r1 in V:
        if not feasible([r1], g, rh): found = ("single", r1); break
    if not found:
        for r1, r2 in itertools.combinations(V, 2):
            if abs(r1[1] - r2[1]) > 6: continue
            if not feasible([r1, r2], g, rh): found = ("pair", r1, r2); break
    if not found:
        VV = sorted(V, key=lambda e: rh[e])[:14]
        for tr in itertools.combinations(VV, 3):
            if not feasible(list(tr), g, rh): found = ("triple",) + tr; break
    if not found:
        found = ("all" if not feasible(V, g, rh) else "beyond V", len(V))
    kinds[found[0] + ("" if found[0] not in ("pair",) else " dk=%d same_obs=%d" % (abs(found[1][1] - found[2][1]), found[1][0] == found[2][0]))] += 1
    print(int(bs[p]), len(V), found, [round(float(rh[e]), 3) for e in found[1:] if isinstance(e, tuple)], flush=True)
print(kinds)
