import numpy as np, itertools
z = np.load("gpurun_out/study/rows.npz")
bs, G, RH, racc, lim = z["bs"], z["g"], z["rh"], z["racc"], z["lim"]
nP, nobs, H, nj = G.shape
dt = 0.1 if False else None
# dt: racc[0] = 0.5 dt^2 MAX_input[0]; MAX_input = [1,1,pi,pi,pi]*dt -> racc[0,0] = 0.5 dt^3
dt = (2 * racc[0, 0]) ** (1 / 3)
print("dt", dt)
v0 = np.zeros(nj)
f = dt * (np.arange(H) + 0.5)
lo = np.maximum(f[:, None] * (-lim - v0)[None], -racc); hi = np.minimum(f[:, None] * (lim - v0)[None], racc)
cen, rad = 0.5 * (lo + hi), np.maximum(0.5 * (hi - lo), 0)
slo, shi = dt * (-lim - v0), dt * (lim - v0)
scen, srad = 0.5 * (slo + shi), 0.5 * (shi - slo)

def norm_rows(g, rh):
    n = np.linalg.norm(g, axis=-1)
    ok = n > 0
    gn = np.where(ok[..., None], g / np.where(ok, n, 1)[..., None], 0)
    r = np.where(ok, -rh / np.where(ok, n, 1), -np.inf)
    return gn, r, ok

def current(g, rh, weights=(1.0, 0.5, 2.0)):
    gn, r, ok = norm_rows(g, rh)
    for i in range(H):
        for a in range(nobs):
            if not ok[a, i]: continue
            for b2 in range(a + 1, nobs):
                if not ok[b2, i]: continue
                for w in weights:
                    cc = gn[a, i] + w * gn[b2, i]
                    lhs = cc @ cen[i] + np.abs(cc) @ rad[i]
                    rhs_ = r[a, i] + w * r[b2, i]
                    if lhs < rhs_ - 1e-9 * (1 + abs(rhs_)): return ("same", i, a, b2, w)
            if i + 1 < H:
                for b2 in range(nobs):
                    if not ok[b2, i + 1]: continue
                    for w in weights:
                        cb = w * gn[b2, i + 1]; cc = gn[a, i] + cb
                        lhs = cc @ cen[i] + np.abs(cc) @ rad[i] + cb @ scen + np.abs(cb) @ srad
                        rhs_ = r[a, i] + w * r[b2, i + 1]
                        if lhs < rhs_ - 1e-9 * (1 + abs(rhs_)): return ("adj", i, a, b2, w)
    return None

def single(g, rh):
    gn, r, ok = norm_rows(g, rh)
    for i in range(H):
        for a in range(nobs):
            if ok[a, i] and (gn[a, i] @ cen[i] + np.abs(gn[a, i]) @ rad[i] < r[a, i] - 1e-9): return ("single", i, a)
    return None

def anypair(g, rh, K=H, weights=(1.0, 0.5, 2.0, 0.25, 4.0, 0.75, 1.5)):
    gn, r, ok = norm_rows(g, rh)
    for i in range(H):
        for a in range(nobs):
            if not ok[a, i]: continue
            for k in range(0, K):
                if i + k >= H: break
                for b2 in range(nobs):
                    if not ok[b2, i + k] or (k == 0 and b2 <= a): continue
                    for w in weights:
                        cb = w * gn[b2, i + k]; cc = gn[a, i] + cb
                        lhs = cc @ cen[i] + np.abs(cc) @ rad[i] + k * (cb @ scen + np.abs(cb) @ srad)
                        rhs_ = r[a, i] + w * r[b2, i + k]
                        if lhs < rhs_ - 1e-9 * (1 + abs(rhs_)): return ("pair", i, k, a, b2, w)
    return None

caught = [current(G[p], RH[p]) for p in range(nP)]
print("current certificate catches", sum(c is not None for c in caught), "of", nP)
unc = [p for p in range(nP) if caught[p] is None]
s1 = [single(G[p], RH[p]) for p in unc]
print("single row catches of uncaught:", sum(c is not None for c in s1))
ap = [anypair(G[p], RH[p]) for p in unc]
print("any pair / more weights catches of uncaught:", sum(c is not None for c in ap))
from collections import Counter
print(Counter((c[0], c[2]) if c else None for c in ap))
print(Counter(c[-1] if c else None for c in ap))
np.save("gpurun_out/study/unc.npy", np.array([p for p, c in zip(unc, ap) if c is None]))
