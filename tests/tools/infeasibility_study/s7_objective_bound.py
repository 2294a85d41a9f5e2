import os, sys, json
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from types import SimpleNamespace
from oracle import oracle as O
from motionplanning_5d_m_amd import workloads
from helpers import oracle_obs
sys.path.insert(0, "gpurun_out/study")
B = 1024
orb = O.robotproperty2("M200i")
def dist_fn(rb, th, ob):
    return np.array([[O.dist_arm(orb, t, np.stack([o[:3], o[3:]], axis=1))[0] for o in ob] for t in th])
s, bt = workloads.config3(dist_fn, B=B)
H, nj, nobs = s.H, 5, bt.nobs
nn = H * nj; per = 1 + 2 * nj; dt = 0.5
G = 0.5 * (s.QQ + s.QQ.T)
lmaxH = np.linalg.eigvalsh(G).max()
# s = Bvel u  (Bvel = dt * cumsum): u = D s / dt with D the difference matrix
L = np.kron(np.tril(np.ones((H, H))), np.eye(nj)) * dt
Li = np.linalg.inv(L)
Gs = Li.T @ G @ Li
lmaxV = np.linalg.eigvalsh(Gs).max()
print("lmax(H) %.3e  lmax_vel %.3e" % (lmaxH, lmaxV))
b = 39
x0 = -np.linalg.solve(G, bt.ff[b]); s0 = L @ x0
ru = np.linalg.norm(s.MAX_input) + np.linalg.norm(x0); rs = np.linalg.norm(np.tile(s.lim, H)) + np.linalg.norm(s0)
print("ru %.2f rs %.2f  fbound_u %.3e fbound_s %.3e" % (ru, rs, 0.5 * lmaxH * ru * ru, 0.5 * lmaxV * rs * rs))
# exact-ish: max over vel box corners by power-like heuristics (lower bound of the true max) to see how loose lmax*r^2 is
rng = np.random.default_rng(0)
best = 0
w = np.tile(s.lim, H)
for _ in range(200):
    sg = np.sign(rng.standard_normal(nn))
    for it in range(50):
        d = sg * w - s0
        sg2 = np.sign(Gs @ d); sg2[sg2 == 0] = 1
        if (sg2 == sg).all(): break
        sg = sg2
    d = sg * w - s0
    best = max(best, 0.5 * d @ Gs @ d)
print("max over the velocity box of f - f(x0) (local search): %.3e" % best)
