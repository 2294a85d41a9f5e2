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
nn = H * nj
margin = bt.margin_cfs
dt = s.robot.delta_t
study = json.load(open("gpurun_out/study/infeasible.json"))
def run(K):
    return O.optimizer_batch(orb, "CFS", H, 5, bt.x_init, bt.xR1, s.QQ, bt.ff, bt.caug, s.Aaug, s.Baug, s.lim, s.MAX_input, bt.obs, margin, s.epsilon_O, K, s.alpha, nthreads=0)
ks = sorted(set(d["k"] for d in study))
prev = {}
for k in ks:
    if k == 1: continue
    w = run(k - 1)
    for d in study:
        if d["k"] == k: prev[d["b"]] = w.u[d["b"]].copy()
per = 1 + 2 * nj
coefm = np.array([[((i - k) + 0.5) * dt * dt if k <= i else 0.0 for k in range(H)] for i in range(H)])
racc = coefm @ s.MAX_input.reshape(H, nj)           # (H, nj)
res = []
data = {}
for d in study:
    b, k = d["b"], d["k"]
    s2 = SimpleNamespace(**vars(s)); s2.xR1, s2.robot = bt.xR1[b], orb
    if k == 1: u, x_ = np.zeros(nn), bt.x_init[b]
    else: u = prev[b]; x_ = O.rollout(H, nj, dt, bt.xR1[b], u)
    A, rhs, dist, _, grad = O.get_con("M200i", s2, oracle_obs(bt, b, margin), x_, u, mode="CFS")
    col = np.arange(0, nobs * H * per, per)
    rh = rhs[col].reshape(nobs, H)
    g = np.asarray(grad).reshape(nobs, H, nj)
    data[b] = (g, rh)
np.savez("gpurun_out/study/rows.npz", bs=np.array(list(data)), g=np.stack([data[b][0] for b in data]), rh=np.stack([data[b][1] for b in data]), racc=racc, v0=np.zeros(nj), lim=s.lim)
print("saved", len(data))
