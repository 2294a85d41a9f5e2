"""Study tool (CPU, uses the oracle): what do the infeasible linearisations of config 3 look like that the step-free
certificate of cfs_fused.hip does NOT catch?  For every problem the oracle ends with status 2, the failing QP is rebuilt
(oracle iterate before it), a phase-1 LP (scipy HiGHS) is solved, and the collision rows carrying Farkas weight are listed.
usage: python tests/tools/infeasible_study.py [B]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scipy.optimize import linprog
from types import SimpleNamespace
from oracle import oracle as O
from motionplanning_5d_m_amd import workloads
from helpers import oracle_obs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
orb = O.robotproperty2("M200i")
def dist_fn(rb, th, ob):
    return np.array([[O.dist_arm(orb, t, np.stack([o[:3], o[3:]], axis=1))[0] for o in ob] for t in th])
s, bt = workloads.config3(dist_fn, B=B)
H, nj, nobs = s.H, 5, bt.nobs
nn = H * nj
margin = bt.margin_cfs
def run(K):
    return O.optimizer_batch(orb, "CFS", H, 5, bt.x_init, bt.xR1, s.QQ, bt.ff, bt.caug, s.Aaug, s.Baug, s.lim, s.MAX_input, bt.obs, margin, s.epsilon_O, K, s.alpha, nthreads=0)
want = run(s.MAX_O_ITER)
bad = np.nonzero(want.status == 2)[0]
print("infeasible:", bad.size, "of", B, "; failing iteration histogram", np.bincount(want.iter_O[bad]))
# iterate before the failing QP: run with MAX_O_ITER = iter_O - 1
prev = {}
for k in sorted(set(want.iter_O[bad].tolist())):
    if k == 1: continue
    w = run(k - 1)
    for b in bad[want.iter_O[bad] == k]:
        prev[int(b)] = w.u[b].copy()
dt = s.robot.delta_t
out = []
for b in bad:
    b = int(b); k = int(want.iter_O[b])
    s2 = SimpleNamespace(**vars(s)); s2.xR1, s2.robot = bt.xR1[b], orb
    if k == 1: u, x_ = np.zeros(nn), bt.x_init[b]
    else: u = prev[b]; x_ = O.rollout(H, nj, dt, bt.xR1[b], u)
    A, rhs, dist, _, grad = O.get_con("M200i", s2, oracle_obs(bt, b, margin), x_, u, mode="CFS")
    per = 1 + 2 * nj
    col = np.arange(0, nobs * H * per, per)
    velrows = np.setdiff1d(np.arange(H * per), col[:H])       # obstacle 0's copy of the velocity rows
    Ac, bc = A[col], rhs[col]
    Av, bv = A[velrows], rhs[velrows]
    # phase 1: min t : Ac u - t <= bc, Av u <= bv, |u| <= MAX_input
    c = np.zeros(nn + 1); c[-1] = 1
    Aub = np.vstack([np.hstack([Ac, -np.ones((Ac.shape[0], 1))]), np.hstack([Av, np.zeros((Av.shape[0], 1))])])
    bub = np.concatenate([bc, bv])
    r = linprog(c, A_ub=Aub, b_ub=bub, bounds=[(-m, m) for m in s.MAX_input] + [(0, None)], method="highs")
    y = -r.ineqlin.marginals[:Ac.shape[0]]
    nz = np.nonzero(y > 1e-9)[0]
    rows = [(int(e // H), int(e % H), round(float(y[e]), 4)) for e in nz]      # (obstacle, waypoint, weight)
    nbox = int((np.abs(r.ineqlin.marginals[Ac.shape[0]:]) > 1e-9).sum()); nbnd = int((np.abs(r.lower.marginals[:nn]) + np.abs(r.upper.marginals[:nn]) > 1e-9).sum())
    out.append(dict(b=b, k=k, t=float(r.fun), rows=rows, nvel=nbox, nbnd=nbnd))
    print(b, "iter", k, "t* %.3e" % r.fun, "collision rows", rows, "vel rows", nbox, "bound rows", nbnd, flush=True)
json.dump(out, open("gpurun_out/study/infeasible.json", "w"))
