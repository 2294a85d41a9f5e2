import sys, os
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, copy
from scipy.optimize import linprog
from oracle import oracle as O, rrt_oracle as R
import test_rrt
robot, obs, x0, goal, rg, rs, ratial = test_rrt._setup(O)
for sd in (1, 5, 6):
    r = R.find_route(robot, obs, x0, goal, goal, rg, rs, np.zeros(5), ratial, np.random.default_rng(sd), "RRT")
    P = O.problem_RRTstar_CFS(r["route"])
    s = P.sys_info
    s1 = copy.copy(s); s1.MAX_O_ITER = 1
    w1 = O.optimizer(P.ROBOT, s1, P.obs, "CFS")
    print("seed", sd, "after iteration 1: status", w1.status, "iter", w1.iter_O)
    H, nj = s.H, 5; nn = H * nj
    x_ = w1.x_; u = w1.u
    s2 = copy.copy(s)
    A, rhs, dist, lid, grad = O.get_con(P.ROBOT, s2, P.obs, x_, u, mode="CFS")
    dist = np.asarray(dist)
    print("  min dist of the iterate per obstacle", dist.min(axis=1), "waypoints below margin", (dist < 0.2).sum(axis=1), "near-zero-surrogate (negative) entries", (dist < 0).sum())
    c = np.zeros(nn + 1); c[-1] = 1
    per = 11; col = np.arange(0, A.shape[0], per)
    iscol = np.zeros(A.shape[0], bool); iscol[col] = True
    Aub = np.hstack([A, -iscol[:, None].astype(float)])
    res = linprog(c, A_ub=Aub, b_ub=rhs, bounds=[(-m, m) for m in s.MAX_input] + [(0, None)], method="highs")
    print("  phase-1 LP: least max violation of the collision rows t* = %.4e (%s)" % (res.fun, "INFEASIBLE" if res.fun > 1e-9 else "feasible"))
    y = -res.ineqlin.marginals[col]
    nz = np.nonzero(y > 1e-6)[0]
    print("  rows carrying weight (obstacle, waypoint, weight, dist):", [(int(e // H), int(e % H), round(float(y[e]), 3), round(float(dist.ravel()[e]), 4)) for e in nz])
    # is the linearisation point itself feasible w.r.t. velocity / input limits?
    X = x_.reshape(H, 10)
    print("  max |omega| %.3f  max |u|/MAX %.3f" % (np.abs(X[:, 5:]).max(), (np.abs(u) / s.MAX_input).max()))
