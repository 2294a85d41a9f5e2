"""Generates tests/golden/cfs_cases.npz from the CPU oracle (oracle/cfs_oracle.c).

The reference (MATLAB) ships no golden vectors and cannot be run here, so these fixtures are
outputs of the oracle restatement, not of the reference itself ("parity unpinned").
They pin the oracle against regressions and travel to the GPU box, where /root/reference does
not exist.  The only reference-owned datum is route_wp_200i_xori.npy = the 5x16 matrix `route_wp`
of data/200i_xori.mat (a data file, loaded with scipy.io.loadmat and re-saved as .npy).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

out = {}


def case(name, P, mode="CFS", noise=None):
    s = P.sys_info
    A, b, dist, lid, grad = O.get_con(P.ROBOT, s, P.obs, s.x_, np.zeros(s.H * s.nu), mode)
    r = O.optimizer(P.ROBOT, s, P.obs, mode, noise=noise, history=True)
    out[name + "/x_init"] = s.x_
    out[name + "/binq1"] = b
    out[name + "/Ainq1_sum"] = np.array([A.sum(), np.abs(A).sum()])
    out[name + "/dist1"], out[name + "/linkid1"], out[name + "/grad1"] = dist, lid, grad
    out[name + "/u"], out[name + "/x_"] = r.u, r.x_
    out[name + "/cost_all"], out[name + "/e_u_all"], out[name + "/e_cost_all"] = r.cost_all, r.e_u_all, r.e_cost_all
    out[name + "/iter_status"] = np.array([r.iter_O, r.total_iter, r.status])
    out[name + "/hist_u1"] = r.hist_u[0] if len(r.hist_u) else np.zeros(0)
    if noise is not None:
        out[name + "/noise"] = noise


rng = np.random.default_rng(7)
case("main_FANUC_CFS", O.problem_main_FANUC())
case("main_FANUC_PSGCFS", O.problem_main_FANUC(), "PSGCFS", 0.1 * rng.standard_normal((20, 150)))
case("main_2L_CFS", O.problem_main_2L())
case("main_2L_lim1_CFS", O.problem_main_2L(lim=(1, 1)))
rw = np.load(os.path.join(ROOT, "tests", "golden", "route_wp_200i_xori.npy"))
case("RRTstar_CFS", O.problem_RRTstar_CFS(rw))

# forward kinematics / distance samples for the three robot models
for rid, nj in (("M200i", 5), ("M16iB", 5), ("2L", 2)):
    robot = O.robotproperty2(rid)
    th = rng.uniform(-1.5, 1.5, (16, nj))
    obs = np.array([[3.4, 8.2, 0.0, 3.5, 8.3, 1.2], [0.3, 0.3, 0.0, 0.3, 0.3, 0.0], [3.6, 8.4, 0.5, 3.6, 8.4, 0.5]])
    d = np.zeros((16, 3)); lid = np.zeros((16, 3), np.int32); pos = np.zeros((16, nj, 2, 3))
    for n in range(16):
        pos[n] = O.arm_pos(robot, th[n])
        for j in range(3):
            d[n, j], lid[n, j] = O.dist_arm(robot, th[n], np.stack([obs[j, :3], obs[j, 3:]], axis=1))
    out[f"geom_{rid}/theta"], out[f"geom_{rid}/obs"] = th, obs
    out[f"geom_{rid}/d"], out[f"geom_{rid}/linkid"], out[f"geom_{rid}/pos"] = d, lid, pos

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cfs_cases.npz"), **out)
print("wrote", len(out), "arrays")
