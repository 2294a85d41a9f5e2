"""Host-side problem builder: the ``sys_info`` struct the reference's drivers assemble before
calling the solver classes (main_FANUC.m:64-127, main_2L.m:69-121, RRTstar_CFS.m:124-187).

Runs once per problem family on the host, exactly as in the reference.  The dynamics are the
robot's double integrator (robotproperty2.m:136-139), so ``Aaug`` / ``Baug`` are written in closed
form: ``A^i = [I i*dt*I; 0 I]`` and ``A^(i-j) B = [(i-j+1/2) dt^2 I; dt I]``.

Field names follow the reference struct so that code written against it reads the same:
``H nstate njoint nu x_ xR Aaug Baug QQ ff Qaug paug caug lim MAX_input epsilon_O MAX_O_ITER alpha robot``.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

# literals of main_FANUC.m:66-77,90-94
FANUC_Qp = np.diag([10.0, 10.0, 1.0, 1.0, 1.0])
FANUC_Rblk = np.array([[10.0, 0, 0, 0, 0], [0, 10, 1, 0, 0], [0, 1, 2, 0, 0], [0, 0, 0, 2, 0], [0, 0, 0, 0, 1]])
RRT_Qv = np.diag([100.0, 20.0, 1.0, 1.0, 1.0])  # RRTstar_CFS.m:133-137


def double_integrator(H: int, nj: int, dt: float):
    """(Aaug, Baug) of main_FANUC.m:79-86 for A=[I dt I;0 I], B=[dt^2/2 I; dt I]."""
    ns = 2 * nj
    I = np.eye(nj)
    Aaug = np.zeros((H * ns, ns))
    Baug = np.zeros((H * ns, H * nj))
    for i in range(H):
        Aaug[i * ns : i * ns + nj, :nj] = I
        Aaug[i * ns : i * ns + nj, nj:] = (i + 1) * dt * I
        Aaug[i * ns + nj : (i + 1) * ns, nj:] = I
        for j in range(i + 1):
            Baug[i * ns : i * ns + nj, j * nj : (j + 1) * nj] = ((i - j) + 0.5) * dt * dt * I
            Baug[i * ns + nj : (i + 1) * ns, j * nj : (j + 1) * nj] = dt * I
    return Aaug, Baug


def cost_terms(Aaug, Baug, Qaug, xR1, xg, H, nj):
    """ff, caug of main_FANUC.m:98-103 for a start state xR1 and goal angles xg."""
    gaug = np.tile(np.concatenate([np.asarray(xg, float), np.zeros(nj)]), H)
    e = Aaug @ xR1 - gaug
    return Baug.T @ (Qaug @ e), float(e @ Qaug @ e)


def build_sys_info(robot, njoint, horizon, x0, xg, x_init, *, Qp, Qv, Rblk, cR, lim, max_input_blk,
                   epsilon_O, MAX_O_ITER) -> SimpleNamespace:
    nj, H = int(njoint), int(horizon)
    ns, nu = 2 * nj, nj
    dt = float(robot.delta_t)
    Aaug, Baug = double_integrator(H, nj, dt)
    Q = np.block([[np.asarray(Qp, float), 0.1 * np.eye(nj)], [0.1 * np.eye(nj), np.asarray(Qv, float)]])
    Qaug = np.kron(np.eye(H), Q * 0.1)
    Qaug[-ns:, -ns:] = Q * 10000  # terminal weight, main_FANUC.m:82-84
    R = np.kron(np.eye(H), np.asarray(Rblk, float))
    R = R + R.T  # :96
    QQ = Baug.T @ Qaug @ Baug + R * cR  # :97
    xR1 = np.concatenate([np.asarray(x0, float), np.zeros(nj)])
    ff, caug = cost_terms(Aaug, Baug, Qaug, xR1, xg, H, nj)
    s = SimpleNamespace()
    s.robot, s.H, s.nstate, s.njoint, s.nu = robot, H, ns, nj, nu
    s.Aaug, s.Baug, s.Qaug_state = Aaug, Baug, Qaug
    s.QQ, s.ff, s.caug = QQ, ff, caug
    s.Qaug, s.paug = QQ, ff  # the drivers alias them (main_FANUC.m:110-111)
    s.xR = xR1.reshape(ns, 1)
    s.x_ = np.asarray(x_init, float).reshape(-1).copy()
    s.alpha = 1.0 / float(np.linalg.svd(QQ, compute_uv=False).max())  # :120
    s.lim = np.asarray(lim, float).reshape(-1)
    s.epsilon_O, s.MAX_O_ITER = float(epsilon_O), int(MAX_O_ITER)
    s.MAX_input = np.tile(np.asarray(max_input_blk, float).reshape(-1), H)  # :127
    # the weights the matrices above were assembled from: what cfs_problem_create_from_weights takes instead of QQ / Qaug
    s.weights = dict(Qp=np.asarray(Qp, float), Qv=np.asarray(Qv, float), q_cross=0.1, w_stage=0.1, w_terminal=10000.0,
                     Rblk=np.asarray(Rblk, float), cR=float(cR))
    return s


def line_reference(x0, xg, horizon):
    """Straight line in joint space with zero velocities; waypoint 0 dropped (main_FANUC.m:38-49)."""
    x0, xg = np.asarray(x0, float), np.asarray(xg, float)
    th = np.stack([np.linspace(x0[c], xg[c], horizon + 1)[1:] for c in range(x0.size)], axis=1)
    return np.concatenate([th, np.zeros_like(th)], axis=1).reshape(-1)


def cubic_resample(route, dt, horizon):
    """Resampling of an RRT route to horizon+1 samples with zero waypoint velocities, as
    ``cubicpolytraj(route, wpTimes, trajTimes)`` does by default (RRTstar_CFS.m:94-100)."""
    route = np.asarray(route, float)
    nwp = route.shape[1]
    wp_t = np.arange(nwp) * dt
    tr_t = np.linspace(0.0, wp_t[-1], horizon + 1)
    out = np.empty((route.shape[0], horizon + 1))
    for n, t in enumerate(tr_t):
        k = min(max(int(np.searchsorted(wp_t, t, side="right")) - 1, 0), nwp - 2)
        tau = (t - wp_t[k]) / (wp_t[k + 1] - wp_t[k])
        out[:, n] = route[:, k] + (3 * tau * tau - 2 * tau**3) * (route[:, k + 1] - route[:, k])
    return out


def cylinder(p1_mm, p2_mm, D, epsilon):
    """An obstacle as the drivers write it (main_FANUC.m:56-60): obs{j}.l (3x2, metres), .D, .epsilon."""
    l = np.stack([np.asarray(p1_mm, float), np.asarray(p2_mm, float)], axis=1) / 1000
    return dict(shape="cylinder", l=l, D=float(D), epsilon=float(epsilon))


# ---- the three demo problems as checked in -----------------------------------------------------
def main_FANUC_problem():
    """main_FANUC.m:13-127 (M200i, H=30, one line obstacle)."""
    from .robotproperty2 import robotproperty2
    robot = robotproperty2("M200i")
    x0 = np.array([0.7825, 0.0284, 0.2172, 0.1444, -1.1779])
    xg = np.array([-0.7825, 0.0284, 0.2172, 0.1444, -1.1779])
    H = 30
    s = build_sys_info(robot, 5, H, x0, xg, line_reference(x0, xg, H), Qp=FANUC_Qp, Qv=FANUC_Qp, Rblk=FANUC_Rblk,
                       cR=50.0, lim=np.ones(5), max_input_blk=np.array([1, 1, np.pi, np.pi, np.pi]) * robot.delta_t,
                       epsilon_O=1e-1, MAX_O_ITER=20)
    obs = [cylinder((3806, 8413, 1), (3606, 8413, 1038), 0.2, 0.25)]
    return "M200i", s, obs


def main_2L_problem(lim=(0.1, 0.2)):
    """main_2L.m:13-121 (two-link arm, H=40, point obstacle, stationary initial trajectory)."""
    from .robotproperty2 import robotproperty2
    robot = robotproperty2("2L")
    x0, xg, H = np.zeros(2), np.array([np.pi / 2, 0.0]), 40
    x_init = np.tile(np.concatenate([x0, np.zeros(2)]), H)  # main_2L.m:50
    s = build_sys_info(robot, 2, H, x0, xg, x_init, Qp=np.diag([10.0, 1.0]), Qv=np.diag([10.0, 1.0]),
                       Rblk=np.diag([5.0, 4.0]), cR=0.1, lim=np.asarray(lim, float),
                       max_input_blk=np.ones(2) * 0.5 * robot.delta_t, epsilon_O=1e-6, MAX_O_ITER=100)
    c = np.array([0.3, 0.3, 0.0])
    obs = [dict(shape="circle", l=np.stack([c, c], axis=1), D=0.05, epsilon=0.05)]
    return "2L", s, obs


def RRTstar_CFS_problem(route_wp):
    """The CFS stage of RRTstar_CFS.m:94-187 for a given RRT route (5 x nwp)."""
    from .robotproperty2 import robotproperty2
    robot = robotproperty2("M200i")
    H, dt = 40, robot.delta_t
    sampled = cubic_resample(route_wp, dt, H)
    x0, xg = sampled[:, 0], sampled[:, -1]
    x_init = np.concatenate([sampled[:, 1:].T, np.zeros((H, 5))], axis=1).reshape(-1)
    s = build_sys_info(robot, 5, H, x0, xg, x_init, Qp=FANUC_Qp, Qv=RRT_Qv, Rblk=FANUC_Rblk, cR=10.0,
                       lim=np.ones(5), max_input_blk=np.array([1, 1, np.pi, np.pi, np.pi]) * dt,
                       epsilon_O=1e-1, MAX_O_ITER=20)
    obs = [cylinder((3606, 8413, 1), (3606, 8413, 1038), 0.2, 0.2),
           cylinder((3406, 7813, 800), (3406, 7813, 1538), 0.2, 0.2)]
    return "M200i", s, obs
