"""Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``motionplanning_5d_m_amd`` never does.

It holds
  * a ctypes binding of ``oracle/cfs_oracle.c`` (the literal C restatement of the hot path), and
  * literal numpy restatements of the host-side pieces the reference runs once per solve:
    ``robotproperty2`` (Lib/functions/robotproperty2.m:1-153) and the cost/dynamics assembly of
    the drivers (main_FANUC.m:64-127, main_2L.m:69-121, RRTstar_CFS.m:124-187), plus the three
    deterministic demo problems those drivers define.

PARITY STATUS: forward kinematics and the rollout are pinned by two outputs of the reference's own MATLAB runs found among
its data files (figure/M16iBCapsules.mat:RoCap, data/good_xori.mat + data/M16_ref_2.mat; tests/test_oracle_golden.py); for
everything else "parity unpinned" (see the header of cfs_oracle.c and DESIGN.md section 2): the reference ships no tests or
golden vectors and its QP solver (MathWorks quadprog) is closed source.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcfs_oracle.so")
_MAXLINK = 8

ROBOT_KIND = {"M16iB": 0, "M200i": 1, "2L": 2}
STATUS = {0: "OK_CONVERGED", 1: "OK_MAXITER", 2: "QP_INFEASIBLE", 3: "NUMERIC"}


class _Robot(C.Structure):
    _fields_ = [
        ("kind", C.c_int),
        ("nlink", C.c_int),
        ("DH", C.c_double * (_MAXLINK * 4)),
        ("base", C.c_double * 3),
        ("cap", C.c_double * (_MAXLINK * 6)),
        ("T", C.c_double * 9),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/cfs_oracle.c with gcc (building the checker is not using it)."""
    srcs = [os.path.join(_HERE, f) for f in ("cfs_oracle.c", "mesh_oracle.c", "chomp_oracle.c")]
    if force or (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libcfs_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_dist_lin_seg.restype = C.c_double
        _lib.orc_dist_arm.restype = C.c_double
        _lib.orc_get_cost.restype = C.c_double
        _lib.orc_mesh_seg_distance.restype = C.c_double
        _lib.orc_derivest.restype = C.c_double
        _lib.orc_chomp_fobs.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ----------------------------------------------------------------------------------------------
# robotproperty2 (Lib/functions/robotproperty2.m:1-153), literal constants
# ----------------------------------------------------------------------------------------------
def robotproperty2(rid: str) -> SimpleNamespace:
    r = SimpleNamespace(name=rid)
    if rid == "M200i":  # :12-55
        r.nlink = 6
        r.delta_t = 0.5
        r.DH = np.array(
            [
                [0, 0, 0.050, -1.5708],
                [-1.5708, 0, 0.440, 3.1416],
                [0, 0, 0.035, -1.5708],
                [0, -0.420, 0, 1.5708],
                [0, 0, 0, -1.5708],
                [0, -0.080, 0, 3.1416],
            ],
            dtype=np.float64,
        )
        r.cap = [
            np.array([[0, 0], [0, 0], [0, 0]], dtype=np.float64),
            np.array([[-0.4, 0], [0, 0], [0, 0]], dtype=np.float64),
            np.array([[-0.03, -0.03], [0, 0], [0.05, 0.05]], dtype=np.float64),
            np.array([[0, 0], [0, 0.4], [0, 0]], dtype=np.float64),
            np.array([[0, 0], [0, 0], [-0.26, 0.01]], dtype=np.float64),
            np.array([[0.05, 0.18], [0, 0], [0.1107, 0.1107]], dtype=np.float64),
        ]
        r.cap_r = [0, 0.13, 0, 0.068, 0.01, 0.06]
        r.base = np.array([3150, 8500, 330], dtype=np.float64) / 1000  # :53-54
        r.T = np.zeros((3, 3))
    elif rid == "M16iB":  # :58-99
        r.nlink = 6
        r.delta_t = 0.5
        r.DH = np.array(
            [
                [0.5, 0.65, 0.15, 1.5708],
                [1.5708, 0, 0.77, 0],
                [0, 0, 0.1, 1.5708],
                [0, 0.74, 0, -1.5708],
                [-np.pi / 2, 0, 0, 1.5708],
                [np.pi, 0.1, 0, 0],
            ],
            dtype=np.float64,
        )
        r.cap = [
            np.array([[0, 0], [0, 0], [-0.1, 0.1]], dtype=np.float64),
            np.array([[-0.75, 0], [0, 0], [-0.15, -0.15]], dtype=np.float64),
            np.array([[-0.03, -0.03], [0, 0], [0.05, 0.05]], dtype=np.float64),
            np.array([[0, 0], [0, 0.55], [0, 0]], dtype=np.float64),
            np.array([[0, 0], [0, 0], [-0.05, 0.110]], dtype=np.float64),
            np.array([[-0.11, -0.11], [0, 0], [0.09, 0.09]], dtype=np.float64),
        ]
        r.cap_r = [0.15, 0.13, 0.22, 0.11, 0.07, 0.11]
        r.base = np.array([3250, 8500, 0], dtype=np.float64) / 1000  # :97-98
        r.T = np.zeros((3, 3))
    elif rid == "2L":  # :102-130
        r.nlink = 3
        r.delta_t = 0.5
        r.DH = np.array([[0, 0, 0.3, 0], [0, 0, 0.2, 0], [0, 0, 0, 0]], dtype=np.float64)
        r.T = np.array([[0, 0, 0.3], [0, 0, 0], [0, 0, 0.0]], dtype=np.float64)
        r.cap = [
            np.array([[0, 0.3], [0, 0], [0, 0]], dtype=np.float64),
            np.array([[0, 0.2], [0, 0], [0, 0]], dtype=np.float64),
        ]
        r.cap_r = [0.05, 0.05]
        r.base = np.array([0, 0, 0], dtype=np.float64) / 1000
    else:
        raise ValueError(rid)
    n, dt = r.nlink, r.delta_t
    r.A = np.block([[np.eye(n), dt * np.eye(n)], [np.zeros((n, n)), np.eye(n)]])  # :136-137
    r.B = np.vstack([0.5 * dt**2 * np.eye(n), dt * np.eye(n)])  # :138-139
    return r


def c_robot(robot: SimpleNamespace) -> _Robot:
    rb = _Robot()
    rb.kind = ROBOT_KIND[robot.name]
    rb.nlink = robot.nlink
    for i in range(robot.DH.shape[0]):
        for c in range(4):
            rb.DH[i * 4 + c] = float(robot.DH[i, c])
    for r in range(3):
        rb.base[r] = float(robot.base[r])
    for i, cp in enumerate(robot.cap):
        for k in range(2):
            for r in range(3):
                rb.cap[i * 6 + k * 3 + r] = float(cp[r, k])
    Tm = np.asarray(robot.T, dtype=np.float64)
    for c in range(3):
        for r in range(3):
            rb.T[c * 3 + r] = float(Tm[r, c])
    return rb


# ----------------------------------------------------------------------------------------------
# cost / dynamics assembly (main_FANUC.m:64-127; variants main_2L.m:69-121, RRTstar_CFS.m:124-187)
# ----------------------------------------------------------------------------------------------
def build_sys_info(robot, njoint, horizon, x0, xg, x_init, *, Qp, Qv, Rblk, cR, lim, max_input_blk,
                   epsilon_O, MAX_O_ITER) -> SimpleNamespace:
    nstate, nu, H = 2 * njoint, njoint, horizon
    idx = list(range(njoint)) + list(range(robot.nlink, robot.nlink + njoint))  # [1:nj, nlink+1:nlink+nj]
    A10 = robot.A[np.ix_(idx, idx)]
    B10 = robot.B[np.ix_(idx, list(range(nu)))]
    Q = np.zeros((nstate, nstate))
    Q[:njoint, :njoint] = Qp
    Q[:njoint, njoint:] = 0.1 * np.eye(njoint)
    Q[njoint:, :njoint] = 0.1 * np.eye(njoint)
    Q[njoint:, njoint:] = Qv
    Aaug = np.zeros((H * nstate, nstate))
    Baug = np.zeros((H * nstate, H * nu))
    Qaug = np.zeros((H * nstate, H * nstate))
    for i in range(1, H + 1):
        Aaug[(i - 1) * nstate : i * nstate, :] = np.linalg.matrix_power(A10, i)
        Qaug[(i - 1) * nstate : i * nstate, (i - 1) * nstate : i * nstate] = Q * 0.1
        if i == H:
            Qaug[(i - 1) * nstate : i * nstate, (i - 1) * nstate : i * nstate] = Q * 10000
        for j in range(1, i + 1):
            Baug[(i - 1) * nstate : i * nstate, (j - 1) * nu : j * nu] = np.linalg.matrix_power(A10, i - j) @ B10
    R = np.eye(H * nu)
    for i in range(H):
        R[i * nu : (i + 1) * nu, i * nu : (i + 1) * nu] = Rblk
    R = R + R.T
    QQ = Baug.T @ Qaug @ Baug + R * cR
    xR1 = np.concatenate([np.asarray(x0, dtype=np.float64), np.zeros(njoint)])
    gaug = np.kron(np.ones(H), np.concatenate([np.asarray(xg, dtype=np.float64), np.zeros(njoint)]))
    e = Aaug @ xR1 - gaug
    ff = (e @ Qaug @ Baug).copy()
    caug = float(e @ Qaug @ e)
    s = SimpleNamespace()
    s.robot, s.H, s.nstate, s.njoint, s.nu = robot, H, nstate, njoint, nu
    s.Aaug, s.Baug, s.QQ, s.ff, s.Qaug, s.paug, s.caug = Aaug, Baug, QQ, ff, QQ, ff, caug
    s.xR1 = xR1
    s.x_ = np.asarray(x_init, dtype=np.float64).copy()
    s.alpha = 1.0 / np.linalg.svd(QQ, compute_uv=False).max()
    s.lim = np.asarray(lim, dtype=np.float64)
    s.epsilon_O, s.MAX_O_ITER = float(epsilon_O), int(MAX_O_ITER)
    s.MAX_input = np.kron(np.ones(H), np.asarray(max_input_blk, dtype=np.float64))
    return s


def line_reference(x0, xg, horizon):
    """main_FANUC.m:38-49: straight line in joint space, zero velocities, waypoint 0 dropped."""
    x0, xg = np.asarray(x0, float), np.asarray(xg, float)
    nj = x0.size
    th = np.stack([np.linspace(x0[i], xg[i], horizon + 1) for i in range(nj)])  # nj x (H+1)
    full = np.vstack([th, np.zeros((nj, horizon + 1))])  # ns x (H+1)
    return full[:, 1:].T.reshape(-1).copy()  # xori = xref_(nstate+1:end)


FANUC_Qp = np.diag([10.0, 10, 1, 1, 1])
FANUC_Rblk = np.array([[10.0, 0, 0, 0, 0], [0, 10, 1, 0, 0], [0, 1, 2, 0, 0], [0, 0, 0, 2, 0], [0, 0, 0, 0, 1]])


def problem_main_FANUC(nobs_variant: int = 1) -> SimpleNamespace:
    """main_FANUC.m as checked in (1 obstacle); nobs_variant=3 appends the two RRTstar_CFS.m
    obstacles (RRTstar_CFS.m:42,48) for BASELINE config 2's "3 capsule obstacles"."""
    robot = robotproperty2("M200i")
    x0 = np.array([0.7825, 0.0284, 0.2172, 0.1444, -1.1779])
    xg = np.array([-0.7825, 0.0284, 0.2172, 0.1444, -1.1779])
    H = 30
    x_init = line_reference(x0, xg, H)
    s = build_sys_info(robot, 5, H, x0, xg, x_init, Qp=FANUC_Qp, Qv=FANUC_Qp, Rblk=FANUC_Rblk, cR=50.0,
                       lim=np.ones(5), max_input_blk=np.array([1, 1, np.pi, np.pi, np.pi]) * robot.delta_t,
                       epsilon_O=1e-1, MAX_O_ITER=20)
    obs = [dict(l=np.array([[3806, 8413, 1], [3606, 8413, 1038]], float).T / 1000, D=0.2, epsilon=0.25)]
    if nobs_variant == 3:
        obs.append(dict(l=np.array([[3606, 8413, 1], [3606, 8413, 1038]], float).T / 1000, D=0.2, epsilon=0.25))
        obs.append(dict(l=np.array([[3406, 7813, 800], [3406, 7813, 1538]], float).T / 1000, D=0.2, epsilon=0.25))
    return SimpleNamespace(ROBOT="M200i", sys_info=s, obs=obs)


def problem_main_2L(lim=(0.1, 0.2), horizon=40) -> SimpleNamespace:
    """main_2L.m as checked in (stationary start, H=40, point obstacle)."""
    robot = robotproperty2("2L")
    x0 = np.array([0.0, 0.0])
    xg = np.array([np.pi / 2, 0.0])
    H = horizon
    x_init = np.kron(np.ones(H), np.concatenate([x0, [0.0, 0.0]]))  # main_2L.m:50
    s = build_sys_info(robot, 2, H, x0, xg, x_init, Qp=np.diag([10.0, 1]), Qv=np.diag([10.0, 1]),
                       Rblk=np.diag([5.0, 4]), cR=0.1, lim=np.array(lim, float),
                       max_input_blk=np.array([1.0, 1.0]) * 0.5 * robot.delta_t,
                       epsilon_O=1e-6, MAX_O_ITER=100)
    c = np.array([0.3, 0.3, 0.0])
    obs = [dict(l=np.stack([c, c], axis=1), D=0.05, epsilon=0.05)]
    return SimpleNamespace(ROBOT="2L", sys_info=s, obs=obs)


def cubicpolytraj_zero_vel(route, wp_times, traj_times):
    """Restatement of Robotics System Toolbox cubicpolytraj with its default zero waypoint
    velocities (RRTstar_CFS.m:100; closed source, itself unpinned): per segment
    q_k + (3 tau^2 - 2 tau^3)(q_{k+1}-q_k)."""
    route = np.asarray(route, float)
    out = np.zeros((route.shape[0], len(traj_times)))
    for n, t in enumerate(traj_times):
        k = int(np.searchsorted(wp_times, t, side="right") - 1)
        k = min(max(k, 0), len(wp_times) - 2)
        tau = (t - wp_times[k]) / (wp_times[k + 1] - wp_times[k])
        out[:, n] = route[:, k] + (3 * tau**2 - 2 * tau**3) * (route[:, k + 1] - route[:, k])
    return out


def problem_RRTstar_CFS(route_wp) -> SimpleNamespace:
    """RRTstar_CFS.m:94-187 fed with an RRT route (5 x nwp), e.g. data/200i_xori.mat:route_wp."""
    robot = robotproperty2("M200i")
    dt, H = robot.delta_t, 40
    route_wp = np.asarray(route_wp, float)
    wp_times = np.arange(route_wp.shape[1]) * dt
    traj_times = np.linspace(0, wp_times[-1], H + 1)
    sampled = cubicpolytraj_zero_vel(route_wp, wp_times, traj_times)
    x0, xg = sampled[:, 0], sampled[:, -1]
    full = np.vstack([sampled, np.zeros((5, H + 1))])
    x_init = full[:, 1:].T.reshape(-1).copy()
    s = build_sys_info(robot, 5, H, x0, xg, x_init, Qp=FANUC_Qp, Qv=np.diag([100.0, 20, 1, 1, 1]),
                       Rblk=FANUC_Rblk, cR=10.0, lim=np.ones(5),
                       max_input_blk=np.array([1, 1, np.pi, np.pi, np.pi]) * dt, epsilon_O=1e-1, MAX_O_ITER=20)
    obs = [
        dict(l=np.array([[3606, 8413, 1], [3606, 8413, 1038]], float).T / 1000, D=0.2, epsilon=0.2),
        dict(l=np.array([[3406, 7813, 800], [3406, 7813, 1538]], float).T / 1000, D=0.2, epsilon=0.2),
    ]
    return SimpleNamespace(ROBOT="M200i", sys_info=s, obs=obs)


# ----------------------------------------------------------------------------------------------
# thin wrappers over the C restatement
# ----------------------------------------------------------------------------------------------
def obs_array(obs) -> np.ndarray:
    """list of obs dicts -> (nobs, 6) rows [l(:,1); l(:,2)]."""
    return _f(np.stack([np.concatenate([o["l"][:, 0], o["l"][:, 1]]) for o in obs]))


def dist_lin_seg(p1s, p1e, p2s, p2e):
    pts = np.zeros(6)
    d = lib().orc_dist_lin_seg(_p(_f(p1s)), _p(_f(p1e)), _p(_f(p2s)), _p(_f(p2e)), _p(pts))
    return d, pts


def arm_pos(robot, theta):
    rb = c_robot(robot)
    theta = _f(theta)
    pos = np.zeros(theta.size * 6)
    lib().orc_arm_pos(C.byref(rb), _p(theta), C.c_int(theta.size), _p(pos))
    return pos.reshape(theta.size, 2, 3)  # [link][endpoint k][xyz]


def dist_arm(robot, theta, obs_l):
    rb = c_robot(robot)
    theta = _f(theta)
    o = _f(np.concatenate([obs_l[:, 0], obs_l[:, 1]]))
    lid = C.c_int(0)
    d = lib().orc_dist_arm(C.byref(rb), _p(theta), C.c_int(theta.size), _p(o), C.byref(lid))
    return d, lid.value


def num_jac_dist(robot, theta, obs_l):
    rb = c_robot(robot)
    theta = _f(theta)
    o = _f(np.concatenate([obs_l[:, 0], obs_l[:, 1]]))
    g = np.zeros(theta.size)
    lib().orc_num_jac_dist(C.byref(rb), _p(theta), C.c_int(theta.size), _p(o), _p(g))
    return g


def get_con(ROBOT, sys_info, obs, x_, u, mode="CFS", dense=True):
    """Returns (Ainq, binq, dist, linkid, grad); Ainq (rows, nn) in the reference's row order."""
    s = sys_info
    rb = c_robot(s.robot)
    H, nj, ns = s.H, s.njoint, s.nstate
    nn, nobs = H * nj, len(obs)
    rows = nobs * H * (1 + 2 * nj)
    margin = _f([o["epsilon"] if mode == "CFS" else o["D"] for o in obs])
    Ainq = np.zeros((rows, nn), order="F") if dense else None
    binq = np.zeros(rows)
    dist = np.zeros(nobs * H)
    lid = np.zeros(nobs * H, dtype=np.int32)
    grad = np.zeros(nobs * H * nj)
    Baug = np.asfortranarray(s.Baug)
    Aaug = np.asfortranarray(s.Aaug)
    lib().orc_get_con(C.byref(rb), C.c_int(H), C.c_int(nj), C.c_int(ns), _p(_f(x_)), _p(_f(u)), _p(Baug), _p(Aaug),
                      _p(_f(s.xR1)), _p(_f(s.lim)), C.c_int(nobs), _p(obs_array(obs)), _p(margin),
                      _p(Ainq), _p(binq), _p(dist), _p(lid), _p(grad))
    return Ainq, binq, dist.reshape(nobs, H), lid.reshape(nobs, H), grad.reshape(nobs, H, nj)


def qp_solve(G, g0, A, b):
    """min 1/2 x'Gx+g0'x s.t. Ax<=b. Returns (x, lambda, iters, status, kkt[4])."""
    G = np.asfortranarray(G, dtype=np.float64)
    A = np.asfortranarray(A, dtype=np.float64)
    g0, b = _f(g0), _f(b)
    n, m = G.shape[0], A.shape[0]
    x, lam = np.zeros(n), np.zeros(m)
    it = C.c_int(0)
    st = lib().orc_qp_solve(C.c_int(n), _p(G), _p(g0), C.c_int(m), _p(A), _p(b), _p(x), _p(lam), C.byref(it))
    kkt = np.zeros(4)
    if st == 0:
        lib().orc_qp_kkt(C.c_int(n), _p(G), _p(g0), C.c_int(m), _p(A), _p(b), _p(x), _p(lam), _p(kkt))
    return x, lam, it.value, st, kkt


def rollout(H, nj, dt, xR1, u):
    x_ = np.zeros(H * 2 * nj)
    lib().orc_rollout(C.c_int(H), C.c_int(nj), C.c_double(dt), _p(_f(xR1)), _p(_f(u)), _p(x_))
    return x_


def optimizer(ROBOT, sys_info, obs, mode="CFS", noise=None, history=False):
    """Literal CFS_FANUC(obs,sys_info,ROBOT).optimizer() / PSGCFS_FANUC(...).optimizer()."""
    s = sys_info
    rb = c_robot(s.robot)
    H, nj = s.H, s.njoint
    nn, nx, K = H * nj, H * 2 * nj, s.MAX_O_ITER
    m = 0 if mode == "CFS" else 1
    margin = _f([o["epsilon"] if m == 0 else o["D"] for o in obs])
    u, x_ = np.zeros(nn), np.zeros(nx)
    cost_all, e_cost_all, e_u_all = np.zeros(K), np.zeros(K), np.zeros(K)
    hist_u = np.zeros((K, nn)) if history else None
    hist_x = np.zeros((K, nx)) if history else None
    kkt = np.zeros(4)
    it_o, tot = C.c_int(0), C.c_int(0)
    nz = _f(noise) if noise is not None else None
    st = lib().orc_optimizer(
        C.byref(rb), C.c_int(m), C.c_int(H), C.c_int(nj), C.c_double(s.robot.delta_t), _p(_f(s.x_)), _p(_f(s.xR1)),
        _p(np.asfortranarray(s.QQ)), _p(_f(s.ff)), C.c_double(s.caug), _p(np.asfortranarray(s.Aaug)),
        _p(np.asfortranarray(s.Baug)), _p(_f(s.lim)), _p(_f(s.MAX_input)), C.c_int(len(obs)), _p(obs_array(obs)),
        _p(margin), C.c_double(s.epsilon_O), C.c_int(K), C.c_double(s.alpha), _p(nz),
        C.c_int(0 if nz is None else nz.shape[0]), _p(u), _p(x_), _p(cost_all), _p(e_cost_all), _p(e_u_all),
        C.byref(it_o), C.byref(tot), _p(hist_u), _p(hist_x), _p(kkt))
    n_it = it_o.value - 1
    out = SimpleNamespace(u=u, x_=x_, iter_O=it_o.value, total_iter=tot.value, status=st,
                          cost_all=cost_all[:n_it], e_cost_all=e_cost_all[:n_it], e_u_all=e_u_all[:n_it], kkt=kkt)
    if history:
        out.hist_u, out.hist_x = hist_u[:n_it], hist_x[:n_it]
    return out


def optimizer_batch(robot, mode, H, nj, x_init, xR1, QQ, ff, caug, Aaug, Baug, lim, max_input, obs, margin,
                    epsilon_O, max_o_iter, alpha, noise=None, nthreads=0):
    """B independent problems (shared robot/QQ/dynamics).  obs: (B, nobs, 6); noise: (B, rows, nn)."""
    rb = c_robot(robot)
    x_init, xR1, ff, caug, obs = _f(x_init), _f(xR1), _f(ff), _f(caug), _f(obs)
    B, nobs = x_init.shape[0], obs.shape[1]
    nn, nx, K = H * nj, H * 2 * nj, max_o_iter
    u, x_ = np.zeros((B, nn)), np.zeros((B, nx))
    cost_all, e_cost_all, e_u_all = np.zeros((B, K)), np.zeros((B, K)), np.zeros((B, K))
    it_o, tot, st = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
    nz = _f(noise) if noise is not None else None
    lib().orc_optimizer_batch(
        C.byref(rb), C.c_int(0 if mode == "CFS" else 1), C.c_int(B), C.c_int(H), C.c_int(nj),
        C.c_double(robot.delta_t), _p(x_init), _p(xR1), _p(np.asfortranarray(QQ)), _p(ff), _p(caug),
        _p(np.asfortranarray(Aaug)), _p(np.asfortranarray(Baug)), _p(_f(lim)), _p(_f(max_input)), C.c_int(nobs),
        _p(obs), _p(_f(margin)), C.c_double(epsilon_O), C.c_int(K), C.c_double(alpha), _p(nz),
        C.c_int(0 if nz is None else nz.shape[1]), _p(u), _p(x_), _p(cost_all), _p(e_cost_all), _p(e_u_all),
        _p(it_o), _p(tot), _p(st), C.c_int(nthreads))
    return SimpleNamespace(u=u, x_=x_, cost_all=cost_all, e_cost_all=e_cost_all, e_u_all=e_u_all, iter_O=it_o,
                           total_iter=tot, status=st)


def max_threads() -> int:
    return int(lib().orc_max_threads())


# ----------------------------------------------------------------------------------------------
# mesh obstacles (row f3; mesh_oracle.c): brute force over every triangle, no hierarchy
# ----------------------------------------------------------------------------------------------
def mesh_register(mesh_id, tri):
    """Register a (nt, 3, 3) triangle soup under id 0..15; returns the 3x2 `l` that flags the obstacle as a mesh
    in every obs{j}.l argument of this module (l(:,1) = [NaN; id; 0], l(:,2) = 0)."""
    tri = _f(np.asarray(tri, float).reshape(-1, 9))
    if lib().orc_mesh_register(int(mesh_id), tri.shape[0], _p(tri)) != 0:
        raise ValueError("mesh id outside 0..15")
    return np.array([[np.nan, 0.0], [float(mesh_id), 0.0], [0.0, 0.0]])


def mesh_seg_distance(mesh_id, segs):
    """point2surface_dis for (n, 6) segments: dis (n), points (n, 6), triangle index (n)."""
    segs = _f(np.atleast_2d(segs))
    n = segs.shape[0]
    dis, pts, tri = np.zeros(n), np.zeros((n, 6)), np.zeros(n, np.int32)
    lib().orc_mesh_seg_distance_batch(int(mesh_id), n, _p(segs), _p(dis), _p(pts), _p(tri))
    return dis, pts, tri


# ----------------------------------------------------------------------------------------------
# CHOMP_FANUC (row f4; chomp_oracle.c)
# ----------------------------------------------------------------------------------------------
_DERIVEST_FUN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)


def derivest(fun, x0):
    """derivest(fun, x0, 'Vectorized','no') with the suite's defaults (DERIVESTsuite/derivest.m): (der, errest)."""
    tab = (C.c_double * 64)()
    lib().orc_derivest_setup(tab)
    cb = _DERIVEST_FUN(lambda x, _ctx: float(fun(x)))
    err = C.c_double(0)
    return lib().orc_derivest(tab, cb, None, C.c_double(float(x0)), C.byref(err)), err.value


def chomp_dm(robot, theta, obs_l, D):
    """dm_f (CHOMP_FANUC.m:105-126): per-link distance minus D, without the M200i joint offset."""
    rb, th = c_robot(robot), _f(theta)
    o = _f(np.concatenate([np.asarray(obs_l, float)[:, 0], np.asarray(obs_l, float)[:, 1]]))
    d = np.zeros(th.size)
    lib().orc_chomp_dm(C.byref(rb), C.c_int(th.size), _p(th), _p(o), C.c_double(D), _p(d))
    return d


def chomp_optimizer(ROBOT, sys_info, obs, uref):
    """CHOMP_FANUC(obs_, sys_info, uref, ROBOT).optimizer(); obs = list of dict(l, D, epsilon) (the cell without its header)."""
    s = sys_info
    rb = c_robot(s.robot)
    H, nj = s.H, s.njoint
    nn, nx, K = H * nj, H * 2 * nj, s.MAX_O_ITER
    u, x_ = np.zeros(nn), np.zeros(nx)
    cost_all, e_cost_all, e_u_all = np.zeros(K), np.zeros(K), np.zeros(K)
    D, eps = _f([o["D"] for o in obs]), _f([o["epsilon"] for o in obs])
    it = lib().orc_chomp_optimizer(
        C.byref(rb), C.c_int(H), C.c_int(nj), C.c_double(s.robot.delta_t), _p(_f(s.x_)), _p(_f(s.xR1)), _p(_f(uref)),
        _p(np.asfortranarray(s.QQ)), _p(_f(s.ff)), C.c_double(s.caug), _p(np.asfortranarray(s.Baug)), C.c_int(len(obs)),
        _p(obs_array(obs)), _p(D), _p(eps), C.c_double(s.epsilon_O), C.c_int(K), C.c_double(s.alpha), _p(u), _p(x_),
        _p(cost_all), _p(e_cost_all), _p(e_u_all))
    n = it - 1
    return SimpleNamespace(u=u, x_=x_, iter_O=it, cost_all=cost_all[:n], e_cost_all=e_cost_all[:n], e_u_all=e_u_all[:n])
