"""Synthetic problem batches of the BASELINE configs (inputs only -- no solver code here).

``config3`` is the headline workload: M200i, 5 joints,
H=30, cost matrices exactly as main_FANUC.m:64-127, B problems that differ in start, goal and
8 vertical line-segment obstacles, D=0.2, epsilon=0.25, plus explicit PSGCFS noise.

The obstacle rejection test (axis distance < 0.25 m to the start or goal pose) needs the arm
distance function.  It is injected as ``dist_fn(robot, theta (N,5), obs (M,6)) -> (N,M)`` so that
bench.py passes the GPU entry point (``motionplanning_5d_m_amd.dist_arm``) and CPU-only tests pass
the oracle's; this module itself computes no distances.

Random draws, in this order, from ``numpy.random.default_rng(seed)``:
  start  = x0c + U(-0.1,0.1)^(B,5);  goal = mirror_joint1(x0c) + U(-0.1,0.1)^(B,5)
  NC=48 obstacle candidates per problem: radius U(0.35,0.75)^(B,NC), bearing U(0,2pi)^(B,NC),
  top z U(0.6,1.5)^(B,NC); the first `nobs` candidates that pass the rejection test are kept
  noise  = 0.1 * standard_normal((B, 20, nn))
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from .robotproperty2 import robotproperty2
from .sysinfo import FANUC_Qp, FANUC_Rblk, RRT_Qv, build_sys_info, cost_terms

X0C = np.array([0.7825, 0.0284, 0.2172, 0.1444, -1.1779])  # main_FANUC.m:30
NCAND = 48


def _family(H, Qv, cR):
    robot = robotproperty2("M200i")
    xg = X0C * np.array([-1.0, 1, 1, 1, 1])  # main_FANUC.m:31
    x_init = np.zeros(H * 10)
    return build_sys_info(robot, 5, H, X0C, xg, x_init, Qp=FANUC_Qp, Qv=Qv, Rblk=FANUC_Rblk, cR=cR, lim=np.ones(5),
                          max_input_blk=np.array([1, 1, np.pi, np.pi, np.pi]) * robot.delta_t, epsilon_O=1e-1,
                          MAX_O_ITER=20)


def _batch_terms(s, x0, xg):
    """x_init (line reference), xR1, ff, caug for B (start, goal) pairs, vectorised."""
    B, nj, H = x0.shape[0], s.njoint, s.H
    th = np.linspace(x0, xg, H + 1)[1:].transpose(1, 0, 2)  # main_FANUC.m:38-49, waypoint 0 dropped
    x_init = np.concatenate([th, np.zeros_like(th)], axis=2).reshape(B, -1)
    xR1 = np.concatenate([x0, np.zeros((B, nj))], axis=1)
    ff = np.zeros((B, H * nj))
    caug = np.zeros(B)
    for b in range(B):
        ff[b], caug[b] = cost_terms(s.Aaug, s.Baug, s.Qaug_state, xR1[b], xg[b], H, nj)
    return x_init, xR1, ff, caug


def config3(dist_fn, B=1024, nobs=8, seed=20260101, H=30):
    """BASELINE config 3: returns (sys_info family, batch namespace)."""
    s = _family(H, FANUC_Qp, 50.0)
    robot = s.robot
    rng = np.random.default_rng(seed)
    x0 = X0C + rng.uniform(-0.1, 0.1, (B, 5))
    xg = X0C * np.array([-1.0, 1, 1, 1, 1]) + rng.uniform(-0.1, 0.1, (B, 5))
    rad = rng.uniform(0.35, 0.75, (B, NCAND))
    ang = rng.uniform(0.0, 2 * np.pi, (B, NCAND))
    z2 = rng.uniform(0.6, 1.5, (B, NCAND))
    noise = 0.1 * rng.standard_normal((B, 20, H * 5))
    cx = robot.base[0] + rad * np.cos(ang)
    cy = robot.base[1] + rad * np.sin(ang)
    cand = np.stack([cx, cy, np.full_like(cx, 0.001), cx, cy, z2], axis=2)  # (B, NC, 6)
    obs = np.zeros((B, nobs, 6))
    for b in range(B):
        d = np.asarray(dist_fn(robot, np.stack([x0[b], xg[b]]), cand[b]))  # (2, NC)
        ok = np.nonzero((d >= 0.25).all(axis=0))[0]
        if ok.size < nobs:
            raise RuntimeError(f"problem {b}: only {ok.size} of {NCAND} obstacle candidates accepted")
        obs[b] = cand[b, ok[:nobs]]
    x_init, xR1, ff, caug = _batch_terms(s, x0, xg)
    batch = SimpleNamespace(B=B, nobs=nobs, x0=x0, xg=xg, x_init=x_init, xR1=xR1, ff=ff, caug=caug, obs=obs,
                            noise=noise, margin_cfs=np.full(nobs, 0.25), margin_psg=np.full(nobs, 0.2))
    return s, batch


def config4(route_wp, B=4096, seed=20260104, H=40, sigma=0.02):
    """BASELINE config 4 in synthetic form: the reference runs ONE CFS solve on one
    RRT route (RRTstar_CFS.m:94-195); the batch here is B perturbed copies of a feasible RRT route
    (`route_wp`, 5 x nwp, e.g. the matrix of data/200i_xori.mat): every waypoint is jittered by
    N(0, sigma^2) rad, resampled to H+1 points with zero waypoint velocities (RRTstar_CFS.m:96-100), and
    solved with the two obstacles and the cost matrices of RRTstar_CFS.m:40-50,124-187."""
    from .sysinfo import cubic_resample
    s = _family(H, RRT_Qv, 10.0)
    rng = np.random.default_rng(seed)
    route_wp = np.asarray(route_wp, float)
    routes = route_wp[None] + sigma * rng.standard_normal((B,) + route_wp.shape)
    x0, xg = routes[:, :, 0].copy(), routes[:, :, -1].copy()
    one = np.array([[3.606, 8.413, 0.001, 3.606, 8.413, 1.038], [3.406, 7.813, 0.800, 3.406, 7.813, 1.538]])
    obs = np.broadcast_to(one, (B, 2, 6)).copy()
    _, xR1, ff, caug = _batch_terms(s, x0, xg)
    x_init = np.zeros((B, H * 10))
    for b in range(B):
        smp = cubic_resample(routes[b], s.robot.delta_t, H)
        x_init[b] = np.concatenate([smp[:, 1:].T, np.zeros((H, 5))], axis=1).reshape(-1)
    batch = SimpleNamespace(B=B, nobs=2, x0=x0, xg=xg, x_init=x_init, xR1=xR1, ff=ff, caug=caug, obs=obs, noise=None,
                            margin_cfs=np.full(2, 0.2), margin_psg=np.full(2, 0.2))
    return s, batch


def config5(B=256, seed=20260105, H=50, n_tri=10000, margin=0.12, tri=None):
    """BASELINE config 5 (mesh map, 50 waypoints, 256 seeds), synthetic: the reference's assembly-line map is an STL
    file that cannot travel (map/assembly line_Assem1.STL, 27 396 triangles, mm) and its distance function is missing
    from the reference, so the map is ``mesh.assembly_line`` (about n_tri triangles, metres, around the M200i base) and
    the single obstacle is that mesh.  Start / goal draws as config 3.  Returns (family, batch, triangles)."""
    from .mesh import assembly_line
    s = _family(H, FANUC_Qp, 50.0)
    rng = np.random.default_rng(seed)
    x0 = X0C + rng.uniform(-0.1, 0.1, (B, 5))
    xg = X0C * np.array([-1.0, 1, 1, 1, 1]) + rng.uniform(-0.1, 0.1, (B, 5))
    noise = 0.1 * rng.standard_normal((B, 20, H * 5))
    x_init, xR1, ff, caug = _batch_terms(s, x0, xg)
    if tri is None:
        tri = assembly_line(s.robot.base, n_target=n_tri, seed=seed)
    batch = SimpleNamespace(B=B, nobs=1, x0=x0, xg=xg, x_init=x_init, xR1=xR1, ff=ff, caug=caug, obs=np.zeros((B, 1, 6)),
                            noise=noise, margin_cfs=np.full(1, margin), margin_psg=np.full(1, margin))
    return s, batch, tri


def config5_reference_map(B=256, seed=20260105, H=50, fixture=None):
    """BASELINE config 5 on the REFERENCE's own triangles: the cell of map/assembly line_Assem1.STL around robot.base after
    Lib/functions/MapFromSTL.m:6-10 and mm -> m (tests/golden/assembly_line_cell.npz, a data fixture made in the build
    container by tests/golden/make_reference_map.py: 13 258 of the file's 27 396 triangles, every one with a vertex within
    2.5 m of the base).  Same start / goal draws, horizon and noise as `config5`; margins are main_FANUC.m:59-60's own
    (D = 0.2 for PSGCFS_FANUC, epsilon = 0.25 for CFS_FANUC): the joint-space line between start and goal passes the
    assembly line at 0.17-0.21 m, so every problem has active collision rows and (oracle, first 16) all are feasible.  The
    distance function is still the build's own (the reference calls an undefined point2surface_dis): parity unpinned,
    see DESIGN.md section 9."""
    import os
    from .mesh import load_map_fixture
    if fixture is None:
        fixture = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "assembly_line_cell.npz")
    s, batch, tri = config5(B=B, seed=seed, H=H, tri=load_map_fixture(fixture))
    batch.margin_cfs, batch.margin_psg = np.full(1, 0.25), np.full(1, 0.2)
    return s, batch, tri
