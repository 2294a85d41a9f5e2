"""One outer iteration of the FUSED kernel (the kernel bench.py times), piece by piece and as a whole, on every problem of
the config-3 batch (1024 problems, CFS and PSGCFS) and of a config-4-shaped batch (512 problems, H = 40, nn = 200).

One iteration = identical linearisation on both sides, no amplification by the outer loop: what differs is the QP solver.
Checked per problem:
  * the linearisation of the fused kernel (FusedParams::piece = 1 behind cfs_linearize) against the oracle's, and that the
    candidate pruning leaves every distance and every finite difference BIT-identical;
  * the QP of the fused kernel (piece = 2 behind cfs_qp) bit-identical to the u of a whole solve with MAX_O_ITER = 1;
  * the same QP piece fed with the ORACLE's linearisation (so that both solvers see bit-identical data: the two
    linearisations differ by ~1e-10 in the finite differences, which multipliers of 1e5..1e8 turn into 1e-9..1e-8 of u)
    against the oracle's first-iteration u: <= 1e-9 |u| -- where the two fp64 solvers differ by more, an extended
    precision solve on the optimal active set arbitrates and the device must be no farther from it than
    max(1e-9, 4x the oracle's own distance); end to end (own linearisation) the waypoints agree to < 1e-6 rad;
  * a KKT certificate of the DEVICE answer from the device's own multipliers: stationarity, primal and dual feasibility,
    complementarity, each <= 1e-9 relative (the QPs are strictly convex: a KKT point is the unique minimiser).
"""
import copy
import ctypes as C

import numpy as np
import pytest

from motionplanning_5d_m_amd import workloads

from helpers import device_lambda_to_rows, kkt_certificate, truth_on_active_set

pytestmark = pytest.mark.gpu
CHUNK = 64


def _workload(gpu, route_wp, tag):
    if tag == "c3":
        s, bt = workloads.config3(lambda rb, th, ob: gpu.dist_arm(rb, th, ob)[0], B=1024)
    else:
        s, bt = workloads.config4(route_wp, B=512)
    return s, bt


def _dense_qp(s, bt, mode, b, O, margin):
    """(G, g0, A, rhs) of problem b's first QP from the ORACLE's get_con (dense, reference row order + bound rows)."""
    from types import SimpleNamespace
    obs = [dict(l=np.stack([bt.obs[b, j, :3], bt.obs[b, j, 3:]], axis=1), epsilon=margin[j], D=margin[j]) for j in range(bt.nobs)]
    s2 = SimpleNamespace(**vars(s))
    s2.xR1, s2.robot = bt.xR1[b], O.robotproperty2("M200i")
    nn = s.H * 5
    A, rhs, *_ = O.get_con("M200i", s2, obs, bt.x_init[b], np.zeros(nn), mode=mode)
    if mode == "CFS":
        return 0.5 * (s.QQ + s.QQ.T), bt.ff[b], np.vstack([A, np.eye(nn), -np.eye(nn)]), np.concatenate([rhs, s.MAX_input, s.MAX_input])
    u_ = -s.alpha * (bt.ff[b] + 10.0 * bt.noise[b, 0] / 2.0)          # PSGCFS_FANUC.m:109 at u = 0, iter_O = 1
    return np.eye(nn), -u_, A, rhs


@pytest.mark.parametrize("tag,mode", [("c3", "CFS"), ("c3", "PSGCFS"), ("c4", "CFS")])
def test_first_outer_iteration_of_the_fused_kernel(gpu, O, route_wp, tag, mode):
    s, bt = _workload(gpu, route_wp, tag)
    B, H, nj, nn, nobs = bt.x_init.shape[0], s.H, 5, s.H * 5, bt.nobs
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    noise = bt.noise if (mode == "PSGCFS" and bt.noise is not None) else None
    s1 = copy.copy(s)
    s1.MAX_O_ITER = 1
    slv = gpu.CFSBatch(s1, nobs, margin, mode=mode, max_batch=B)

    # ---- linearisation piece of the fused kernel: pruned == unpruned bit for bit, and == oracle --------------------
    dist, lid, grad = slv.linearize(bt.x_init, bt.obs)
    slv.debug_options(no_prune=True)                    # cfs_debug_set_options, per handle
    try:
        dist_u, lid_u, grad_u = slv.linearize(bt.x_init, bt.obs)
    finally:
        slv.debug_options()
    np.testing.assert_array_equal(dist, dist_u)
    np.testing.assert_array_equal(grad, grad_u)
    np.testing.assert_array_equal(lid, lid_u)
    robot = O.robotproperty2("M200i")
    for b in range(0, B, 37):
        for j in range(nobs):
            ol = np.stack([bt.obs[b, j, :3], bt.obs[b, j, 3:]], axis=1)
            for i in (0, H // 3, H - 1):
                th = bt.x_init[b].reshape(H, 10)[i, :5]
                d, l = O.dist_arm(robot, th, ol)
                assert abs(dist[b, j, i] - d) < 1e-14 and lid[b, j, i] == l
                np.testing.assert_allclose(grad[b, j, i], O.num_jac_dist(robot, th, ol), rtol=0, atol=2e-9)

    # ---- QP piece == whole solve with MAX_O_ITER = 1, bit for bit ---------------------------------------------------
    if mode == "CFS":
        lin = bt.ff
    else:
        lin = -s.alpha * (bt.ff + 10.0 * noise[:, 0] / 2.0)
    u_qp, lam, it, st = slv.qp(lin, np.zeros((B, nn)), bt.xR1, dist, grad)
    whole = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs, noise=noise)
    solved = st == 0
    assert np.array_equal(solved, whole.status == 1) and np.array_equal(st == 2, whole.status == 2) and not (st == 3).any()
    np.testing.assert_array_equal(u_qp[solved], whole.u[solved])
    np.testing.assert_array_equal(it, whole.total_iter)

    # ---- against the oracle's first iteration ------------------------------------------------------------------------
    want = O.optimizer_batch(robot, mode, H, 5, bt.x_init, bt.xR1, s.QQ, bt.ff, bt.caug, s.Aaug, s.Baug, s.lim, s.MAX_input, bt.obs,
                             margin, s.epsilon_O, 1, s.alpha, noise=noise, nthreads=0)
    assert np.array_equal(whole.status, want.status) and np.array_equal(whole.iter_O, want.iter_O)
    assert solved.sum() > 0.6 * B
    assert np.abs(whole.x_ - want.x_).max(axis=1)[solved].max() < 1e-6            # rad; the north-star bar is 1e-5
    # the QP alone: both solvers on the oracle's linearisation
    from types import SimpleNamespace
    o_dist, o_grad = np.zeros_like(dist), np.zeros_like(grad)
    for b in range(B):
        obs = [dict(l=np.stack([bt.obs[b, j, :3], bt.obs[b, j, 3:]], axis=1), epsilon=margin[j], D=margin[j]) for j in range(nobs)]
        s2 = SimpleNamespace(**vars(s))
        s2.xR1, s2.robot = bt.xR1[b], robot
        _, _, o_dist[b], _, o_grad[b] = O.get_con("M200i", s2, obs, bt.x_init[b], np.zeros(nn), mode=mode, dense=False)
    assert np.abs(o_dist - dist).max() < 1e-14 and np.abs(o_grad - grad).max() < 2e-9
    u_dev, lam_dev, _, st_dev = slv.qp(lin, np.zeros((B, nn)), bt.xR1, o_dist, o_grad)
    assert np.array_equal(st_dev == 0, solved)
    scale = np.maximum(np.abs(want.u).max(axis=1), 1e-300)
    rel = np.where(solved, np.abs(u_dev - want.u).max(axis=1), 0.0) / np.where(solved, scale, 1.0)
    assert np.median(rel[solved]) < 1e-9
    arbitrate = np.nonzero(solved & (rel > 1e-9))[0]
    assert arbitrate.size < 0.15 * B
    worst = 0.0
    for b in arbitrate:
        G, g0, A, rhs = _dense_qp(s, bt, mode, b, O, margin)
        xo, lo, _, sto, _ = O.qp_solve(G, g0, A, rhs)
        assert sto == 0
        xt, lt = truth_on_active_set(G, g0, A, rhs, np.nonzero(lo > 0)[0])
        assert lt.min() > -1e-9 * max(lt.max(), 1.0) and (rhs - A @ xt).min() > -1e-9       # the oracle's active set is optimal
        sc = np.abs(xt).max()
        e_dev, e_orc = np.abs(u_dev[b] - xt).max() / sc, np.abs(xo - xt).max() / sc
        worst = max(worst, e_dev)
        assert e_dev <= max(1e-9, 4.0 * e_orc), (b, e_dev, e_orc)
    print(f"[{tag} {mode}] solved {solved.sum()}/{B}; rel u err vs oracle: median {np.median(rel[solved]):.1e} max {rel[solved].max():.1e}; "
          f"{arbitrate.size} arbitrated in extended precision, worst device error {worst:.1e}")

    # ---- KKT certificate of the device answer from the device's multipliers -------------------------------------------
    Gm = 0.5 * (s.QQ + s.QQ.T) if mode == "CFS" else np.eye(nn)
    g0 = bt.ff if mode == "CFS" else -lin
    cert = np.zeros((4, B))
    for c0 in range(0, B, CHUNK):
        sl_ = slice(c0, min(B, c0 + CHUNK))
        n = sl_.stop - sl_.start
        A, rhs = slv.get_con(bt.x_init[sl_], np.zeros((n, nn)), bt.xR1[sl_], bt.obs[sl_])
        if mode == "CFS":
            eye = np.broadcast_to(np.eye(nn), (n, nn, nn))
            A = np.concatenate([A, eye, -eye], axis=1)
            rhs = np.concatenate([rhs, np.broadcast_to(s.MAX_input, (n, nn)), np.broadcast_to(s.MAX_input, (n, nn))], axis=1)
        lam_rows = device_lambda_to_rows(lam[sl_], nobs, H, nj, mode == "CFS")
        cert[:, sl_] = kkt_certificate(Gm, g0[sl_], A, rhs, u_qp[sl_], lam_rows)
    stat, prim, dual, comp = cert[:, solved]
    print(f"[{tag} {mode}] KKT of the device answers: stationarity {stat.max():.1e} primal {prim.max():.1e} dual {dual.max():.1e} "
          f"complementarity {comp.max():.1e}")
    assert stat.max() <= 1e-9 and prim.max() <= 1e-9 and dual.max() <= 1e-9 and comp.max() <= 1e-9
    slv.close()
