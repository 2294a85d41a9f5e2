"""GPU parity: the HIP path (through the C ABI of libcfs_hip.so) against the CPU oracle and the committed
fixtures.  fp64 everywhere; tolerances are stated per test (the north-star bar is l_inf < 1e-5 rad).

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_RAD = 1e-7        # waypoint tolerance asserted on the reference's own demo problems (bar: 1e-5)


# ---- a1-a4: forward kinematics, segment distance, dist_arm, literal num_jac ------------------------
def test_dist_arm_all_models(gpu, O, golden):
    for rid, nj in (("M200i", 5), ("M16iB", 5), ("2L", 2)):
        robot = gpu.robotproperty2(rid)
        th, obs = golden[f"geom_{rid}/theta"], golden[f"geom_{rid}/obs"]
        d, lid, pos = gpu.dist_arm(robot, th, obs, want_pos=True)
        np.testing.assert_allclose(pos, golden[f"geom_{rid}/pos"], rtol=0, atol=1e-14)
        np.testing.assert_allclose(d, golden[f"geom_{rid}/d"], rtol=0, atol=1e-14)
        np.testing.assert_array_equal(lid, golden[f"geom_{rid}/linkid"])


def test_dist_arm_random_and_edge_cases(gpu, O):
    rng = np.random.default_rng(11)
    robot, orobot = gpu.robotproperty2("M200i"), O.robotproperty2("M200i")
    th = rng.uniform(-2, 2, (257, 5))                           # ragged vs the 256-thread block
    obs = np.concatenate([rng.uniform([2.8, 8.0, 0, 2.8, 8.0, 0], [3.6, 9.0, 1.5, 3.6, 9.0, 1.5], (5, 6)),
                          [[3.3, 8.6, 0.4, 3.3, 8.6, 0.4]]])     # + a point obstacle (distLinSeg D2 == 0)
    d, lid = gpu.dist_arm(robot, th, obs)
    for n in range(0, 257, 8):
        for j in range(6):
            dd, ll = O.dist_arm(orobot, th[n], np.stack([obs[j, :3], obs[j, 3:]], axis=1))
            assert abs(d[n, j] - dd) < 1e-13 and lid[n, j] == ll
    # near-zero surrogate branch (dist_arm_2L.m:15-17): obstacle point on link 1
    r2 = gpu.robotproperty2("2L")
    d2, l2 = gpu.dist_arm(r2, np.zeros((1, 2)), np.array([[0.15, 0, 0, 0.15, 0, 0]]))
    assert l2[0, 0] == 1 and abs(d2[0, 0] + 0.15) < 1e-15
    # empty input is a no-op
    d0, _ = gpu.dist_arm(robot, np.zeros((0, 5)), obs)
    assert d0.shape == (0, 6)


@pytest.mark.parametrize("name", ["main_FANUC_CFS", "main_2L_CFS", "RRTstar_CFS"])
def test_linearize_and_dense_get_con(gpu, O, golden, route_wp, name):
    if name == "main_FANUC_CFS":
        (R, s, obs), P = gpu.main_FANUC_problem(), O.problem_main_FANUC()
    elif name == "main_2L_CFS":
        (R, s, obs), P = gpu.main_2L_problem(), O.problem_main_2L()
    else:
        (R, s, obs), P = gpu.RRTstar_CFS_problem(route_wp), O.problem_RRTstar_CFS(route_wp)
    slv = gpu.CFS_FANUC(obs, s, R)
    dist, lid, grad = slv._batch.linearize(s.x_[None], gpu.obs_to_array(obs)[None])
    np.testing.assert_allclose(dist[0], golden[name + "/dist1"], rtol=0, atol=1e-14)
    np.testing.assert_array_equal(lid[0], golden[name + "/linkid1"])
    # central difference with eps = 1e-5 amplifies 1e-16 distance rounding to ~1e-10 (num_jac.m:15)
    np.testing.assert_allclose(grad[0], golden[name + "/grad1"], rtol=0, atol=2e-9)
    slv.get_con()                                               # public self.Ainq / self.binq, reference row order
    A, b, *_ = O.get_con(P.ROBOT, P.sys_info, P.obs, P.sys_info.x_, np.zeros(s.H * s.nu))
    assert slv.Ainq.shape == A.shape
    np.testing.assert_allclose(slv.Ainq, A, rtol=0, atol=5e-9)
    np.testing.assert_allclose(slv.binq, golden[name + "/binq1"], rtol=0, atol=1e-13)
    # at a non-zero linearisation point u (exercises the Diff'*Bj*u term of CFS_FANUC.m:120)
    u = np.sin(np.arange(s.H * s.nu)) * 0.05
    x_ = O.rollout(s.H, s.njoint, s.robot.delta_t, P.sys_info.xR1, u)
    A2, b2, *_ = O.get_con(P.ROBOT, P.sys_info, P.obs, x_, u)
    Ag, bg = slv._batch.get_con(x_[None], u[None], P.sys_info.xR1[None], gpu.obs_to_array(obs)[None])
    np.testing.assert_allclose(Ag[0], A2, rtol=0, atol=5e-9)
    np.testing.assert_allclose(bg[0], b2, rtol=0, atol=5e-9)


# ---- a6/a7: the QP on given linearisation data -----------------------------------------------------
def test_qp_entry_point_cfs_and_psgcfs(gpu, O):
    R, s, obs = gpu.main_FANUC_problem()
    P = O.problem_main_FANUC()
    nn = 150
    A, b, dist, lid, grad = O.get_con(P.ROBOT, P.sys_info, P.obs, s.x_, np.zeros(nn))
    G = np.vstack([A, np.eye(nn), -np.eye(nn)]); h = np.concatenate([b, s.MAX_input, s.MAX_input])
    want, lam_o, _, st, _ = O.qp_solve(s.QQ, s.ff, G, h)
    slv = gpu.CFSBatch(s, 1, [0.25], mode="CFS", max_batch=2)
    xR1 = P.sys_info.xR1
    u, lam, it, stg = slv.qp(np.stack([s.ff, s.ff]), np.zeros((2, nn)), np.stack([xR1, xR1]), np.stack([dist, dist]),
                             np.stack([grad, grad]))
    assert st == 0 and (stg == 0).all()
    np.testing.assert_allclose(u[0], want, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(u[0], u[1])
    # multipliers: collision rows first, ordered (j, i)
    lam_col = lam_o[0:330:11]
    np.testing.assert_allclose(lam[0][:30], lam_col, rtol=1e-6, atol=1e-6)
    # PSGCFS projection: min |v - u_|^2 s.t. Ainq v <= binq, no bounds (PSGCFS_FANUC.m:117-120)
    A2, b2, dist2, _, grad2 = O.get_con(P.ROBOT, P.sys_info, P.obs, s.x_, np.zeros(nn), "PSGCFS")
    u_ = -s.alpha * s.ff
    want2, _, _, st2, _ = O.qp_solve(np.eye(nn), -u_, A2, b2)
    slv2 = gpu.CFSBatch(s, 1, [0.2], mode="PSGCFS", max_batch=1)
    u2, _, _, st3 = slv2.qp(u_[None], np.zeros((1, nn)), xR1[None], dist2[None], grad2[None])
    assert st2 == 0 and st3[0] == 0
    np.testing.assert_allclose(u2[0], want2, rtol=0, atol=1e-12)


# ---- a8/a9: the whole optimizer() on the reference's demo problems -----------------------------------
def _compare(got, g, name, caug):
    it, tot, st = g[name + "/iter_status"]
    assert got.status == st and got.iter_O == it
    assert np.abs(got.x_ - g[name + "/x_"]).max() < TOL_RAD
    assert np.abs(got.u - g[name + "/u"]).max() < TOL_RAD
    np.testing.assert_allclose(got.eval.cost_all, g[name + "/cost_all"], rtol=1e-8)
    np.testing.assert_allclose(got.eval.e_u_all, g[name + "/e_u_all"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(got.eval.e_cost_all, g[name + "/e_cost_all"], rtol=1e-6, atol=1e-4)


def test_main_fanuc_cfs(gpu, golden):
    R, s, obs = gpu.main_FANUC_problem()
    got = gpu.CFS_FANUC(obs, s, R).optimizer()
    _compare(got, golden, "main_FANUC_CFS", s.caug)
    assert got.status == 0 and got.iter_O == 11               # "Converged at step11" 
    x = got.x_.reshape(30, 10)
    np.testing.assert_allclose(x[14, :5], [-0.170879, -0.202055, 1.012415, 0.183991, -0.739707], atol=1e-6)


def test_main_fanuc_psgcfs(gpu, golden):
    R, s, obs = gpu.main_FANUC_problem()
    got = gpu.PSGCFS_FANUC(obs, s, R).optimizer(noise=golden["main_FANUC_PSGCFS/noise"])
    _compare(got, golden, "main_FANUC_PSGCFS", s.caug)
    assert got.status == 1 and got.iter_O == 21               # always MAX_O_ITER iterations


def test_main_2l_reports_infeasible_at_iteration_2(gpu, golden):
    R, s, obs = gpu.main_2L_problem()
    got = gpu.CFS_FANUC(obs, s, R).optimizer()
    assert got.status == 2 and got.iter_O == 2                 # SURVEY N9; the reference would crash at CFS_FANUC.m:92
    assert np.abs(got.u - golden["main_2L_CFS/hist_u1"]).max() < TOL_RAD      # iteration-1 quantities are defined
    assert np.abs(got.x_ - golden["main_2L_CFS/x_"]).max() < TOL_RAD
    np.testing.assert_allclose(got.eval.cost_all, golden["main_2L_CFS/cost_all"], rtol=1e-9)
    R, s, obs = gpu.main_2L_problem(lim=(1, 1))                # the converging variant
    _compare(gpu.CFS_FANUC(obs, s, R).optimizer(), golden, "main_2L_lim1_CFS", s.caug)


def test_rrtstar_cfs_stage(gpu, golden, route_wp):
    R, s, obs = gpu.RRTstar_CFS_problem(route_wp)
    got = gpu.CFS_FANUC(obs, s, R).optimizer()
    _compare(got, golden, "RRTstar_CFS", s.caug)
    assert got.status == 0 and got.iter_O == 18


def test_m16ib_model_and_three_obstacles(gpu, O):
    # M16iB distance function (dist_arm_3D_Heu_2.m) and a 3-obstacle problem (BASELINE config 2 variant)
    R, s, obs = gpu.main_FANUC_problem()
    # (a set on which the oracle itself is insensitive to a 1e-10 perturbation of x_init; other triples
    #  amplify such a perturbation to 4e-4 rad -- see DESIGN.md "Numerical limits")
    obs3 = obs + [gpu.cylinder((2700, 8900, 1), (2700, 8900, 900), 0.2, 0.25), gpu.cylinder((3150, 7800, 1), (3150, 7800, 700), 0.2, 0.25)]
    got = gpu.CFS_FANUC(obs3, s, R).optimizer()
    P = O.problem_main_FANUC()
    want = O.optimizer(P.ROBOT, P.sys_info, [dict(l=o["l"], D=o["D"], epsilon=o["epsilon"]) for o in obs3], "CFS")
    assert got.status == want.status and got.iter_O == want.iter_O and np.abs(got.x_ - want.x_).max() < TOL_RAD
    s16 = copy.copy(s); s16.robot = gpu.robotproperty2("M16iB")
    th0 = np.array([0.5, 1.2, 0.1, 0.0, -1.2]); th1 = np.array([-0.5, 1.2, 0.1, 0.0, -1.2])
    s16 = gpu.build_sys_info(s16.robot, 5, 20, th0, th1, gpu.line_reference(th0, th1, 20), Qp=np.diag([10.0, 10, 1, 1, 1]),
                             Qv=np.diag([10.0, 10, 1, 1, 1]), Rblk=np.eye(5) * 2, cR=50.0, lim=np.ones(5), max_input_blk=np.ones(5),
                             epsilon_O=0.1, MAX_O_ITER=20)
    ob = [gpu.cylinder((4300, 8500, 1), (4300, 8500, 1500), 0.2, 0.3)]
    got = gpu.CFS_FANUC(ob, s16, "M16iB").optimizer()
    Ps = copy.copy(P.sys_info); orb = O.robotproperty2("M16iB")
    t = O.build_sys_info(orb, 5, 20, th0, th1, O.line_reference(th0, th1, 20), Qp=np.diag([10.0, 10, 1, 1, 1]),
                         Qv=np.diag([10.0, 10, 1, 1, 1]), Rblk=np.eye(5) * 2, cR=50.0, lim=np.ones(5), max_input_blk=np.ones(5),
                         epsilon_O=0.1, MAX_O_ITER=20)
    want = O.optimizer("M16iB", t, [dict(l=ob[0]["l"], D=0.2, epsilon=0.3)], "CFS")
    assert got.status == want.status and got.iter_O == want.iter_O and np.abs(got.x_ - want.x_).max() < TOL_RAD


def test_cfs_mode_does_not_depend_on_alpha(gpu):
    """sys_info.alpha is PSGCFS's step (main_FANUC.m:120).  The CFS kernel's early infeasibility bound needs
    lambda_max(QQ); it is bounded rigorously at cfs_problem_create, not read from 1/alpha -- a CFS handle created with
    alpha = 0 or 1.0 must give bit-identical results, infeasibility verdicts included."""
    from motionplanning_5d_m_amd import workloads
    s, bt = workloads.config3(lambda rb, th, ob: gpu.dist_arm(rb, th, ob)[0], B=96)
    ref = None
    for alpha in (s.alpha, 0.0, 1.0):
        s2 = copy.copy(s)
        s2.alpha = alpha
        slv = gpu.CFSBatch(s2, bt.nobs, bt.margin_cfs, mode="CFS", max_batch=96)
        got = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs)
        slv.close()
        if ref is None:
            ref = got
            assert (got.status == 2).sum() >= 10 and (got.status == 0).sum() >= 30
        else:
            np.testing.assert_array_equal(got.status, ref.status)
            np.testing.assert_array_equal(got.x_, ref.x_)
            np.testing.assert_array_equal(got.total_iter, ref.total_iter)


def test_handles_on_every_visible_device(gpu):
    """cfs_set_device + cfs_problem_create on every visible GPU of the process, then a solve on each handle, interleaved: the
    fused kernel's dynamic-LDS attribute is a per-device function attribute (it used to be set once per process)."""
    R, s, obs = gpu.main_FANUC_problem()
    n = gpu.device_count()
    slvs = [gpu.CFSBatch(s, 1, [0.25], mode="CFS", max_batch=1, device=k) for k in range(n)]
    args = (s.x_[None], s.xR[:, 0][None], s.ff[None], np.array([s.caug]), gpu.obs_to_array(obs)[None])
    ref = None
    for rep in range(2):
        for k in range(n):
            got = slvs[k].solve(*args)
            assert got.status[0] == 0 and got.iter_O[0] == 11
            ref = got if ref is None else ref
            np.testing.assert_array_equal(got.x_, ref.x_)
    for sl in slvs:
        sl.close()
    gpu.lib().cfs_set_device(0)


def test_zero_iterations_and_max_iter_edge(gpu):
    R, s, obs = gpu.main_FANUC_problem()
    s0 = copy.copy(s); s0.MAX_O_ITER = 0
    got = gpu.CFS_FANUC(obs, s0, R).optimizer()
    assert got.status == 1 and got.iter_O == 1 and len(got.eval.cost_all) == 0 and np.array_equal(got.x_, s.x_)
    s1 = copy.copy(s); s1.MAX_O_ITER = 1
    got = gpu.CFS_FANUC(obs, s1, R).optimizer()
    assert got.status == 1 and got.iter_O == 2 and len(got.eval.cost_all) == 1


def test_maximum_horizon_and_many_obstacles(gpu, O):
    # H = 64 (CFS_MAX_H), 12 obstacles, M16iB kinematics: the largest shape that fits one CU's LDS next to the solver state
    robot, orobot = gpu.robotproperty2("M16iB"), O.robotproperty2("M16iB")
    th0 = np.array([0.5, 1.2, 0.1, 0.0, -1.2]); th1 = np.array([-0.4, 1.1, 0.2, 0.1, -1.0])
    kw = dict(Qp=np.diag([10.0, 10, 1, 1, 1]), Qv=np.diag([10.0, 10, 1, 1, 1]), Rblk=np.eye(5) * 2, cR=50.0, lim=np.ones(5),
              max_input_blk=np.ones(5), epsilon_O=0.1, MAX_O_ITER=8)
    s = gpu.build_sys_info(robot, 5, 64, th0, th1, gpu.line_reference(th0, th1, 64), **kw)
    t = O.build_sys_info(orobot, 5, 64, th0, th1, O.line_reference(th0, th1, 64), **kw)
    rng = np.random.default_rng(2)
    obs = []
    for k in range(12):
        ang, rad = rng.uniform(0, 2 * np.pi), rng.uniform(0.9, 1.5)      # converges at step 7 with 20 active-set steps
        x, y = 3250 + 1000 * rad * np.cos(ang), 8500 + 1000 * rad * np.sin(ang)
        obs.append(gpu.cylinder((x, y, 1), (x, y, rng.uniform(600, 1500)), 0.2, 0.3))
    got = gpu.CFS_FANUC(obs, s, "M16iB").optimizer()
    want = O.optimizer("M16iB", t, [dict(l=o["l"], D=o["D"], epsilon=o["epsilon"]) for o in obs], "CFS")
    assert got.status == want.status and got.iter_O == want.iter_O
    assert np.abs(got.x_ - want.x_).max() < TOL_RAD and got.x_.shape == (64 * 10,)


def test_single_obstacle_short_horizon_and_ragged_batch(gpu, O):
    # H = 3, one obstacle, batch of 3 (not a multiple of anything), PSGCFS without noise (noise = NULL -> zeros)
    robot, orobot = gpu.robotproperty2("M200i"), O.robotproperty2("M200i")
    kw = dict(Qp=np.diag([10.0, 10, 1, 1, 1]), Qv=np.diag([10.0, 10, 1, 1, 1]), Rblk=np.eye(5) * 2, cR=50.0, lim=np.ones(5),
              max_input_blk=np.ones(5), epsilon_O=0.1, MAX_O_ITER=4)
    x0 = np.array([0.7825, 0.0284, 0.2172, 0.1444, -1.1779])
    starts = [x0, x0 + 0.05, x0 - 0.03]
    goals = [x0 * np.array([0.9, 1, 1, 1, 1]), x0 + 0.1, x0 * 0.95]
    sys_g = [gpu.build_sys_info(robot, 5, 3, a, b, gpu.line_reference(a, b, 3), **kw) for a, b in zip(starts, goals)]
    sys_o = [O.build_sys_info(orobot, 5, 3, a, b, O.line_reference(a, b, 3), **kw) for a, b in zip(starts, goals)]
    ob = [gpu.cylinder((3806, 8413, 1), (3606, 8413, 1038), 0.2, 0.25)]
    slv = gpu.CFSBatch(sys_g[0], 1, [0.2], mode="PSGCFS", max_batch=8)
    got = slv.solve(np.stack([s.x_ for s in sys_g]), np.stack([s.xR[:, 0] for s in sys_g]), np.stack([s.ff for s in sys_g]),
                    np.array([s.caug for s in sys_g]), np.stack([gpu.obs_to_array(ob)] * 3))
    for b in range(3):
        want = O.optimizer("M200i", sys_o[b], [dict(l=ob[0]["l"], D=0.2, epsilon=0.25)], "PSGCFS", noise=None)
        assert got.status[b] == want.status and got.iter_O[b] == want.iter_O
        assert np.abs(got.x_[b] - want.x_).max() < TOL_RAD
        np.testing.assert_allclose(got.cost_all[b, :want.iter_O - 1], want.cost_all, rtol=1e-9)


@pytest.mark.parametrize("nj,H,mode", [(3, 24, "CFS"), (4, 20, "PSGCFS"), (6, 16, "CFS"), (6, 40, "PSGCFS"), (3, 60, "CFS"), (4, 64, "CFS")])
def test_other_joint_counts(gpu, O, nj, H, mode):
    """The kernels are instantiated for 2..6 joints; the drivers only use 5 (and 2).  First nj joints of the M200i
    (all 6 DH rows exist, robotproperty2.m:24-29), one line obstacle across the sweep."""
    robot, orobot = gpu.robotproperty2("M200i"), O.robotproperty2("M200i")
    x0 = np.array([0.7825, 0.0284, 0.2172, 0.1444, -1.1779, 0.3])[:nj]
    xg = x0 * np.array([-1.0, 1, 1, 1, 1, 1])[:nj]
    kw = dict(Qp=np.diag([10.0, 10, 1, 1, 1, 1][:nj]), Qv=np.diag([10.0, 10, 1, 1, 1, 1][:nj]), Rblk=np.eye(nj) * 2, cR=50.0, lim=np.ones(nj),
              max_input_blk=np.ones(nj), epsilon_O=0.1, MAX_O_ITER=12)
    s = gpu.build_sys_info(robot, nj, H, x0, xg, gpu.line_reference(x0, xg, H), **kw)
    t = O.build_sys_info(orobot, nj, H, x0, xg, O.line_reference(x0, xg, H), **kw)
    ob = [gpu.cylinder((3700, 8500, 1), (3700, 8500, 1200), 0.15, 0.2)]
    noise = np.random.default_rng(nj).standard_normal((12, H * nj)) * 0.1 if mode == "PSGCFS" else None
    cls = gpu.CFS_FANUC if mode == "CFS" else gpu.PSGCFS_FANUC
    got = cls(ob, s, "M200i").optimizer(noise=noise)
    want = O.optimizer("M200i", t, [dict(l=ob[0]["l"], D=0.15, epsilon=0.2)], mode, noise=noise)
    assert got.status == want.status and got.iter_O == want.iter_O and got.iter_O > 2
    # PSGCFS never stops early: 12 forced iterations amplify the 1e-16 differences of the two arithmetic
    # paths (7e-7 rad at nj = 6, H = 40), so those cases are held to the north star's bar instead of TOL_RAD
    assert np.abs(got.x_ - want.x_).max() < (TOL_RAD if mode == "CFS" else 1e-5)
    np.testing.assert_allclose(got.eval.cost_all, want.cost_all, rtol=1e-8 if mode == "CFS" else 1e-6)


def test_two_link_arm_long_horizon(gpu, O):
    # nj = 2, H = 56: nn = 112 selects the 160-row instantiation for the planar arm (main_2L's own H = 40 uses the 96-row one)
    robot, orobot = gpu.robotproperty2("2L"), O.robotproperty2("2L")
    x0, xg, H = np.zeros(2), np.array([np.pi / 2, 0.0]), 56
    kw = dict(Qp=np.diag([10.0, 1.0]), Qv=np.diag([10.0, 1.0]), Rblk=np.diag([5.0, 4.0]), cR=0.1, lim=np.ones(2), max_input_blk=np.ones(2) * 0.25,
              epsilon_O=1e-6, MAX_O_ITER=30)
    x_init = np.tile(np.concatenate([x0, np.zeros(2)]), H)
    s = gpu.build_sys_info(robot, 2, H, x0, xg, x_init, **kw)
    t = O.build_sys_info(orobot, 2, H, x0, xg, x_init, **kw)
    c = np.array([0.3, 0.3, 0.0])
    ob = [dict(shape="circle", l=np.stack([c, c], axis=1), D=0.05, epsilon=0.05)]
    got = gpu.CFS_FANUC(ob, s, "2L").optimizer()
    want = O.optimizer("2L", t, [dict(l=ob[0]["l"], D=0.05, epsilon=0.05)], "CFS")
    assert got.status == want.status and got.iter_O == want.iter_O
    assert np.abs(got.x_ - want.x_).max() < 1e-6


def test_eval_get_cost_b_and_get_cost(gpu, O, route_wp):
    """EVAL.get_Cost_b (Lib/EVAL.m:75-78, called at main_FANUC.m:131-132) and get_cost (:51-53) through cfs_cost_b / cfs_get_cost
    against the oracle's unconstrained QP.  Tolerance: 1e-10 relative on the cost (two fp64 MFMA products against a Cholesky
    solve; cond(QQ) = 6e5 .. 1.7e8)."""
    for name, (R, s, obs), P in (("main_FANUC", gpu.main_FANUC_problem(), O.problem_main_FANUC()),
                                  ("RRTstar_CFS", gpu.RRTstar_CFS_problem(route_wp), O.problem_RRTstar_CFS(route_wp)),
                                  ("main_2L", gpu.main_2L_problem(), O.problem_main_2L())):
        so = P.sys_info
        nn = s.H * s.nu
        ub, _, _, st, _ = O.qp_solve(so.QQ, so.ff, np.zeros((0, nn)), np.zeros(0))          # quadprog(Qaug, paug) without constraints
        assert st == 0
        want = 0.5 * ub @ so.QQ @ ub + so.ff @ ub + so.caug
        ev = gpu.EVAL(s)
        cost_b = ev.get_Cost_b()
        assert abs(cost_b - want) <= 1e-10 * max(abs(want), abs(so.caug)), (name, cost_b, want)
        assert ev.Cost_b == cost_b
        u = np.random.default_rng(3).standard_normal(nn) * 0.01
        want_u = 0.5 * u @ so.QQ @ u + so.ff @ u + so.caug
        assert abs(ev.get_cost(u) - want_u) <= 1e-12 * max(abs(want_u), abs(so.caug)), name
        assert abs(ev.get_cost(np.zeros(nn)) - so.caug) <= 1e-15 * abs(so.caug)             # get_cost(zeros) = caug
    # batched, from a handle of either mode, with the minimiser returned
    R, s, obs = gpu.main_FANUC_problem()
    rng = np.random.default_rng(4)
    ff = s.ff[None] * (1.0 + 0.1 * rng.standard_normal((7, 1)))
    caug = np.full(7, s.caug)
    for mode in ("CFS", "PSGCFS"):
        slv = gpu.CFSBatch(s, 1, [0.2], mode=mode, max_batch=7)
        cb, ub = slv.cost_b(ff, caug, want_u=True)
        H = 0.5 * (s.QQ + s.QQ.T)
        for b in range(7):
            wu = -np.linalg.solve(H, ff[b])
            assert np.abs(ub[b] - wu).max() <= 1e-9 * np.abs(wu).max()
            wc = 0.5 * wu @ s.QQ @ wu + ff[b] @ wu + caug[b]
            assert abs(cb[b] - wc) <= 1e-10 * abs(caug[b])
        slv.close()


def test_rollout_against_the_reference_own_stored_trajectory(gpu):
    """The reference's stored (u -> x_) pair (data/M16_ref_2.mat:uref, data/good_xori.mat:xuori; fixture
    tests/golden/reference_rollout_M16.npz, see tests/test_oracle_golden.py) through the PRODUCT: a CFS_FANUC problem whose QP
    has H = I, f = -uref and no active constraint (one far obstacle, wide limits), so that the fused kernel's QP returns
    u = uref and its rollout (prefix sums in LDS, not the literal recurrence of Lib/CFS_FANUC.m:90-94) must land on the
    trajectory MATLAB saved.  Bar: u exact, x_ within 1e-13 rad of the stored one (the two summation orders differ by
    a few ulp of |x| <= 2.6)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_rollout_M16.npz"))
    uref, xuori = g["uref"], g["xuori"]
    robot = gpu.robotproperty2("M16iB")
    th0, th1 = xuori[:5], xuori[240:245]
    s = gpu.build_sys_info(robot, 5, 24, th0, th1, gpu.line_reference(th0, th1, 24), Qp=np.eye(5), Qv=np.eye(5), Rblk=np.eye(5), cR=1.0,
                           lim=np.ones(5), max_input_blk=np.full(5, 0.5), epsilon_O=1e-12, MAX_O_ITER=1)
    s.QQ = s.Qaug = np.eye(120)
    s.ff = s.paug = -uref
    s.xR = xuori[:10].reshape(10, 1).copy()
    del s.weights                                                      # a dense family: QQ is not what the weights would assemble
    far = [gpu.cylinder((40000, 40000, 1), (40000, 40000, 900), 0.2, 0.25)]
    got = gpu.CFS_FANUC(far, s, "M16iB").optimizer()
    assert got.status == 1 and got.iter_O == 2                          # one iteration, then MAX_O_ITER
    np.testing.assert_array_equal(got.u, uref)
    assert np.abs(got.x_ - xuori[10:]).max() < 1e-13, np.abs(got.x_ - xuori[10:]).max()


def test_forward_kinematics_against_the_reference_own_stored_capsules(gpu):
    """The reference's stored FK output (figure/M16iBCapsules.mat:RoCap; fixture tests/golden/reference_capsules_M16iB.npz, see
    tests/test_oracle_golden.py) through the PRODUCT: cfs_dist_arm's capsule end points for the M16iB at the stored pose, six
    joints.  Eleven end points within 2e-15 m; the twelfth differs by the 0.01 m of a constant the reference has since edited
    (robotproperty2.m:89), and with the constant as it was all twelve agree."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_capsules_M16iB.npz"))
    want = g["p"].transpose(0, 2, 1)                                   # (link, end point, xyz)
    th = np.array([[0.0, 1.5708, 0.0, 0.0, -np.pi / 2, np.pi]])
    far = np.array([[40.0, 40.0, 0.0, 40.0, 40.0, 1.0]])
    robot = gpu.robotproperty2("M16iB")
    base = np.asarray(robot.base, float).ravel()
    _, _, pos = gpu.dist_arm(robot, th, far, want_pos=True)
    err = np.abs(pos[0] - base - want).max(axis=2)
    legacy = np.zeros((6, 2), bool)
    legacy[4, 1] = True
    assert err[~legacy].max() < 2e-15, err
    assert abs(err[4, 1] - 0.01) < 1e-12
    old = copy.deepcopy(robot)
    old.cap[4].p = np.array([[0.0, 0.0], [0.0, 0.0], [-0.05, 0.10]])
    _, _, pos = gpu.dist_arm(old, th, far, want_pos=True)
    assert np.abs(pos[0] - base - want).max() < 2e-15
