"""Oracle against the committed fixtures and against the survey's independent scratch numbers (CPU).

"parity unpinned": the reference has no tests/golden vectors and MATLAB/quadprog cannot run here.
What pins the oracle: (1) the distLinSeg doc example, (2) KKT certificates + an independent NNLS
solve (test_oracle_qp.py), (3) the numbers below, which SURVEY.md (N3, N8-N10) obtained with a
separately written numpy/NNLS restatement -- two independent restatements agree.
"""
import numpy as np


def _run(O, P, mode="CFS", noise=None):
    return O.optimizer(P.ROBOT, P.sys_info, P.obs, mode, noise=noise, history=True)


def _check(r, g, name):
    assert [r.iter_O, r.total_iter, r.status] == list(g[name + "/iter_status"])
    np.testing.assert_allclose(r.x_, g[name + "/x_"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r.u, g[name + "/u"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r.cost_all, g[name + "/cost_all"], rtol=1e-12)


def test_main_fanuc_cfs(O, golden):
    P = O.problem_main_FANUC()
    r = _run(O, P)
    _check(r, golden, "main_FANUC_CFS")
    # SURVEY N3 / N8 (independent restatement): converged at step 11, costs, waypoints
    assert r.status == 0 and r.iter_O == 11
    c = r.cost_all - P.sys_info.caug
    assert abs(c[0] + 122076.00) < 0.01 and abs(c[-1] + 122461.43) < 0.01
    x = r.x_.reshape(30, 10)
    np.testing.assert_allclose(x[0, :5], [0.775666, 0.024983, 0.229527, 0.143153, -1.169345], atol=1e-6)
    np.testing.assert_allclose(x[14, :5], [-0.170879, -0.202055, 1.012415, 0.183991, -0.739707], atol=1e-6)
    np.testing.assert_allclose(x[29, :5], [-0.782747, 0.028201, 0.218965, 0.145129, -1.177633], atol=1e-6)
    dx = [np.linalg.norm(r.hist_x[k] - (r.hist_x[k - 1] if k else P.sys_info.x_)) for k in range(10)]
    np.testing.assert_allclose(dx, [15.98, 5.71, 13.87, 1.42, 0.80, 0.46, 0.30, 0.20, 0.106, 0.070], atol=6e-3)
    assert r.kkt[0] < 1e-12 and r.kkt[1] < 1e-12 and r.kkt[3] < 1e-9


def test_main_fanuc_psgcfs(O, golden):
    r = _run(O, O.problem_main_FANUC(), "PSGCFS", golden["main_FANUC_PSGCFS/noise"])
    _check(r, golden, "main_FANUC_PSGCFS")
    assert r.status == 1 and r.iter_O == 21          # x_old is never refreshed -> always MAX_O_ITER


def test_main_2l(O, golden):
    r = _run(O, O.problem_main_2L())
    _check(r, golden, "main_2L_CFS")
    assert r.status == 2 and r.iter_O == 2           # the QP of outer iteration 2 is infeasible
    P = O.problem_main_2L(lim=(1, 1))
    r = _run(O, P)
    _check(r, golden, "main_2L_lim1_CFS")
    assert r.status == 0 and r.iter_O == 9 and abs((r.cost_all - P.sys_info.caug)[-1] + 123412.5759) < 1e-3


def test_rrtstar_cfs(O, golden, route_wp):
    P = O.problem_RRTstar_CFS(route_wp)
    r = _run(O, P)
    _check(r, golden, "RRTstar_CFS")
    assert r.status == 0 and r.iter_O == 18 and abs((r.cost_all - P.sys_info.caug)[-1] + 73637.01) < 0.01


def test_first_iteration_fixtures(O, golden, route_wp):
    for name, P in (("main_FANUC_CFS", O.problem_main_FANUC()), ("main_2L_CFS", O.problem_main_2L()),
                    ("RRTstar_CFS", O.problem_RRTstar_CFS(route_wp))):
        s = P.sys_info
        A, b, dist, lid, grad = O.get_con(P.ROBOT, s, P.obs, s.x_, np.zeros(s.H * s.nu))
        np.testing.assert_array_equal(b, golden[name + "/binq1"])
        np.testing.assert_array_equal(dist, golden[name + "/dist1"])
        np.testing.assert_array_equal(lid, golden[name + "/linkid1"])
        np.testing.assert_array_equal(grad, golden[name + "/grad1"])
        np.testing.assert_allclose([A.sum(), np.abs(A).sum()], golden[name + "/Ainq1_sum"], rtol=1e-13)


def test_rollout_reproduces_the_reference_own_stored_trajectory(O):
    """The one stored OUTPUT of the reference's MATLAB runs that lies on the hot path: data/good_xori.mat:xuori (250 x 1) is the
    state trajectory its legacy CFS script saved, data/M16_ref_2.mat:uref (120 x 1) the inputs that produced it
    (M16iB/main_CFS.m:19-21, :162-169).  The oracle's rollout (Lib/CFS_FANUC.m:90-94: xR(:,i) = A*xR(:,i-1) + B*u(i-1) with the
    double integrator of robotproperty2.m:136-139, delta_t = 0.5; x_ stacked [theta; omega] per waypoint, waypoint 0 dropped:
    SURVEY N5) reproduces all 240 numbers BIT FOR BIT; data/M16_ref_2.mat:xref is the same trajectory with 75 angles shifted
    by 2*pi.  Fixture: tests/golden/reference_rollout_M16.npz (data only; tests/golden/make_reference_rollout.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_rollout_M16.npz"))
    uref, xref, xuori = g["uref"], g["xref"], g["xuori"]
    assert uref.shape == (120,) and xref.shape == (250,) and xuori.shape == (250,)
    dt = O.robotproperty2("M16iB").delta_t
    assert dt == 0.5
    x_ = O.rollout(24, 5, dt, xuori[:10], uref)
    np.testing.assert_array_equal(x_, xuori[10:])                       # the reference's own MATLAB output, to the last bit
    k = (xref - xuori) / (2 * np.pi)
    assert np.abs(k - np.round(k)).max() < 1e-15 and int(np.abs(np.round(k)).sum()) == 75     # xref = xuori + 2 pi on 75 angles
    # what the stored solution looks like: a feasible motion of the driver's problem class (velocity and input limits of main_FANUC.m)
    X = xuori.reshape(25, 10)
    assert np.abs(X[:, 5:]).max() < 1.0 and np.abs(uref).max() < 0.5


REF_CAPSULE_POSE = np.array([0.0, 1.5708, 0.0, 0.0, -np.pi / 2, np.pi])     # robotproperty2.m:67-72's theta column with theta_1 = 0


def test_forward_kinematics_reproduces_the_reference_own_stored_capsules(O):
    """figure/M16iBCapsules.mat:RoCap (robotproperty2.m:96 loads the file) is the stored OUTPUT of the reference's CapPos
    (Lib/functions/CapPos.m:8-22) for the M16iB at the DH table's own pose with theta_1 = 0 and base = 0: six capsules, twelve
    end points in world coordinates.  The oracle's FK lands on eleven of them to 1e-15 m; the twelfth (second end point of
    capsule 5) is off by exactly 0.01 m along the capsule axis because robotproperty2.m:89 has since changed that constant
    (`-0.05 0.110`; the stored capsule is 0.15 m long, i.e. it was 0.100).  Radii as robotproperty2.m:76-92.
    Fixture: tests/golden/reference_capsules_M16iB.npz (data only; tests/golden/make_reference_capsules.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_capsules_M16iB.npz"))
    want = g["p"].transpose(0, 2, 1)                                   # (link, end point, xyz)
    rb = O.robotproperty2("M16iB")
    np.testing.assert_array_equal(g["r"], np.asarray(rb.cap_r, float))
    got = np.asarray(O.arm_pos(rb, REF_CAPSULE_POSE)) - np.asarray(rb.base).ravel()
    err = np.abs(got - want).max(axis=2)                               # (link, end point)
    legacy = np.zeros((6, 2), bool)
    legacy[4, 1] = True
    assert err[~legacy].max() < 2e-15, err
    assert abs(err[4, 1] - 0.01) < 1e-12 and abs(np.linalg.norm(want[4, 1] - want[4, 0]) - 0.15) < 1e-6
    # with the constant as it was when the file was written, all twelve
    import copy
    rb_old = copy.deepcopy(rb)
    rb_old.cap[4] = np.array([[0.0, 0.0], [0.0, 0.0], [-0.05, 0.10]])
    got = np.asarray(O.arm_pos(rb_old, REF_CAPSULE_POSE)) - np.asarray(rb.base).ravel()
    assert np.abs(got - want).max() < 2e-15
