"""Oracle vs the reference's known answers and invariants (CPU)."""
import numpy as np


def test_distlinseg_doc_example(O):
    # the single known-answer in the reference: Lib/functions/distLinSeg.m:15-18
    d, pts = O.dist_lin_seg([0, 0, 0], [1, 1, 0], [1, 0, 0], [2, 0, 0])
    assert abs(d - 0.7071) < 5e-5 and abs(d - np.sqrt(0.5)) < 1e-15
    np.testing.assert_allclose(pts, [0.5, 0.5, 0, 1, 0, 0], atol=1e-15)


def test_distlinseg_branches(O):
    # both points (distLinSeg.m:50-53), segment/point (:39-48), parallel (:55-66), clamped general (:67-82)
    assert O.dist_lin_seg([0, 0, 0], [0, 0, 0], [3, 4, 0], [3, 4, 0])[0] == 5.0
    d, p = O.dist_lin_seg([0, 0, 0], [2, 0, 0], [1, 1, 0], [1, 1, 0])
    assert d == 1.0 and np.allclose(p, [1, 0, 0, 1, 1, 0])
    d, p = O.dist_lin_seg([1, 1, 0], [1, 1, 0], [0, 0, 0], [2, 0, 0])
    assert d == 1.0 and np.allclose(p, [1, 1, 0, 1, 0, 0])
    assert O.dist_lin_seg([0, 0, 0], [1, 0, 0], [0, 2, 0], [1, 2, 0])[0] == 2.0
    d, p = O.dist_lin_seg([0, 0, 0], [1, 0, 0], [3, 1, 0], [3, 5, 0])
    assert abs(d - np.hypot(2, 1)) < 1e-15 and np.allclose(p, [1, 0, 0, 3, 1, 0])
    rng = np.random.default_rng(0)
    for _ in range(200):  # symmetric in its two segments, never above any end-point distance
        a, b, c, e = rng.normal(size=(4, 3))
        d1, d2 = O.dist_lin_seg(a, b, c, e)[0], O.dist_lin_seg(c, e, a, b)[0]
        assert abs(d1 - d2) < 1e-12 and d1 <= min(np.linalg.norm(a - c), np.linalg.norm(b - e)) + 1e-12


def test_fk_structure(O):
    robot = O.robotproperty2("M200i")
    th = np.array([0.3, -0.2, 0.5, 0.1, -1.0])
    pos = O.arm_pos(robot, th)
    # capsules 1 and 3 are points (robotproperty2.m:37,43) -> the D1==0 branch of distLinSeg on every call
    assert np.array_equal(pos[0, 0], pos[0, 1]) and np.array_equal(pos[2, 0], pos[2, 1])
    np.testing.assert_allclose(pos[0, 0], robot.base + [0.05 * np.cos(0.3), 0.05 * np.sin(0.3), 0], atol=1e-15)
    assert abs(np.linalg.norm(pos[1, 1] - pos[1, 0]) - 0.4) < 1e-12  # rigid link lengths
    assert abs(np.linalg.norm(pos[3, 1] - pos[3, 0]) - 0.4) < 1e-12
    assert abs(np.linalg.norm(pos[4, 1] - pos[4, 0]) - 0.27) < 1e-12
    r2 = O.robotproperty2("2L")
    p2 = O.arm_pos(r2, np.array([np.pi / 2, -np.pi / 2]))
    np.testing.assert_allclose(p2[0], [[0, 0, 0], [0, 0.3, 0]], atol=1e-16)
    np.testing.assert_allclose(p2[1], [[0, 0.3, 0], [0.2, 0.3, 0]], atol=1e-16)


def test_near_zero_surrogate(O):
    # dist_arm_3D_200i_2.m:22-24: |dis| < 1e-4 -> -|closest point on link - link end|
    r2 = O.robotproperty2("2L")
    c = np.array([0.15, 0.0, 0.0])
    d, lid = O.dist_arm(r2, np.zeros(2), np.stack([c, c], axis=1))
    assert lid == 1 and abs(d + 0.15) < 1e-15
    # the canonical demo takes that branch on its initial line: waypoint 18, link 5
    P = O.problem_main_FANUC()
    _, _, dist, lid, _ = O.get_con(P.ROBOT, P.sys_info, P.obs, P.sys_info.x_, np.zeros(150))
    assert dist.argmin() == 17 and lid[0, 17] == 5 and abs(dist.min() + 0.0743) < 5e-5


def test_num_jac_is_literal(O):
    # num_jac.m:8-16: xp is copied once and never restored, so column i is differenced at a base point
    # already shifted by -eps/2 in coordinates 1..i-1
    robot = O.robotproperty2("M200i")
    th = np.array([0.4, 0.1, 0.3, 0.2, -1.0])
    l = np.array([[3.5, 8.3, 0.1], [3.6, 8.5, 1.0]]).T
    g = O.num_jac_dist(robot, th, l)
    eps = 1e-5
    lit, clean = np.zeros(5), np.zeros(5)
    xp = th.copy()
    for i in range(5):
        xp[i] = th[i] + eps / 2
        hi = O.dist_arm(robot, xp, l)[0]
        xp[i] = th[i] - eps / 2
        lit[i] = (hi - O.dist_arm(robot, xp, l)[0]) / eps
        e = np.zeros(5); e[i] = eps / 2
        clean[i] = (O.dist_arm(robot, th + e, l)[0] - O.dist_arm(robot, th - e, l)[0]) / eps
    assert np.array_equal(g, lit)
    assert np.abs(g - clean).max() > 1e-9 and np.abs(g - clean).max() < 1e-4 and g[0] == clean[0]


def test_get_con_row_order_and_duplicates(O):
    # CFS_FANUC.m:119-129: per (obstacle, waypoint): 1 collision row, nj +vel rows, nj -vel rows; the
    # velocity rows are re-appended for every obstacle
    P = O.problem_main_FANUC(); s = P.sys_info
    obs = P.obs + [dict(l=P.obs[0]["l"] + 0.3, D=0.2, epsilon=0.25)]
    A, b, dist, _, grad = O.get_con(P.ROBOT, s, obs, s.x_, np.zeros(150))
    assert A.shape == (2 * 30 * 11, 150)
    np.testing.assert_array_equal(A[1:11], A[331:341])
    np.testing.assert_array_equal(b[1:11], b[331:341])
    i = 7
    np.testing.assert_allclose(A[i * 11, :], -(grad[0, i] @ s.Baug[i * 10:i * 10 + 5, :]), atol=1e-15)
    assert abs(b[i * 11] - (dist[0, i] - 0.25)) < 1e-15          # u = 0
    np.testing.assert_array_equal(A[i * 11 + 1:i * 11 + 6], s.Baug[i * 10 + 5:i * 10 + 10])
    np.testing.assert_array_equal(A[i * 11 + 6:i * 11 + 11], -s.Baug[i * 10 + 5:i * 10 + 10])
    np.testing.assert_array_equal(b[i * 11 + 1:i * 11 + 11], np.ones(10))


def test_geometry_golden(O, golden):
    for rid, nj in (("M200i", 5), ("M16iB", 5), ("2L", 2)):
        robot = O.robotproperty2(rid)
        th, obs = golden[f"geom_{rid}/theta"], golden[f"geom_{rid}/obs"]
        for n in range(th.shape[0]):
            np.testing.assert_array_equal(O.arm_pos(robot, th[n]), golden[f"geom_{rid}/pos"][n])
            for j in range(3):
                d, lid = O.dist_arm(robot, th[n], np.stack([obs[j, :3], obs[j, 3:]], axis=1))
                assert d == golden[f"geom_{rid}/d"][n, j] and lid == golden[f"geom_{rid}/linkid"][n, j]
