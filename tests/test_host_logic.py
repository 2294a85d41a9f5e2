"""Host-side mirrors (robot model, problem builder, workloads) against the oracle's literal restatements (CPU)."""
import numpy as np

import motionplanning_5d_m_amd as pkg
from motionplanning_5d_m_amd import sysinfo, workloads


def test_robotproperty2_matches_literal_restatement(O):
    for rid in ("M200i", "M16iB", "2L"):
        a, b = pkg.robotproperty2(rid), O.robotproperty2(rid)
        np.testing.assert_array_equal(a.DH, b.DH)
        np.testing.assert_array_equal(a.base, b.base)
        np.testing.assert_array_equal(a.A, b.A)
        np.testing.assert_array_equal(a.B, b.B)
        np.testing.assert_array_equal(a.T, b.T)
        assert a.nlink == b.nlink and a.delta_t == b.delta_t and len(a.cap) == len(b.cap)
        for ca, cb in zip(a.cap, b.cap):
            np.testing.assert_array_equal(ca.p, cb)
    rb = pkg.to_c_robot(pkg.robotproperty2("M200i"))
    assert rb.kind == 1 and rb.nlink == 6 and rb.DH[1 + 2 * 6] == 0.440 and rb.cap[1 * 6 + 0] == -0.4


def test_sys_info_matches_literal_restatement(O, route_wp):
    # closed-form double integrator vs matrix_power loops of main_FANUC.m:79-86
    for (R, s, obs), P in ((pkg.main_FANUC_problem(), O.problem_main_FANUC()),
                           (pkg.main_2L_problem(), O.problem_main_2L()),
                           (pkg.RRTstar_CFS_problem(route_wp), O.problem_RRTstar_CFS(route_wp))):
        t = P.sys_info
        assert R == P.ROBOT and s.H == t.H and s.njoint == t.njoint
        np.testing.assert_array_equal(s.Aaug, t.Aaug)
        np.testing.assert_array_equal(s.Baug, t.Baug)
        np.testing.assert_allclose(s.QQ, t.QQ, rtol=1e-13, atol=1e-9)
        np.testing.assert_allclose(s.ff, t.ff, rtol=1e-12, atol=1e-8)
        assert abs(s.caug - t.caug) <= 1e-12 * abs(t.caug)
        np.testing.assert_allclose(s.x_, t.x_, rtol=0, atol=1e-15)
        np.testing.assert_array_equal(s.MAX_input, t.MAX_input)
        np.testing.assert_array_equal(s.lim, t.lim)
        assert abs(s.alpha - t.alpha) < 1e-12 * t.alpha and s.epsilon_O == t.epsilon_O and s.MAX_O_ITER == t.MAX_O_ITER
        for a, b in zip(obs, P.obs):
            np.testing.assert_array_equal(a["l"], b["l"])
            assert a["D"] == b["D"] and a["epsilon"] == b["epsilon"]


def test_problem_facts(O):
    # conditioning of QQ, its tiny asymmetry (N6), alpha = 1/sigma_max
    _, s, _ = pkg.main_FANUC_problem()
    ev = np.linalg.eigvalsh((s.QQ + s.QQ.T) / 2)
    assert abs(ev[0] - 100.006) < 1e-2 and abs(ev[-1] / 5.69e7 - 1) < 1e-2
    assert 0 <= np.abs(s.QQ - s.QQ.T).max() < 1e-9
    assert abs(s.alpha * ev[-1] - 1) < 1e-9
    assert s.x_.shape == (300,) and s.ff.shape == (150,) and s.xR.shape == (10, 1)


def test_config3_workload_is_deterministic_and_valid(O):
    robot = O.robotproperty2("M200i")

    def dist_fn(rb, th, ob):
        return np.array([[O.dist_arm(robot, t, np.stack([o[:3], o[3:]], axis=1))[0] for o in ob] for t in th])

    s, b1 = workloads.config3(dist_fn, B=6)
    _, b2 = workloads.config3(dist_fn, B=6)
    for k in ("x_init", "xR1", "ff", "caug", "obs", "noise"):
        np.testing.assert_array_equal(getattr(b1, k), getattr(b2, k))
    assert b1.obs.shape == (6, 8, 6) and b1.noise.shape == (6, 20, 150) and s.H == 30
    rad = np.hypot(b1.obs[:, :, 0] - robot.base[0], b1.obs[:, :, 1] - robot.base[1])
    assert (rad >= 0.35 - 1e-12).all() and (rad <= 0.75 + 1e-12).all()
    assert (b1.obs[:, :, 2] == 0.001).all() and (b1.obs[:, :, 5] >= 0.6).all() and (b1.obs[:, :, 5] <= 1.5).all()
    for b in range(6):          # rejection rule: no obstacle closer than 0.25 m to the start or goal pose
        for j in range(8):
            l = np.stack([b1.obs[b, j, :3], b1.obs[b, j, 3:]], axis=1)
            assert O.dist_arm(robot, b1.x0[b], l)[0] >= 0.25 and O.dist_arm(robot, b1.xg[b], l)[0] >= 0.25
    # per-problem terms equal what the single-problem builder gives
    t = sysinfo.build_sys_info(s.robot, 5, 30, b1.x0[2], b1.xg[2], b1.x_init[2], Qp=sysinfo.FANUC_Qp, Qv=sysinfo.FANUC_Qp,
                               Rblk=sysinfo.FANUC_Rblk, cR=50.0, lim=np.ones(5), max_input_blk=np.ones(5), epsilon_O=0.1,
                               MAX_O_ITER=20)
    np.testing.assert_allclose(b1.ff[2], t.ff, rtol=1e-13)
    assert abs(b1.caug[2] - t.caug) < 1e-12 * abs(t.caug)
    np.testing.assert_array_equal(b1.x_init[2], sysinfo.line_reference(b1.x0[2], b1.xg[2], 30))


def test_cubic_resample_end_points(route_wp):
    r = sysinfo.cubic_resample(route_wp, 0.5, 40)
    assert r.shape == (5, 41)
    np.testing.assert_array_equal(r[:, 0], route_wp[:, 0])
    np.testing.assert_allclose(r[:, -1], route_wp[:, -1], atol=1e-15)


def test_config5_workload_shapes_and_determinism():
    from motionplanning_5d_m_amd import workloads
    s, bt, tri = workloads.config5(B=5, n_tri=3000)
    s2, bt2, tri2 = workloads.config5(B=5, n_tri=3000)
    assert s.H == 50 and bt.x_init.shape == (5, 500) and bt.ff.shape == (5, 250) and bt.obs.shape == (5, 1, 6) and bt.noise.shape == (5, 20, 250)
    assert tri.shape[1:] == (3, 3) and 2000 < tri.shape[0] < 4500
    assert np.array_equal(tri, tri2) and np.array_equal(bt.x_init, bt2.x_init)
    base = s.robot.base
    v = tri.reshape(-1, 3)
    assert v[:, 0].min() > base[0] + 0.4 and np.isfinite(v).all()      # the map stands in front of the arm, clear of its base


def test_eval_host_bookkeeping_follows_the_reference():
    """EVAL.stop_outer / store_result (Lib/EVAL.m:55-73) are host-side bookkeeping in the reference too: the constructor
    state (x_old = ones :47, cost_old = 100000 :29), the two stop conditions and the history appends."""
    import motionplanning_5d_m_amd as pkg
    R, s, obs = pkg.main_FANUC_problem()
    ev = pkg.EVAL(s)
    assert ev.cost_old == 100000.0 and ev.cost_new == 0.0 and ev.Cost_b == 0.0 and np.all(ev.x_old == 1.0)
    assert ev.x_.shape == (s.H * s.nstate,) and ev.MAX_O_ITER == s.MAX_O_ITER and ev.epsilon_O == s.epsilon_O
    assert not ev.stop_outer(1)                       # ||x_ - ones|| is far above epsilon_O and iter_O <= MAX_O_ITER
    assert ev.stop_outer(s.MAX_O_ITER + 1)            # "MAX_ITER"
    ev.x_old = ev.x_ + 0.5 * s.epsilon_O / np.sqrt(ev.x_.size)
    assert ev.stop_outer(3)                           # "Converged at step3"
    ev.u_old, ev.cost_new = np.zeros(s.H * s.nu), 5.0
    ev.store_result(np.full(s.H * s.nu, 2.0))
    assert ev.cost_all.tolist() == [5.0] and ev.e_cost_all.tolist() == [abs(100000.0 - 5.0)]
    assert abs(ev.e_u_all[0] - 2.0 * np.sqrt(s.H * s.nu)) < 1e-12
