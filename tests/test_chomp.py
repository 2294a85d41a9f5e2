"""CHOMP_FANUC: the oracle's derivest restatement against analytic derivatives (CPU), and the
HIP path against the oracle (GPU).  Parity unpinned: neither CHOMP_FANUC.m nor DERIVESTsuite can be run here."""
import copy

import numpy as np
import pytest


def test_derivest_restatement_matches_analytic_derivatives(O):
    cases = [(np.sin, np.cos, 0.7), (np.exp, np.exp, 1.3), (lambda x: np.sin(3 * x) + x * x, lambda x: 3 * np.cos(3 * x) + 2 * x, -0.4),
             (np.tan, lambda x: 1 / np.cos(x) ** 2, 0.01), (lambda x: x ** 3, lambda x: 3 * x * x, 0.0)]
    for f, df, x in cases:
        d, err = O.derivest(f, x)
        assert abs(d - df(x)) <= 1e-11 * max(1.0, abs(df(x))), (x, d, df(x))
        assert abs(d - df(x)) <= max(20 * err, 1e-13)            # the suite's own error estimate is honest here


def test_chomp_dm_has_no_joint_offset_and_the_literal_update_diverges(O):
    P = O.problem_main_FANUC()
    s, obs = P.sys_info, [dict(l=o["l"], D=o["D"], epsilon=o["epsilon"]) for o in P.obs]
    th = np.asarray(s.x_).reshape(30, 10)[14, :5]
    d = O.chomp_dm(s.robot, th, obs[0]["l"], 0.2)
    th_shift = th.copy(); th_shift[1] += np.pi / 2                # dist_arm subtracts pi/2 from joint 2, dm_f does not
    assert abs(d.min() + 0.2 - O.dist_arm(s.robot, th_shift, obs[0]["l"])[0]) < 1e-12
    s5 = copy.copy(s); s5.MAX_O_ITER = 6
    w = O.chomp_optimizer("M200i", s5, obs, np.zeros(150))
    assert w.iter_O == 7 and np.all(np.diff(w.e_u_all) > 0)      # step 3*alpha > 2/lambda_max: |du| grows every iteration
    assert np.all(w.e_u_all[1:] / w.e_u_all[:-1] > 1.9)


@pytest.mark.gpu
def test_chomp_against_oracle(gpu, O):
    R, s, obs = gpu.main_FANUC_problem()
    P = O.problem_main_FANUC()
    rng = np.random.default_rng(4)
    for K, uref in ((4, np.zeros(150)), (8, 0.01 * rng.standard_normal(150))):
        s.MAX_O_ITER = K
        so = copy.copy(P.sys_info); so.MAX_O_ITER = K
        # a second obstacle whose band (epsilon) the straight line crosses, so that both regimes of dcostObs_f are exercised
        obs2 = obs + [gpu.cylinder((2950, 8950, 1), (2950, 8950, 900), 0.05, 0.35)]
        got = gpu.CHOMP_FANUC([dict(num_obs=2)] + obs2, s, uref, R).optimizer()
        want = O.chomp_optimizer("M200i", so, [dict(l=o["l"], D=o["D"], epsilon=o["epsilon"]) for o in obs2], uref)
        assert got.iter_O == want.iter_O == K + 1
        scale = np.abs(want.u).max()
        assert np.abs(got.u - want.u).max() < 1e-9 * scale        # the iteration doubles every error each step: relative bar
        np.testing.assert_allclose(got.eval.cost_all, want.cost_all, rtol=1e-9)
        np.testing.assert_allclose(got.eval.e_u_all, want.e_u_all, rtol=1e-9)


@pytest.mark.gpu
def test_chomp_batch_and_m16ib(gpu, O):
    from motionplanning_5d_m_amd import workloads
    s, bt = workloads.config3(lambda rb, th, ob: gpu.dist_arm(rb, th, ob)[0], B=9, nobs=3, seed=5)
    s.MAX_O_ITER = 3
    slv = gpu.CFSBatch(s, 3, [0.25] * 3, mode="CFS", max_batch=9)
    u0 = 0.02 * np.random.default_rng(0).standard_normal((9, 150))
    r = slv.chomp(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs, u0, [0.1, 0.2, 0.05], [0.3, 0.25, 0.4])
    so = copy.copy(O.problem_main_FANUC().sys_info)              # same family (main_FANUC.m:64-127 cost matrices), the oracle's robot type
    so.MAX_O_ITER = 3
    assert np.array_equal(np.asarray(so.QQ), np.asarray(s.QQ))
    for b in range(9):
        so.x_, so.xR1, so.ff, so.caug = bt.x_init[b], bt.xR1[b], bt.ff[b], bt.caug[b]
        ob = [dict(l=np.stack([bt.obs[b, j, :3], bt.obs[b, j, 3:]], axis=1), D=d, epsilon=e) for j, (d, e) in enumerate(zip([0.1, 0.2, 0.05], [0.3, 0.25, 0.4]))]
        w = O.chomp_optimizer("M200i", so, ob, u0[b])
        assert r.iter_O[b] == w.iter_O == 4
        assert np.abs(r.u[b] - w.u).max() < 1e-9 * np.abs(w.u).max()
        np.testing.assert_allclose(r.cost_all[b], w.cost_all, rtol=1e-9)
    slv.close()
