"""Every shortcut the fused solver takes, switched OFF through the declared per-handle debug entry
(cfs_debug_set_options, include/cfs_hip.h) and compared with the default, on the config-3 batch (1024 problems, both
solvers) and on 512 routes of config 4's shape.

* step-free infeasibility certificate (both solvers since round 3): it only ever REPLACES the dual steps that would prove the same QP
  infeasible, so status, iteration counts, u and x_ must be the same BITS with and without it; and every linearisation it
  flags must be infeasible for the oracle's QP too (the reference ignores quadprog's exitflag, Lib/CFS_FANUC.m:85: what an
  infeasible QP "returns" there is undefined, so a wrong verdict here would be a silent change of behaviour).
* warm start of the active set: the same strictly convex QP from another S-pair: same optimum, other rounding (~1e-16 of u
  per QP), which the outer iteration then amplifies like any other perturbation.  Status and iteration count must be
  identical and x_ within 1e-5 rad (the north-star bar) on every problem the oracle pins (helpers.chaotic_problems decides,
  from the ORACLE alone, where a 1e-12 perturbation is amplified beyond 1e-6 rad), median below 1e-8 rad.
  Measured (MI355X, round 3): status and iteration counts identical on ALL problems, chaotic ones included; pinned problems
  differ by at most 2.9e-7 rad (config 3 CFS), 6.8e-9 (PSGCFS), 1.8e-6 (config-4 shape).
* rollouts of the entering direction by prefix sums in LDS (default) vs gathered from the precomputed family-matrix
  rollouts: other rounding again, same bars (measured 1.2e-7 / 2.4e-6 rad).
* candidate pruning of the linearisation: bit-identical (tests/test_gpu_first_iteration.py).
"""
import numpy as np
import pytest

from helpers import oracle_obs

pytestmark = pytest.mark.gpu


def _solve(gpu, s, bt, mode, **flags):
    B = bt.x_init.shape[0]
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    slv = gpu.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=B)
    slv.debug_options(**flags)
    r = slv.solve(bt.x_init, bt.xR1, bt.ff, bt.caug, bt.obs, noise=bt.noise if (mode == "PSGCFS" and bt.noise is not None) else None)
    slv.close()
    return r


@pytest.fixture(scope="module")
def base(gpu, c3, c4):
    """default solves, once"""
    return {("c3", "CFS"): _solve(gpu, *c3, "CFS"), ("c3", "PSGCFS"): _solve(gpu, *c3, "PSGCFS"), ("c4", "CFS"): _solve(gpu, *c4, "CFS")}


@pytest.mark.parametrize("tag,mode", [("c3", "CFS"), ("c4", "CFS"), ("c3", "PSGCFS")])
def test_certificate_on_off_same_bits(gpu, c3, c4, base, tag, mode):
    s, bt = c3 if tag == "c3" else c4
    on, off = base[(tag, mode)], _solve(gpu, s, bt, mode, no_certificate=True)
    for k in ("status", "iter_O", "u", "x_", "cost_all", "e_u_all"):
        np.testing.assert_array_equal(getattr(on, k), getattr(off, k), err_msg=k)
    saved = int(off.total_iter.sum()) - int(on.total_iter.sum())
    print(f"[{tag} {mode}] certificate: {int((on.total_iter != off.total_iter).sum())} problems end earlier, {saved} of {int(off.total_iter.sum())} "
          f"active-set steps saved; {int((on.status == 2).sum())} infeasible linearisations in all")
    assert (on.total_iter <= off.total_iter).all()
    if tag == "c3":
        assert saved > 0                                           # it does fire on this workload (a third of it is infeasible)


@pytest.mark.parametrize("tag,mode", [("c3", "CFS"), ("c3", "PSGCFS"), ("c4", "CFS")])
@pytest.mark.parametrize("flag", ["no_warm_start", "gather_rollouts"])
def test_other_rounding_same_answers(gpu, c3, c4, c3_oracle, c4_oracle, base, tag, mode, flag):
    if flag == "gather_rollouts" and mode == "PSGCFS":
        pytest.skip("H = I has closed-form normals: no family-matrix gather")
    s, bt = c3 if tag == "c3" else c4
    _, chaotic, _ = (c3_oracle if tag == "c3" else c4_oracle)(mode)
    a, b = base[(tag, mode)], _solve(gpu, s, bt, mode, **{flag: True})
    same = (a.status == b.status) & (a.iter_O == b.iter_O)
    err = np.abs(a.x_ - b.x_).max(axis=1)
    pinned = ~chaotic
    print(f"[{tag} {mode}] {flag}: status / iteration count differ on {int((~same).sum())} problems ({int((~same & pinned).sum())} pinned by the oracle), "
          f"max |dx_| over the pinned ones {err[pinned & same].max():.2e} rad, over all {err[same].max():.2e}; "
          f"active-set steps {int(a.total_iter.sum())} vs {int(b.total_iter.sum())}")
    assert same[pinned].all(), np.nonzero(~same & pinned)[0]
    assert err[pinned].max() < 1e-5, (np.nonzero(pinned & (err >= 1e-5))[0], err[pinned].max())
    assert np.median(err[pinned]) < 1e-8, np.median(err[pinned])      # config-4 shape: cond(H) = 7e6, typical differences 1e-10 .. 1e-7


def test_every_certificate_hit_is_infeasible_for_the_oracle_psgcfs(gpu, O, c3):
    """The same for the projection QP of PSGCFS_FANUC (H = I, margins obs{j}.D, NO input bounds: Lib/PSGCFS_FANUC.m:117-120):
    the first linearisation of every config-3 problem through cfs_qp, projecting the first PSG iterate u_ = -alpha (ff + 5 xi_1)."""
    s, bt = c3
    B, H, nn, nobs = bt.x_init.shape[0], s.H, s.H * 5, bt.nobs
    margin = bt.margin_psg
    u_ = -s.alpha * (bt.ff + 10.0 * bt.noise[:, 0] / 2.0)            # PSGCFS_FANUC.m:109 at u = 0, iter_O = 1
    slv = gpu.CFSBatch(s, nobs, margin, mode="PSGCFS", max_batch=B)
    dist, _, grad = slv.linearize(bt.x_init, bt.obs)
    u_on, _, it_on, st_on = slv.qp(u_, np.zeros((B, nn)), bt.xR1, dist, grad, want_lambda=False)
    slv.debug_options(no_certificate=True)
    u_off, _, it_off, st_off = slv.qp(u_, np.zeros((B, nn)), bt.xR1, dist, grad, want_lambda=False)
    slv.close()
    np.testing.assert_array_equal(st_on, st_off)
    np.testing.assert_array_equal(u_on[st_on == 0], u_off[st_on == 0])
    hits = np.nonzero(it_on != it_off)[0]
    assert (st_on[hits] == 2).all()
    print(f"[c3 PSGCFS] certificate fired on {hits.size} of {int((st_on == 2).sum())} infeasible first projections ({B} problems)")
    assert hits.size >= 20
    from types import SimpleNamespace
    import concurrent.futures as cf

    def oracle_status(b):
        s2 = SimpleNamespace(**vars(s))
        s2.xR1, s2.robot = bt.xR1[b], O.robotproperty2("M200i")
        A, rhs, od, _, og = O.get_con("M200i", s2, oracle_obs(bt, b, margin), bt.x_init[b], np.zeros(nn), mode="PSGCFS")
        assert np.abs(od - dist[b]).max() < 1e-13 and np.abs(og - grad[b]).max() < 2e-9
        return O.qp_solve(np.eye(nn), -u_[b], A, rhs)[3]

    with cf.ThreadPoolExecutor(16) as ex:
        sts = list(ex.map(oracle_status, hits.tolist()))
    bad = [int(b) for b, st in zip(hits, sts) if st != 2]
    assert not bad, bad


@pytest.mark.parametrize("tag", ["c3", "c4"])
def test_every_certificate_hit_is_infeasible_for_the_oracle(gpu, O, c3, c4, tag):
    """The first linearisation of every problem through cfs_qp with and without the certificate: wherever it changes the
    step count it returned QP_INFEASIBLE, the certificate-free device run proves the same by dual steps, and the oracle's
    dense QP on the SAME dist / grad is infeasible as well."""
    s, bt = c3 if tag == "c3" else c4
    B, H, nn, nobs = bt.x_init.shape[0], s.H, s.H * 5, bt.nobs
    margin = bt.margin_cfs
    slv = gpu.CFSBatch(s, nobs, margin, mode="CFS", max_batch=B)
    dist, _, grad = slv.linearize(bt.x_init, bt.obs)
    u_on, _, it_on, st_on = slv.qp(bt.ff, np.zeros((B, nn)), bt.xR1, dist, grad, want_lambda=False)
    slv.debug_options(no_certificate=True)
    u_off, _, it_off, st_off = slv.qp(bt.ff, np.zeros((B, nn)), bt.xR1, dist, grad, want_lambda=False)
    slv.close()
    np.testing.assert_array_equal(st_on, st_off)
    np.testing.assert_array_equal(u_on[st_on == 0], u_off[st_on == 0])
    hits = np.nonzero(it_on != it_off)[0]
    assert (st_on[hits] == 2).all()
    print(f"[{tag}] certificate fired on {hits.size} of {int((st_on == 2).sum())} infeasible first linearisations ({B} problems)")
    if tag == "c3":
        assert hits.size >= 50
    from types import SimpleNamespace
    import concurrent.futures as cf

    def oracle_status(b):
        s2 = SimpleNamespace(**vars(s))
        s2.xR1, s2.robot = bt.xR1[b], O.robotproperty2("M200i")
        A, rhs, od, _, og = O.get_con("M200i", s2, oracle_obs(bt, b, margin), bt.x_init[b], np.zeros(nn), mode="CFS")
        assert np.abs(od - dist[b]).max() < 1e-13 and np.abs(og - grad[b]).max() < 2e-9
        # the device's own linearisation, so that the verdict is about the same rows: collision rows rebuilt from dist / grad
        A2, rhs2 = A.copy(), rhs.copy()
        per = 1 + 2 * 5
        dt = s.robot.delta_t
        for j in range(nobs):
            for i in range(H):
                r = (j * H + i) * per
                coef = np.where(np.arange(H) <= i, ((i - np.arange(H)) + 0.5) * dt * dt, 0.0)
                A2[r] = -(coef[:, None] * grad[b, j, i][None, :]).reshape(-1)
                rhs2[r] = dist[b, j, i] - margin[j]
        assert np.abs(A2 - A).max() < 1e-8 and np.abs(rhs2 - rhs).max() < 1e-12
        A2 = np.vstack([A2, np.eye(nn), -np.eye(nn)])
        rhs2 = np.concatenate([rhs2, s.MAX_input, s.MAX_input])
        return O.qp_solve(s.QQ, bt.ff[b], A2, rhs2)[3]

    with cf.ThreadPoolExecutor(16) as ex:                            # the C oracle releases the GIL
        sts = list(ex.map(oracle_status, hits.tolist()))
    bad = [int(b) for b, st in zip(hits, sts) if st != 2]
    assert not bad, bad
