"""The problems helpers.chaotic_problems sets aside are NOT left unchecked: every outer iteration of every one of them is
compared with ONE iteration of the oracle started from the DEVICE's own previous iterate.

Why this closes the gap.  On a chaotic problem the end-to-end comparison is meaningless -- the oracle itself moves by
> 1e-6 rad under a 1e-12 kick of x_init -- but that amplification builds up over the 5-20 outer iterations.  With the
device's iterate u_{k-1} as the starting point only ONE iteration's amplification is in play, so a solver defect that shows
up at iteration >= 2 (where warm starts, the certificate and the drift projection first run) cannot hide behind the chaos:
u_k(device) must equal oracle_step(u_{k-1}(device)) to 1e-8 of |u_k| (1e-9 at the first iteration) wherever the oracle's own
single step is not itself kinked.  "Kinked" is again decided by the ORACLE alone: its single step moves by more than 1e-9 of |u_k| when u_{k-1} is
kicked by N(0, 1e-12^2) (a min-over-links switch, a clamp of distLinSeg or the near-zero surrogate of
dist_arm_3D_200i_2.m:22-24 crossed inside the finite-difference stencil: amplification > 1e3 in ONE iteration -- these steps
are what makes the problem chaotic, 8-50 % of its iterations).  On a kinked step the device must still be as close to the
oracle as the oracle is to its kicked self (within 1e3 x that move).  Measured (MI355X, round 3): un-kinked steps max 2.7e-9 /
1.3e-9 / 4.7e-9 (config 3 CFS / PSGCFS / config-4 shape), first iterations 5.5e-11 / 0 / 6.8e-10.

The u log comes from cfs_debug_log_u (both solvers; the solve itself is unchanged: asserted bit for bit).
"""
import concurrent.futures as cf

import numpy as np
import pytest

from helpers import oracle_one_step

pytestmark = pytest.mark.gpu

ONE_STEP_TOL = 1e-8       # |u_k(device) - oracle_step(u_{k-1}(device))|_inf / |u_k|_inf on un-kinked steps
FIRST_STEP_TOL = 1e-9     # the same at iteration 1 (identical linearisation point, no warm start, no amplification yet)
KINK = 1e-9               # the oracle's own single step moves by more than this (relative) under a 1e-12 kick


def _logged_solve(gpu, s, bt, mode, idx):
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    nz = bt.noise[idx] if (mode == "PSGCFS" and bt.noise is not None) else None
    n = len(idx)
    slv = gpu.CFSBatch(s, bt.nobs, margin, mode=mode, max_batch=n)
    plain = slv.solve(bt.x_init[idx], bt.xR1[idx], bt.ff[idx], bt.caug[idx], bt.obs[idx], noise=nz)
    slv.log_u(True)
    got = slv.solve(bt.x_init[idx], bt.xR1[idx], bt.ff[idx], bt.caug[idx], bt.obs[idx], noise=nz)
    ulog = slv.read_u_log(n)
    slv.close()
    for k in ("u", "x_", "status", "iter_O", "total_iter", "cost_all"):
        np.testing.assert_array_equal(getattr(plain, k), getattr(got, k), err_msg=k)     # logging changes nothing
    return got, ulog


def _check(gpu, O, s, bt, mode, chaotic, tag):
    idx = np.nonzero(chaotic)[0]
    assert idx.size > 0
    got, ulog = _logged_solve(gpu, s, bt, mode, idx)
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    rng = np.random.default_rng(7)
    jobs = []
    for a, b in enumerate(idx):
        n_it = int(got.iter_O[a]) - 1                               # completed outer iterations
        np.testing.assert_array_equal(ulog[a, n_it - 1] if n_it > 0 else np.zeros(s.H * 5), got.u[a])   # the last logged u is self.u
        rows = 0                                                    # PSGCFS: noise rows consumed so far (PSGCFS_FANUC.m:109)
        for k in range(1, n_it + (1 if got.status[a] >= 2 else 0) + 1):
            failed = k == n_it + 1                                  # the iteration whose QP the device reported infeasible
            u_prev = ulog[a, k - 2] if k >= 2 else None
            u_k = None if failed else ulog[a, k - 1]
            nz_row = None
            if mode == "PSGCFS":
                c = lambda j: (100000.0 if j < 0 else (bt.caug[b] if j == 0 else got.cost_all[a, j - 1]))   # noqa: E731  cost after iteration j
                if abs(c(k - 1) - c(k - 2)) < 1e-4:                 # stop_inner (PSGCFS_FANUC.m:136-142): no step, u stays
                    if not failed:
                        np.testing.assert_array_equal(u_k, u_prev if u_prev is not None else np.zeros_like(u_k))
                    continue
                nz_row = bt.noise[b, rows] if bt.noise is not None and rows < bt.noise.shape[1] else None
                rows += 1
            jobs.append((a, int(b), k, u_prev, u_k, nz_row, failed, 1e-12 * rng.standard_normal(s.H * 5)))

    def one(job):
        a, b, k, u_prev, u_k, nz_row, failed, kick = job
        want, st = oracle_one_step(O, s, bt, mode, b, k, u_prev, margin, noise_row=nz_row)
        if k == 1:                                                   # the first step starts from x_init: kick that, as chaotic_problems does
            class _B:                                                # noqa: N801  a view of bt with problem b's x_init moved
                pass
            b2 = _B()
            b2.__dict__.update(vars(bt))
            b2.x_init = bt.x_init.copy()
            b2.x_init[b] = bt.x_init[b] + np.resize(kick, bt.x_init[b].shape)
            w2, st2 = oracle_one_step(O, s, b2, mode, b, k, None, margin, noise_row=nz_row)
        else:
            w2, st2 = oracle_one_step(O, s, bt, mode, b, k, u_prev + kick, margin, noise_row=nz_row)
        if failed:
            return (a, k, None, None, st, st2)
        sc = max(np.abs(u_k).max(), 1e-300)
        err = np.abs(u_k - want).max() / sc if st == 0 else np.inf
        sens = np.abs(w2 - want).max() / sc if (st == 0 and st2 == 0) else np.inf
        return (a, k, err, sens, st, st2)

    with cf.ThreadPoolExecutor(16) as ex:                            # the C oracle releases the GIL
        res = list(ex.map(one, jobs))
    steps = [r for r in res if r[2] is not None]
    err = np.array([r[2] for r in steps])
    sens = np.array([r[3] for r in steps])
    kink = ~(sens <= KINK)
    first = np.array([r[1] == 1 for r in steps])
    print(f"[{tag} {mode}] {idx.size} chaotic problems, {len(steps)} outer iterations checked one step at a time: "
          f"{int(kink.sum())} kinked (oracle's own step moves > {KINK:g} under a 1e-12 kick); un-kinked: median {np.median(err[~kink]):.1e}, "
          f"max {err[~kink].max():.1e}; first iterations: max {err[first & ~kink].max() if (first & ~kink).any() else 0:.1e}; "
          f"kinked steps: median err {np.median(err[kink]) if kink.any() else 0:.1e}")
    bad = [(int(idx[steps[i][0]]), steps[i][1], float(err[i]), float(sens[i])) for i in np.nonzero(~kink & ~(err <= ONE_STEP_TOL))[0]]
    assert not bad, bad
    bad1 = [(int(idx[steps[i][0]]), float(err[i])) for i in np.nonzero(first & ~kink & ~(err <= FIRST_STEP_TOL))[0]]
    assert not bad1, bad1
    assert (~kink).sum() >= 0.4 * len(steps)                          # most iterations, even of these problems, are ordinary
    # on a kinked step the device must still be AS CLOSE to the oracle as the oracle is to itself (within 1e3 x its own move)
    worse = [(int(idx[steps[i][0]]), steps[i][1], float(err[i]), float(sens[i])) for i in np.nonzero(kink & np.isfinite(sens) & ~(err <= np.maximum(ONE_STEP_TOL, 1e3 * sens)))[0]]
    assert len(worse) <= 0.02 * len(steps), worse
    # the QP the device reported infeasible: infeasible for the oracle from the same iterate too (or the step is kinked: the
    # kicked oracle disagrees with itself)
    for a, k, _, _, st, st2 in [r for r in res if r[2] is None]:
        assert st == 2 or st2 != st, (int(idx[a]), k, st, st2)


@pytest.mark.parametrize("mode", ["CFS", "PSGCFS"])
def test_config3_chaotic_problems_one_step_at_a_time(gpu, O, c3, c3_oracle, mode):
    s, bt = c3
    _, chaotic, _ = c3_oracle(mode)
    _check(gpu, O, s, bt, mode, chaotic, "config3")


def test_config4_chaotic_problems_one_step_at_a_time(gpu, O, c4, c4_oracle):
    s, bt = c4
    _, chaotic, _ = c4_oracle("CFS")
    _check(gpu, O, s, bt, "CFS", chaotic, "config4 shape")
