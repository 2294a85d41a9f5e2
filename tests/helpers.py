"""Shared checkers for the GPU parity tests (test infrastructure: uses the CPU oracle).

* ``truth_on_active_set``  -- solution of one QP on a given active set in extended precision (np.longdouble Gaussian
  elimination of the KKT system): the arbiter when the fp64 oracle and the device disagree beyond 1e-9.
* ``kkt_certificate``      -- stationarity / primal / dual / complementarity residuals of a DEVICE answer computed from the
  device's own multipliers (quadprog is closed source, so a self-certificate is the strongest evidence there is:
  the QPs are strictly convex, a KKT point is THE minimiser).
* ``chaotic_problems``     -- problems on which the reference algorithm itself amplifies a 1e-12 perturbation of x_init
  beyond 1e-6 rad (or flips status / iteration count): no implementation can match another one there.
"""
import numpy as np

LD = np.longdouble


def ld_solve(M, r):
    M = M.astype(LD).copy()
    r = r.astype(LD).copy()
    n = M.shape[0]
    for k in range(n):
        p = k + int(np.argmax(np.abs(M[k:, k])))
        if p != k:
            M[[k, p]] = M[[p, k]]
            r[[k, p]] = r[[p, k]]
        f = M[k + 1:, k] / M[k, k]
        M[k + 1:, k:] -= f[:, None] * M[k, k:][None]
        r[k + 1:] -= f * r[k]
    x = np.zeros(n, LD)
    for k in range(n - 1, -1, -1):
        x[k] = (r[k] - M[k, k + 1:] @ x[k + 1:]) / M[k, k]
    return x


def truth_on_active_set(G, g0, A, b, act):
    """min 1/2 x'Gx + g0'x s.t. A[act] x = b[act], in extended precision.  Returns (x, multipliers) as float64."""
    n, q = G.shape[0], len(act)
    M = np.zeros((n + q, n + q), LD)
    M[:n, :n] = G
    M[:n, n:] = A[act].T
    M[n:, :n] = A[act]
    sol = ld_solve(M, np.concatenate([-g0, b[act]]))
    return sol[:n].astype(float), sol[n:].astype(float)


def device_lambda_to_rows(lam, nobs, H, nj, with_bounds):
    """Device multiplier order [collision (j,i) | vel+ (i,c) | vel- (i,c) | bound+ | bound-] -> the reference's dense row
    order (per (j,i): 1 collision, nj +vel, nj -vel; CFS_FANUC.m:123-129) + [I; -I] bound rows.  The velocity rows, which
    get_con repeats for every obstacle, carry their multiplier on obstacle 0's copy."""
    B, nn, per = lam.shape[0], H * nj, 1 + 2 * nj
    rows = nobs * H * per
    out = np.zeros((B, rows + (2 * nn if with_bounds else 0)))
    out[:, 0:rows:per] = lam[:, :nobs * H]
    vp = lam[:, nobs * H:nobs * H + nn].reshape(B, H, nj)
    vm = lam[:, nobs * H + nn:nobs * H + 2 * nn].reshape(B, H, nj)
    for c in range(nj):
        out[:, 1 + c:H * per:per] = vp[:, :, c]
        out[:, 1 + nj + c:H * per:per] = vm[:, :, c]
    if with_bounds:
        out[:, rows:rows + nn] = lam[:, nobs * H + 2 * nn:nobs * H + 3 * nn]
        out[:, rows + nn:] = lam[:, nobs * H + 3 * nn:]
    return out


def kkt_certificate(G, g0, A, b, x, lam):
    """Batched.  G (n,n); g0 (B,n); A (B,m,n); b (B,m); x (B,n); lam (B,m) >= 0 multipliers of A x <= b.
    Returns (stationarity, primal, dual, complementarity), each (B,), all relative:
      stationarity    |Gx + g0 + A'lam|_inf / max(|g0|_inf, |Gx|_inf, 1)
      primal          max_i (a_i'x - b_i) / (1 + |b_i|)            (positive = violated)
      dual            max_i (-lam_i) / max(lam_max, 1)
      complementarity max_i lam_i |b_i - a_i'x| / ((1 + |b_i|) max(lam_max, 1))"""
    Gx = x @ G.T
    r = Gx + g0 + np.einsum("bmn,bm->bn", A, lam)
    sc = np.maximum(np.maximum(np.abs(g0).max(axis=1), np.abs(Gx).max(axis=1)), 1.0)
    sl = b - np.einsum("bmn,bn->bm", A, x)
    lmax = np.maximum(lam.max(axis=1), 1.0)
    stat = np.abs(r).max(axis=1) / sc
    prim = (-sl / (1.0 + np.abs(b))).max(axis=1)
    dual = (-lam).max(axis=1) / lmax
    comp = (lam * np.abs(sl) / (1.0 + np.abs(b))).max(axis=1) / lmax
    return stat, prim, dual, comp


def chaotic_problems(O, s, bt, mode, w0, K=None, reps=3, kick=1e-12, move=1e-6, seed=1):
    """Boolean mask over the batch: the ORACLE's own answer moves by more than `move` rad (or changes status / iteration
    count) when its x_init is perturbed by `kick` (N(0, kick^2) per entry, `reps` draws)."""
    margin = bt.margin_cfs if mode == "CFS" else bt.margin_psg
    nz = bt.noise if (mode == "PSGCFS" and bt.noise is not None) else None
    rng = np.random.default_rng(seed)
    B = bt.x_init.shape[0]
    mv, flip = np.zeros(B), np.zeros(B, bool)
    for _ in range(reps):
        w1 = O.optimizer_batch(O.robotproperty2("M200i"), mode, s.H, 5, bt.x_init + kick * rng.standard_normal(bt.x_init.shape),
                               bt.xR1, s.QQ, bt.ff, bt.caug, s.Aaug, s.Baug, s.lim, s.MAX_input, bt.obs, margin, s.epsilon_O,
                               s.MAX_O_ITER if K is None else K, s.alpha, noise=nz, nthreads=0)
        mv = np.maximum(mv, np.abs(w1.x_ - w0.x_).max(axis=1))
        flip |= (w1.status != w0.status) | (w1.iter_O != w0.iter_O)
    return (mv > move) | flip, mv


def oracle_obs(bt, b, margin):
    """obs cell of problem b in the oracle's form"""
    return [dict(l=np.stack([bt.obs[b, j, :3], bt.obs[b, j, 3:]], axis=1), epsilon=margin[j], D=margin[j]) for j in range(bt.obs.shape[1])]


def oracle_one_step(O, s, bt, mode, b, k, u_prev, margin, noise_row=None, robot_name="M200i"):
    """ONE outer iteration of the ORACLE (get_con + QP of Lib/CFS_FANUC.m:66-72 | the PSG step + projection of
    Lib/PSGCFS_FANUC.m:86-103), number k (= iter_O, 1-based), started from a GIVEN iterate u_prev (the u after iteration
    k-1; ignored for k = 1, where u = 0 and x_ = x_init as the constructors leave them).  Returns (u_k, status) with status
    0 = solved, otherwise the oracle's QP failed (2 = infeasible).  This is what makes iterations >= 2 of a chaotic problem
    checkable: the device's own previous iterate is the starting point, so only ONE iteration's amplification is in play."""
    from types import SimpleNamespace
    H, nj = s.H, 5
    nn = H * nj
    s2 = SimpleNamespace(**vars(s))
    s2.xR1, s2.robot = bt.xR1[b], O.robotproperty2(robot_name)
    if k == 1:
        u, x_ = np.zeros(nn), bt.x_init[b]
    else:
        u = np.asarray(u_prev, float)
        x_ = O.rollout(H, nj, s.robot.delta_t, bt.xR1[b], u)
    A, rhs, *_ = O.get_con(robot_name, s2, oracle_obs(bt, b, margin), x_, u, mode=mode)
    if mode == "CFS":
        G, g0 = s.QQ, bt.ff[b]
        A = np.vstack([A, np.eye(nn), -np.eye(nn)])
        rhs = np.concatenate([rhs, s.MAX_input, s.MAX_input])
    else:
        nz = np.zeros(nn) if noise_row is None else noise_row
        u_ = u - s.alpha * ((s.QQ @ u + bt.ff[b]) + 10.0 * nz / (float(k) * float(k) + 1.0))     # PSGCFS_FANUC.m:109
        G, g0 = np.eye(nn), -u_
    x, _, _, st, _ = O.qp_solve(G, g0, A, rhs)
    return x, st
