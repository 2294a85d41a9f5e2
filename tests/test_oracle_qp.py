"""The oracle's QP (stand-in for the closed-source quadprog) against an independent method (CPU)."""
import numpy as np
import scipy.linalg as sl
import scipy.optimize as so


def ldp_nnls(H, f, G, h):
    """Strictly convex QP by the Lawson-Hanson least-distance reduction + scipy NNLS.
    Returns None if infeasible."""
    n = G.shape[1]
    Hs = (H + H.T) / 2
    L = np.linalg.cholesky(Hs)
    q0 = np.linalg.solve(Hs, f)
    E = -sl.solve_triangular(L, G.T, lower=True).T
    e = -(h + G @ q0)
    M = np.vstack([E.T, e[None, :]])
    rhs = np.zeros(n + 1); rhs[-1] = 1
    y, rn = so.nnls(M, rhs, maxiter=50000)
    if rn < 1e-9:
        return None
    res = M @ y - rhs
    return sl.solve_triangular(L.T, -res[:n] / res[n]) - q0


def test_random_qps(O):
    rng = np.random.default_rng(3)
    for trial in range(25):
        n, m = int(rng.integers(3, 25)), int(rng.integers(1, 60))
        B = rng.normal(size=(n, n)); H = B @ B.T + np.eye(n)
        f = rng.normal(size=n) * 3
        G = rng.normal(size=(m, n)); h = rng.normal(size=m) + 0.5
        x, lam, it, st, kkt = O.qp_solve(H, f, G, h)
        ref = ldp_nnls(H, f, G, h)
        if ref is None:
            assert st == 2
            continue
        assert st == 0 and np.abs(x - ref).max() < 1e-8
        assert kkt[0] < 1e-10 and kkt[1] < 1e-10 and kkt[2] >= 0 and kkt[3] < 1e-9


def test_duplicate_rows_and_bounds(O):
    # the reference re-appends the velocity rows per obstacle; duplicates must not disturb the solver
    H, f = np.diag([2.0, 1.0]), np.array([-4.0, -4.0])
    G = np.array([[1.0, 1.0], [1.0, 1.0], [1.0, 0.0], [1.0, 0.0], [-1.0, 0.0]])
    h = np.array([2.0, 2.0, 1.5, 1.5, 0.0])
    x, lam, it, st, kkt = O.qp_solve(H, f, G, h)
    assert st == 0 and np.allclose(x, [2 / 3, 4 / 3]) and kkt[0] < 1e-12


def test_infeasible_is_reported(O):
    x, lam, it, st, kkt = O.qp_solve(np.eye(2), np.zeros(2), np.array([[1.0, 0], [-1.0, 0]]), np.array([-1.0, -1.0]))
    assert st == 2


def test_main_fanuc_first_qp_vs_nnls(O):
    P = O.problem_main_FANUC(); s = P.sys_info
    A, b, *_ = O.get_con(P.ROBOT, s, P.obs, s.x_, np.zeros(150))
    G = np.vstack([A, np.eye(150), -np.eye(150)])
    h = np.concatenate([b, s.MAX_input, s.MAX_input])
    x, lam, it, st, kkt = O.qp_solve(s.QQ, s.ff, G, h)
    ref = ldp_nnls(s.QQ, s.ff, G, h)
    assert st == 0 and np.abs(x - ref).max() < 1e-10 and (lam > 0).sum() == 3      # 3-5 active rows
    assert kkt[0] < 1e-12 and kkt[1] < 1e-12 and kkt[3] < 1e-9
