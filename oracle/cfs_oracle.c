/*
 * cfs_oracle.c -- CPU restatement (plain C, fp64) of the reference's CFS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under motionplanning_5d_m_amd/ may import,
 * link or execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and there only as the checker / the reported CPU
 * baseline.
 *
 * PARITY STATUS: "parity unpinned" at the QP boundary.  The reference is MATLAB
 * source only (no MATLAB/Octave here) and solves its QPs with MathWorks
 * `quadprog` (Optimization Toolbox, proprietary, version unrecorded; call sites
 * Lib/CFS_FANUC.m:85, Lib/PSGCFS_FANUC.m:120).  The reference ships no tests,
 * golden vectors or fixtures for this path; the single known answer in the
 * repository is the doc-comment example of Lib/functions/distLinSeg.m:15-18,
 * which tests/test_oracle_geometry.py checks.  Two stored OUTPUTS of the reference's
 * MATLAB runs were found among its data files in round 3 and pin two functions of
 * this file (tests/test_oracle_golden.py): figure/M16iBCapsules.mat:RoCap is CapPos's
 * result for the M16iB (orc_arm_pos: 7.5e-16 m on 11 of 12 end points, the 12th a
 * constant edited since), data/good_xori.mat + data/M16_ref_2.mat a stored
 * (u -> x_) rollout (orc_rollout: all 240 doubles bit for bit).  Everything else --
 * distLinSeg beyond its example, the near-zero surrogate, num_jac, get_con, the QP
 * and the outer loops -- has no reference-held output: parity unpinned there.
 * The QPs are strictly convex, so
 * the minimiser is unique; every solve here is certified by its KKT residuals
 * (orc_qp_kkt) and cross-checked in tests against an independent Lawson-Hanson
 * least-distance solve (scipy.optimize.nnls).
 *
 * Every function cites the reference file:line it restates.  All matrices are
 * column-major, exactly as MATLAB hands them.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXLINK 8

/* robot kinds = which dist_arm_* the class constructor selects
 * (Lib/CFS_FANUC.m:49-54, Lib/PSGCFS_FANUC.m:52-57) */
enum { ORC_ROBOT_M16IB = 0, ORC_ROBOT_M200I = 1, ORC_ROBOT_2L = 2 };

typedef struct {
    int kind;
    int nlink;                 /* rows of robot.DH / entries of robot.cap           */
    double DH[ORC_MAXLINK * 4];/* DH[i*4 + c], c = theta,d,a,alpha (robotproperty2.m:24-29) */
    double base[3];            /* robot.base                                        */
    double cap[ORC_MAXLINK * 6];/* cap[i*6 + k*3 + r] = robot.cap{i+1}.p(r+1,k+1)   */
    double T[9];               /* 2L only: robot.T column-major (robotproperty2.m:117-119) */
} orc_robot;

/* status codes of the optimiser loops */
enum { ORC_OK_CONVERGED = 0, ORC_OK_MAXITER = 1, ORC_QP_INFEASIBLE = 2, ORC_NUMERIC = 3 };

/* ------------------------------------------------------------------------- */
/* a1  CapPos  (Lib/functions/CapPos.m:8-22)                                   */
/* pos[i*6 + k*3 + r] = pos{i+1}.p(r+1,k+1)                                    */
/* ------------------------------------------------------------------------- */
static void mat4_mul(const double *A, const double *B, double *C)
{   /* 4x4 row-major product, plain triple loop in MATLAB's i,j,k sense */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
            C[i * 4 + j] = s;
        }
}

void orc_cap_pos(const double *base, const double *DH, const double *cap, int nlink, double *pos)
{
    double M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; /* M{1}=eye(4) :11 */
    for (int i = 0; i < nlink; ++i) {
        double th = DH[i * 4 + 0], d = DH[i * 4 + 1], a = DH[i * 4 + 2], al = DH[i * 4 + 3];
        double ct = cos(th), st = sin(th), ca = cos(al), sa = sin(al);
        double Tm[16] = {ct, -st * ca, st * sa, a * ct,      /* R :13-15, T :16 */
                         st, ct * ca, -ct * sa, a * st,
                         0, sa, ca, d,
                         0, 0, 0, 1};
        double Mn[16];
        mat4_mul(M, Tm, Mn);                                 /* :17 */
        memcpy(M, Mn, sizeof M);
        for (int k = 0; k < 2; ++k) {                        /* :18-20 */
            const double *p = cap + i * 6 + k * 3;
            for (int r = 0; r < 3; ++r) {
                double s = 0.0;
                for (int c = 0; c < 3; ++c) s += M[r * 4 + c] * p[c];
                pos[i * 6 + k * 3 + r] = s + M[r * 4 + 3] + base[r];
            }
        }
    }
}

/* a3''  CapPos2 (Lib/2L/CapPos2.m:1-31): M{i}=M{i-1}*[Rz(theta(i-1)) T(:,i)] */
void orc_cap_pos2(const double *theta, const double *base, const double *T, const double *cap,
                  int nlink, double *pos)
{
    double M[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 1; i <= nlink; ++i) {
        double ct = cos(theta[i - 1]), st = sin(theta[i - 1]);
        const double *Ti = T + i * 3;                       /* T(:,i) with i = 2..nlink+1 (1-based) */
        double Tm[16] = {ct, -st, 0, Ti[0],
                         st, ct, 0, Ti[1],
                         0, 0, 1, Ti[2],
                         0, 0, 0, 1};
        double Mn[16];
        mat4_mul(M, Tm, Mn);
        memcpy(M, Mn, sizeof M);
        for (int k = 0; k < 2; ++k) {
            const double *p = cap + (i - 1) * 6 + k * 3;
            for (int r = 0; r < 3; ++r) {
                double s = 0.0;
                for (int c = 0; c < 3; ++c) s += M[r * 4 + c] * p[c];
                pos[(i - 1) * 6 + k * 3 + r] = s + M[r * 4 + 3] + base[r];
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* a2  distLinSeg (Lib/functions/distLinSeg.m:23-101), 3-D points              */
/* points[0..2] = point1s+d1*t, points[3..5] = point2s+d2*u  (:88)            */
/* ------------------------------------------------------------------------- */
static double fixbound(double num)
{   /* :93-101 */
    if (num < 0) num = 0;
    else if (num > 1) num = 1;
    return num;
}

double orc_dist_lin_seg(const double *p1s, const double *p1e, const double *p2s, const double *p2e,
                        double *points)
{
    double d1[3], d2[3], d12[3];
    for (int r = 0; r < 3; ++r) {
        d1[r] = p1e[r] - p1s[r];
        d2[r] = p2e[r] - p2s[r];
        d12[r] = p2s[r] - p1s[r];
    }
    double D1 = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
    double D2 = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
    double S1 = d1[0] * d12[0] + d1[1] * d12[1] + d1[2] * d12[2];
    double S2 = d2[0] * d12[0] + d2[1] * d12[1] + d2[2] * d12[2];
    double R = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
    double den = D1 * D2 - R * R;
    double t, u;
    if (D1 == 0 || D2 == 0) {
        if (D1 != 0) { u = 0; t = fixbound(S1 / D1); }
        else if (D2 != 0) { t = 0; u = fixbound(-S2 / D2); }
        else { t = 0; u = 0; }
    } else if (den == 0) {
        t = 0;
        u = -S2 / D2;
        double uf = fixbound(u);
        if (uf != u) { t = fixbound((uf * R + S1) / D1); u = uf; }
    } else {
        t = fixbound((S1 * D2 - S2 * R) / den);
        u = (t * R - S2) / D2;
        double uf = fixbound(u);
        if (uf != u) { t = fixbound((uf * R + S1) / D1); u = uf; }
    }
    double e[3];
    for (int r = 0; r < 3; ++r) e[r] = d1[r] * t - d2[r] * u - d12[r];
    if (points) {
        for (int r = 0; r < 3; ++r) {
            points[r] = p1s[r] + d1[r] * t;
            points[3 + r] = p2s[r] + d2[r] * u;
        }
    }
    return sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
}

/* ------------------------------------------------------------------------- */
/* a3 / a3' / a3''  dist_arm_*  (Lib/200i/dist_arm_3D_200i_2.m:1-30,            */
/*   Lib/M16iB/dist_arm_3D_Heu_2.m:1-30, Lib/2L/dist_arm_2L.m:1-23)             */
/* theta: nj x 1; obs: 3x2 column-major = [l(:,1) l(:,2)].                     */
/* M16iB's near-zero branch is ill-formed in the reference (6x1 minus 3x1,     */
/* dist_arm_3D_Heu_2.m:23); it is restated with points(1:3), as 200i_2.m:23.  */
/* Returns d; *linkid is 1-based, first minimum wins (strict <).               */
/* ------------------------------------------------------------------------- */
double orc_mesh_seg_distance(int mesh_id, const double *seg, double *points, int *tri_id);   /* mesh_oracle.c */

double orc_dist_arm(const orc_robot *rb, const double *theta, int nj, const double *obs, int *linkid)
{
    double pos[ORC_MAXLINK * 6];
    if (rb->kind == ORC_ROBOT_2L) {
        orc_cap_pos2(theta, rb->base, rb->T, rb->cap, nj, pos);
    } else {
        double DH[ORC_MAXLINK * 4];
        memcpy(DH, rb->DH, sizeof(double) * 4 * nj);          /* DH = robot.DH(1:nstate,:) */
        for (int i = 0; i < nj; ++i) DH[i * 4] = theta[i];
        if (rb->kind == ORC_ROBOT_M200I) DH[1 * 4] = DH[1 * 4] - M_PI / 2; /* 200i_2.m:11 */
        orc_cap_pos(rb->base, DH, rb->cap, nj, pos);
    }
    double d = INFINITY;
    int id = 0;
    /* Mesh obstacle (row f3; M200i/dist_arm_surf_200i.m:20-28 with the build's own point2surface_dis, see
     * mesh_oracle.c): flagged by obs[0] = NaN, obs[1] = id given to orc_mesh_register. */
    const int is_mesh = isnan(obs[0]);
    for (int i = 0; i < nj; ++i) {
        double pts[6];
        double dis = is_mesh ? orc_mesh_seg_distance((int)obs[1], pos + i * 6, pts, 0)
                             : orc_dist_lin_seg(pos + i * 6, pos + i * 6 + 3, obs, obs + 3, pts);
        if (fabs(dis) < 0.0001) {                            /* :22-24 */
            double e0 = pts[0] - pos[i * 6 + 3], e1 = pts[1] - pos[i * 6 + 4], e2 = pts[2] - pos[i * 6 + 5];
            dis = -sqrt(e0 * e0 + e1 * e1 + e2 * e2);
        }
        if (dis < d) { d = dis; id = i + 1; }               /* :25-28 */
    }
    if (linkid) *linkid = id;
    return d;
}

/* FK end points only (for tests / fixtures): pos[nj*6] as orc_cap_pos */
void orc_arm_pos(const orc_robot *rb, const double *theta, int nj, double *pos)
{
    if (rb->kind == ORC_ROBOT_2L) {
        orc_cap_pos2(theta, rb->base, rb->T, rb->cap, nj, pos);
    } else {
        double DH[ORC_MAXLINK * 4];
        memcpy(DH, rb->DH, sizeof(double) * 4 * nj);
        for (int i = 0; i < nj; ++i) DH[i * 4] = theta[i];
        if (rb->kind == ORC_ROBOT_M200I) DH[1 * 4] = DH[1 * 4] - M_PI / 2;
        orc_cap_pos(rb->base, DH, rb->cap, nj, pos);
    }
}

/* ------------------------------------------------------------------------- */
/* a4  num_jac (Lib/functions/num_jac.m:1-17), f = dist_arm first output.      */
/* LITERAL: xp is copied once and xp(i) is left at x(i)-eps/2 afterwards.      */
/* ------------------------------------------------------------------------- */
void orc_num_jac_dist(const orc_robot *rb, const double *x, int nj, const double *obs, double *grad)
{
    const double eps = 1e-5;
    double xp[ORC_MAXLINK];
    memcpy(xp, x, sizeof(double) * nj);
    for (int i = 0; i < nj; ++i) {
        xp[i] = x[i] + eps / 2;
        double yhi = orc_dist_arm(rb, xp, nj, obs, 0);
        xp[i] = x[i] - eps / 2;
        double ylo = orc_dist_arm(rb, xp, nj, obs, 0);
        grad[i] = (yhi - ylo) / eps;
    }
}

/* ------------------------------------------------------------------------- */
/* a5  get_con (Lib/CFS_FANUC.m:101-135, Lib/PSGCFS_FANUC.m:145-184)           */
/* Dense, literal row order: per obstacle j, per waypoint i:                   */
/*   1 collision row, nj "+vel" rows, nj "-vel" rows.                          */
/* Ainq: rows x nn column-major, rows = nobs*H*(1+2nj).  Also returns the      */
/* compact per-(j,i) distance, link id and gradient for kernel-level parity.   */
/* ------------------------------------------------------------------------- */
void orc_get_con(const orc_robot *rb, int H, int nj, int ns, const double *x_, const double *u,
                 const double *Baug, const double *Aaug, const double *xR1, const double *lim,
                 int nobs, const double *obs, const double *margin,
                 double *Ainq, double *binq, double *dist, int *linkid, double *grad)
{
    const int nn = H * nj; /* nu == nj on this path */
    const int nrow_tot = H * ns;
    const int rows = nobs * H * (1 + 2 * nj);
    int row = 0;
    for (int j = 0; j < nobs; ++j) {
        const double *ol = obs + j * 6;
        for (int i = 0; i < H; ++i) {
            const double *theta = x_ + ns * i;               /* :114 */
            int lid;
            double d = orc_dist_arm(rb, theta, nj, ol, &lid); /* :115 */
            double g[ORC_MAXLINK];
            orc_num_jac_dist(rb, theta, nj, ol, g);          /* :118 */
            if (dist) dist[j * H + i] = d;
            if (linkid) linkid[j * H + i] = lid;
            if (grad) memcpy(grad + (j * H + i) * nj, g, sizeof(double) * nj);
            /* l = -Diff'*Bj(1:nj,:) ; s = (d-margin) - Diff'*Bj(1:nj,:)*u   :119-121 */
            double gBu = 0.0;
            for (int c = 0; c < nn; ++c) {
                double gb = 0.0;
                for (int r = 0; r < nj; ++r) gb += g[r] * Baug[(i * ns + r) + (size_t)c * nrow_tot];
                if (Ainq) Ainq[row + (size_t)c * rows] = -gb;
                gBu += gb * u[c];
            }
            binq[row] = (d - margin[j]) - gBu;
            ++row;
            /* velocity rows :126-129 */
            for (int sgn = 0; sgn < 2; ++sgn) {
                for (int r = 0; r < nj; ++r) {
                    int br = i * ns + nj + r;
                    double ax = 0.0;
                    for (int c = 0; c < ns; ++c) ax += Aaug[br + (size_t)c * nrow_tot] * xR1[c];
                    if (Ainq)
                        for (int c = 0; c < nn; ++c) {
                            double b = Baug[br + (size_t)c * nrow_tot];
                            Ainq[row + (size_t)c * rows] = sgn == 0 ? b : -b;
                        }
                    binq[row] = sgn == 0 ? lim[r] - ax : lim[r] + ax;
                    ++row;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* a6/a7  strictly convex QP:  min 1/2 x'Gx + g0'x  s.t.  A x <= b             */
/* (bounds are passed as rows by the caller).  Stands in for quadprog          */
/* (Lib/CFS_FANUC.m:85, Lib/PSGCFS_FANUC.m:120, closed source): dense          */
/* Goldfarb-Idnani dual active set (Math. Prog. 27 (1983) 1-33).               */
/* A: m x n column-major.  lambda: m multipliers (>=0).                        */
/* returns 0 ok, 2 infeasible, 3 numeric trouble.  *iters = active-set steps.  */
/* ------------------------------------------------------------------------- */
static int chol_lower(double *L, int n)
{   /* in place, column-major, lower */
    for (int j = 0; j < n; ++j) {
        double s = L[j + (size_t)j * n];
        for (int k = 0; k < j; ++k) s -= L[j + (size_t)k * n] * L[j + (size_t)k * n];
        if (!(s > 0)) return -1;
        double ljj = sqrt(s);
        L[j + (size_t)j * n] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double t = L[i + (size_t)j * n];
            for (int k = 0; k < j; ++k) t -= L[i + (size_t)k * n] * L[j + (size_t)k * n];
            L[i + (size_t)j * n] = t / ljj;
        }
    }
    return 0;
}

int orc_qp_solve(int n, const double *G, const double *g0, int m, const double *A, const double *b,
                 double *x, double *lambda, int *iters)
{
    int status = 0;
    double *L = (double *)malloc(sizeof(double) * n * n);
    double *J = (double *)calloc((size_t)n * n, sizeof(double));
    double *R = (double *)calloc((size_t)n * n, sizeof(double));
    double *d = (double *)malloc(sizeof(double) * n);
    double *z = (double *)malloc(sizeof(double) * n);
    double *r = (double *)malloc(sizeof(double) * n);
    double *np = (double *)malloc(sizeof(double) * n);
    double *uact = (double *)malloc(sizeof(double) * (n + 1));
    int *act = (int *)malloc(sizeof(int) * (n + 1));
    char *isact = (char *)calloc(m > 0 ? m : 1, 1);
    int q = 0, it = 0;
    /* symmetrise (quadprog does so silently; QQ is asymmetric at 1e-11) */
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) L[i + (size_t)j * n] = 0.5 * (G[i + (size_t)j * n] + G[j + (size_t)i * n]);
    if (chol_lower(L, n)) { status = 3; goto done; }
    /* x = -G^{-1} g0 */
    for (int i = 0; i < n; ++i) {
        double s = -g0[i];
        for (int k = 0; k < i; ++k) s -= L[i + (size_t)k * n] * z[k];
        z[i] = s / L[i + (size_t)i * n];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = z[i];
        for (int k = i + 1; k < n; ++k) s -= L[k + (size_t)i * n] * x[k];
        x[i] = s / L[i + (size_t)i * n];
    }
    /* J = L^{-T}: column j of J solves L' J(:,j) = e_j */
    for (int j = 0; j < n; ++j) {
        for (int i = n - 1; i >= 0; --i) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = i + 1; k < n; ++k) s -= L[k + (size_t)i * n] * J[k + (size_t)j * n];
            J[i + (size_t)j * n] = s / L[i + (size_t)i * n];
        }
    }
    for (int i = 0; i < m; ++i) lambda[i] = 0.0;
    const int maxit = 20 * (m + n) + 100;
    for (;;) {
        /* step 1: most violated constraint; s_i = b_i - a_i'x */
        int p = -1;
        double smin = 0.0;
        for (int i = 0; i < m; ++i) {
            if (isact[i]) continue;
            double ax = 0.0;
            for (int c = 0; c < n; ++c) ax += A[i + (size_t)c * m] * x[c];
            double s = b[i] - ax;
            double tol = 1e-11 * (1.0 + fabs(b[i]));
            if (s < -tol && s < smin) { smin = s; p = i; }
        }
        if (p < 0) break;
        for (int c = 0; c < n; ++c) np[c] = -A[p + (size_t)c * m]; /* n_p'x >= -b_p */
        double up = 0.0;
        double sp = smin;
        for (;;) {
            if (++it > maxit) { status = 3; goto done; }
            /* step 2a: d = J'n_p ; z = J2 d2 ; r = R^{-1} d1 */
            double dn2 = 0.0, d2n2 = 0.0;
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int k = 0; k < n; ++k) s += J[k + (size_t)j * n] * np[k];
                d[j] = s;
                dn2 += s * s;
                if (j >= q) d2n2 += s * s;
            }
            for (int k = 0; k < n; ++k) {
                double s = 0.0;
                for (int j = q; j < n; ++j) s += J[k + (size_t)j * n] * d[j];
                z[k] = s;
            }
            for (int i = q - 1; i >= 0; --i) {
                double s = d[i];
                for (int k = i + 1; k < q; ++k) s -= R[i + (size_t)k * n] * r[k];
                r[i] = s / R[i + (size_t)i * n];
            }
            /* step 2b */
            int dependent = !(d2n2 > 1e-18 * dn2);
            double t1 = INFINITY, t2 = INFINITY;
            int l = -1;
            for (int k = 0; k < q; ++k)
                if (r[k] > 0) {
                    double tt = uact[k] / r[k];
                    if (tt < t1) { t1 = tt; l = k; }
                }
            if (!dependent) {
                double znp = 0.0;
                for (int k = 0; k < n; ++k) znp += z[k] * np[k];
                t2 = -sp / znp;
            }
            double t = t1 < t2 ? t1 : t2;
            if (!(t < INFINITY)) { status = 2; goto done; }   /* infeasible */
            int full = (t2 <= t1);
            if (!dependent) for (int k = 0; k < n; ++k) x[k] += t * z[k];
            for (int k = 0; k < q; ++k) uact[k] -= t * r[k];
            up += t;
            if (!dependent && full) {
                /* add p: Givens on d from the bottom up to position q */
                for (int i = n - 1; i > q; --i) {
                    double a1 = d[i - 1], a2 = d[i];
                    if (a2 == 0.0) continue;
                    double h = hypot(a1, a2);
                    double c = a1 / h, s = a2 / h;
                    d[i - 1] = h; d[i] = 0.0;
                    for (int k = 0; k < n; ++k) {
                        double j1 = J[k + (size_t)(i - 1) * n], j2 = J[k + (size_t)i * n];
                        J[k + (size_t)(i - 1) * n] = c * j1 + s * j2;
                        J[k + (size_t)i * n] = -s * j1 + c * j2;
                    }
                }
                for (int i = 0; i <= q; ++i) R[i + (size_t)q * n] = d[i];
                act[q] = p; uact[q] = up; isact[p] = 1; ++q;
                break;                                         /* back to step 1 */
            }
            /* drop active constraint at position l */
            {
                int gone = act[l];
                isact[gone] = 0;
                lambda[gone] = 0.0;
                for (int k = l; k < q - 1; ++k) {
                    act[k] = act[k + 1]; uact[k] = uact[k + 1];
                    for (int i = 0; i < n; ++i) R[i + (size_t)k * n] = R[i + (size_t)(k + 1) * n];
                }
                --q;
                for (int i = 0; i < n; ++i) R[i + (size_t)q * n] = 0.0;
                for (int k = l; k < q; ++k) {              /* restore triangular form */
                    double a1 = R[k + (size_t)k * n], a2 = R[k + 1 + (size_t)k * n];
                    if (a2 == 0.0) continue;
                    double h = hypot(a1, a2);
                    double c = a1 / h, s = a2 / h;
                    for (int cc = k; cc < q; ++cc) {
                        double r1 = R[k + (size_t)cc * n], r2 = R[k + 1 + (size_t)cc * n];
                        R[k + (size_t)cc * n] = c * r1 + s * r2;
                        R[k + 1 + (size_t)cc * n] = -s * r1 + c * r2;
                    }
                    for (int i = 0; i < n; ++i) {
                        double j1 = J[i + (size_t)k * n], j2 = J[i + (size_t)(k + 1) * n];
                        J[i + (size_t)k * n] = c * j1 + s * j2;
                        J[i + (size_t)(k + 1) * n] = -s * j1 + c * j2;
                    }
                }
            }
            /* recompute slack of p at the new x and repeat step 2 */
            {
                double ax = 0.0;
                for (int c = 0; c < n; ++c) ax += A[p + (size_t)c * m] * x[c];
                sp = b[p] - ax;
            }
        }
    }
    for (int k = 0; k < q; ++k) lambda[act[k]] = uact[k];
done:
    if (iters) *iters = it;
    free(L); free(J); free(R); free(d); free(z); free(r); free(np); free(uact); free(act); free(isact);
    return status;
}

/* KKT certificate: res[0]=stationarity (inf-norm of Gs x+g0+A'lambda, relative to
 * 1+|g0|_inf), res[1]=max primal violation, res[2]=min lambda, res[3]=max |lambda_i*slack_i| */
void orc_qp_kkt(int n, const double *G, const double *g0, int m, const double *A, const double *b,
                const double *x, const double *lambda, double *res)
{
    double stat = 0.0, g0n = 0.0, pv = 0.0, lmin = 0.0, comp = 0.0;
    for (int i = 0; i < n; ++i) {
        double s = g0[i];
        for (int k = 0; k < n; ++k) s += 0.5 * (G[i + (size_t)k * n] + G[k + (size_t)i * n]) * x[k];
        for (int k = 0; k < m; ++k) s += A[k + (size_t)i * m] * lambda[k];
        if (fabs(s) > stat) stat = fabs(s);
        if (fabs(g0[i]) > g0n) g0n = fabs(g0[i]);
    }
    for (int k = 0; k < m; ++k) {
        double ax = 0.0;
        for (int c = 0; c < n; ++c) ax += A[k + (size_t)c * m] * x[c];
        double sl = b[k] - ax;
        if (-sl > pv) pv = -sl;
        if (lambda[k] < lmin) lmin = lambda[k];
        if (fabs(lambda[k] * sl) > comp) comp = fabs(lambda[k] * sl);
    }
    res[0] = stat / (1.0 + g0n); res[1] = pv; res[2] = lmin; res[3] = comp;
}

/* ------------------------------------------------------------------------- */
/* rollout (Lib/CFS_FANUC.m:90-94): xR(:,i)=A10*xR(:,i-1)+B10*u_{i-1}          */
/* A10=[I dt*I;0 I], B10=[dt^2/2*I; dt*I] (robotproperty2.m:136-139)           */
/* ------------------------------------------------------------------------- */
void orc_rollout(int H, int nj, double dt, const double *xR1, const double *u, double *x_)
{
    double cur[2 * ORC_MAXLINK];
    memcpy(cur, xR1, sizeof(double) * 2 * nj);
    for (int i = 0; i < H; ++i) {
        double nxt[2 * ORC_MAXLINK];
        for (int r = 0; r < nj; ++r) {
            nxt[r] = (cur[r] + dt * cur[nj + r]) + (0.5 * dt * dt) * u[i * nj + r];
            nxt[nj + r] = cur[nj + r] + dt * u[i * nj + r];
        }
        memcpy(cur, nxt, sizeof(double) * 2 * nj);
        memcpy(x_ + i * 2 * nj, cur, sizeof(double) * 2 * nj);
    }
}

/* a9  get_cost (Lib/EVAL.m:51-53): 0.5*u'*Qaug*u + paug'*u + caug, Qaug=QQ, paug=ff */
double orc_get_cost(int nn, const double *QQ, const double *ff, double caug, const double *u)
{
    double quad = 0.0, lin = 0.0;
    for (int c = 0; c < nn; ++c) {
        double s = 0.0;
        for (int r = 0; r < nn; ++r) s += u[r] * QQ[r + (size_t)c * nn];
        quad += s * u[c];
        lin += ff[c] * u[c];
    }
    return 0.5 * quad + lin + caug;
}

static double norm2_diff(const double *a, const double *b, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) { double e = a[i] - b[i]; s += e * e; }
    return sqrt(s);
}

/* ------------------------------------------------------------------------- */
/* a8  optimizer loops.  mode 0 = CFS (Lib/CFS_FANUC.m:62-98),                 */
/*                       mode 1 = PSGCFS (Lib/PSGCFS_FANUC.m:65-142).          */
/* Outputs: u[nn], x_[H*ns], cost_all/e_cost_all/e_u_all [max_o_iter],         */
/* *iter_O (reference convention: starts at 1, so iterations run = iter_O-1),  */
/* *total_iter (sum of QP active-set steps, stands in for output.iterations),  */
/* hist_u / hist_x (optional, [max_o_iter][nn] / [max_o_iter][H*ns]).          */
/* noise: [n_noise_rows][nn] draws of normrnd(0,0.1) consumed one row per PSG   */
/* step (PSGCFS_FANUC.m:109); kkt_max (optional) = worst certificate seen.     */
/* ------------------------------------------------------------------------- */
int orc_optimizer(const orc_robot *rb, int mode, int H, int nj, double dt,
                  const double *x_init, const double *xR1,
                  const double *QQ, const double *ff, double caug,
                  const double *Aaug, const double *Baug,
                  const double *lim, const double *max_input,
                  int nobs, const double *obs, const double *margin,
                  double epsilon_O, int max_o_iter, double alpha,
                  const double *noise, int n_noise_rows,
                  double *u, double *x_, double *cost_all, double *e_cost_all, double *e_u_all,
                  int *iter_O_out, int *total_iter_out, double *hist_u, double *hist_x,
                  double *kkt_max)
{
    const int ns = 2 * nj, nn = H * nj, nx = H * ns;
    const int rows = nobs * H * (1 + 2 * nj);
    const int m = rows + (mode == 0 ? 2 * nn : 0);
    double *A = (double *)calloc((size_t)m * nn, sizeof(double));
    double *Acon = (double *)malloc(sizeof(double) * (size_t)rows * nn);
    double *b = (double *)malloc(sizeof(double) * m);
    double *lam = (double *)malloc(sizeof(double) * m);
    double *x_old = (double *)malloc(sizeof(double) * nx);
    double *u_old = (double *)malloc(sizeof(double) * nn);
    double *ev_x = (double *)malloc(sizeof(double) * nx);
    double *uu = (double *)malloc(sizeof(double) * nn);
    double *Ieye = 0, *fneg = 0;
    int status = ORC_OK_MAXITER, iter_O = 1, total_iter = 0, noise_row = 0;
    double cost_old = 100000, cost_new = 0;                 /* EVAL.m:29-30 */
    double kk[4] = {0, 0, 0, 0};
    if (kkt_max) kkt_max[0] = kkt_max[1] = kkt_max[2] = kkt_max[3] = 0.0;
    memcpy(x_, x_init, sizeof(double) * nx);                /* self.x_ = sys_info.x_ */
    memset(u, 0, sizeof(double) * nn);                      /* self.u = zeros(nn,1)  */
    memcpy(ev_x, x_init, sizeof(double) * nx);              /* EVAL.m:46 */
    for (int i = 0; i < nx; ++i) x_old[i] = 1.0;            /* EVAL.m:47 */
    if (mode == 1) {
        Ieye = (double *)calloc((size_t)nn * nn, sizeof(double));
        fneg = (double *)malloc(sizeof(double) * nn);
        for (int i = 0; i < nn; ++i) Ieye[i + (size_t)i * nn] = 1.0;
    }
    cost_new = orc_get_cost(nn, QQ, ff, caug, u);
    for (;;) {
        /* stop_outer (EVAL.m:61-73) */
        double delta = norm2_diff(ev_x, x_old, nx);
        if (delta < epsilon_O) { status = ORC_OK_CONVERGED; break; }
        if (iter_O > max_o_iter) { status = ORC_OK_MAXITER; break; }
        memcpy(u_old, u, sizeof(double) * nn);
        if (mode == 0) cost_old = cost_new;                 /* CFS_FANUC.m:67 */
        orc_get_con(rb, H, nj, ns, x_, u, Baug, Aaug, xR1, lim, nobs, obs, margin, Acon, b, 0, 0, 0);
        for (int c = 0; c < nn; ++c) memcpy(A + (size_t)c * m, Acon + (size_t)c * rows, sizeof(double) * rows);
        int qp_it = 0, rc;
        if (mode == 0) {
            /* bounds -MAX_input <= u <= MAX_input as rows (CFS_FANUC.m:85) */
            for (int c = 0; c < nn; ++c) {
                A[(rows + c) + (size_t)c * m] = 1.0;  b[rows + c] = max_input[c];
                A[(rows + nn + c) + (size_t)c * m] = -1.0; b[rows + nn + c] = max_input[c];
            }
            rc = orc_qp_solve(nn, QQ, ff, m, A, b, uu, lam, &qp_it);
            if (rc == 0 && kkt_max) orc_qp_kkt(nn, QQ, ff, m, A, b, uu, lam, kk);
            total_iter += qp_it;
            if (rc) { status = rc; break; }
            memcpy(u, uu, sizeof(double) * nn);
            memcpy(x_old, x_, sizeof(double) * nx);          /* eval.x_old = self.x_ :88 */
            orc_rollout(H, nj, dt, xR1, u, x_);
            memcpy(ev_x, x_, sizeof(double) * nx);
        } else {
            /* inner_PSG_5 (PSGCFS_FANUC.m:86-103) with MAX_I_ITER=1, epsilon_I=1e-4 */
            int iter_I = 1;
            rc = 0;
            while (!(fabs(cost_new - cost_old) < 1e-4 || iter_I > 1)) {
                cost_old = cost_new;
                double sc = (double)iter_O * (double)iter_O + 1.0;
                for (int r = 0; r < nn; ++r) {
                    double gq = 0.0;
                    for (int c = 0; c < nn; ++c) gq += QQ[r + (size_t)c * nn] * u[c];
                    double nz = (noise && noise_row < n_noise_rows) ? noise[(size_t)noise_row * nn + r] : 0.0;
                    uu[r] = u[r] - alpha * ((gq + ff[r]) + 10.0 * nz / sc);   /* :109 */
                    fneg[r] = -uu[r];
                }
                ++noise_row;
                rc = orc_qp_solve(nn, Ieye, fneg, m, A, b, uu, lam, &qp_it); /* :117-120 */
                if (rc == 0 && kkt_max) orc_qp_kkt(nn, Ieye, fneg, m, A, b, uu, lam, kk);
                total_iter += qp_it;
                if (rc) break;
                memcpy(u, uu, sizeof(double) * nn);
                cost_new = orc_get_cost(nn, QQ, ff, caug, u);
                ++iter_I;
            }
            if (rc) { status = rc; break; }
            orc_rollout(H, nj, dt, xR1, u, x_);
            memcpy(ev_x, x_, sizeof(double) * nx);           /* x_old is never refreshed (N1) */
        }
        if (kkt_max) for (int k = 0; k < 4; ++k) {
            double v = (k == 2) ? -kk[k] : kk[k];
            if (v > kkt_max[k]) kkt_max[k] = v;
        }
        cost_new = orc_get_cost(nn, QQ, ff, caug, u);
        /* store_result (EVAL.m:55-59) */
        int k = iter_O - 1;
        cost_all[k] = cost_new;
        e_cost_all[k] = fabs(cost_old - cost_new);
        e_u_all[k] = norm2_diff(u_old, u, nn);
        if (hist_u) memcpy(hist_u + (size_t)k * nn, u, sizeof(double) * nn);
        if (hist_x) memcpy(hist_x + (size_t)k * nx, x_, sizeof(double) * nx);
        ++iter_O;
    }
    *iter_O_out = iter_O;
    *total_iter_out = total_iter;
    free(A); free(Acon); free(b); free(lam); free(x_old); free(u_old); free(ev_x); free(uu);
    free(Ieye); free(fneg);
    return status;
}

/* Batch driver for the cpu_baseline leg and large parity cases: B independent
 * problems sharing the robot, QQ, dynamics and limits; per problem x_init, xR1,
 * ff, caug, obs, noise.  OpenMP over the batch (nthreads<=0: runtime default). */
void orc_optimizer_batch(const orc_robot *rb, int mode, int B, int H, int nj, double dt,
                         const double *x_init, const double *xR1,
                         const double *QQ, const double *ff, const double *caug,
                         const double *Aaug, const double *Baug,
                         const double *lim, const double *max_input,
                         int nobs, const double *obs, const double *margin,
                         double epsilon_O, int max_o_iter, double alpha,
                         const double *noise, int n_noise_rows,
                         double *u, double *x_, double *cost_all, double *e_cost_all, double *e_u_all,
                         int *iter_O, int *total_iter, int *status, int nthreads)
{
    const int ns = 2 * nj, nn = H * nj, nx = H * ns;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int p = 0; p < B; ++p) {
        status[p] = orc_optimizer(rb, mode, H, nj, dt, x_init + (size_t)p * nx, xR1 + (size_t)p * ns,
                                  QQ, ff + (size_t)p * nn, caug[p], Aaug, Baug, lim, max_input,
                                  nobs, obs + (size_t)p * nobs * 6, margin, epsilon_O, max_o_iter, alpha,
                                  noise ? noise + (size_t)p * n_noise_rows * nn : 0, n_noise_rows,
                                  u + (size_t)p * nn, x_ + (size_t)p * nx,
                                  cost_all + (size_t)p * max_o_iter, e_cost_all + (size_t)p * max_o_iter,
                                  e_u_all + (size_t)p * max_o_iter, iter_O + p, total_iter + p, 0, 0, 0);
    }
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
