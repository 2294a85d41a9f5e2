/* chomp_oracle.c -- CPU restatement of CHOMP_FANUC.  TEST INFRASTRUCTURE ONLY.
 *
 * Follows Lib/CHOMP_FANUC.m:34-165 literally, including what looks unintended:
 *   - dm_f (:105-126) measures the links WITHOUT the M200i joint offset (DH(i,1)=theta(i) only), while the
 *     derivative is taken of dist_link_200i (Lib/200i/dist_link_200i.m:1-27), which subtracts pi/2 from joint 2;
 *   - dcostObs_f (:128-158) maps the joint-space gradient through Baug((i-1)*njoint+1:i*njoint,:) -- a stride of
 *     njoint, not nstate, so waypoint i picks position rows of waypoint (i+1)/2 for odd i and velocity rows of
 *     waypoint i/2 for even i;
 *   - the loop never refreshes eval.x_ / eval.x_old (:55-69, Lib/EVAL.m:61-73), so it always runs MAX_O_ITER steps;
 *   - dm_f's near-zero branch (:121-123) subtracts a 3-vector from the stacked 6x1 `points(:,1)`, which MATLAB
 *     rejects; it is restated with points(1:3), as dist_link_200i.m:19-21 does.
 * The derivative is `derivest(fun, x, 'Vectorized','no')` with its defaults, restated from
 * DERIVESTsuite/DERIVESTsuite/derivest.m (vendored in the reference): lines cited below.
 * PARITY UNPINNED: no MATLAB here to run either file; the restatement is checked against analytic derivatives.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXLINK 8
typedef struct {
    int kind, nlink;
    double DH[ORC_MAXLINK * 4], base[3], cap[ORC_MAXLINK * 6], T[9];
} orc_robot;
enum { ORC_ROBOT_M16IB = 0, ORC_ROBOT_M200I = 1, ORC_ROBOT_2L = 2 };

void orc_cap_pos(const double *base, const double *DH, const double *cap, int nlink, double *pos);
double orc_dist_lin_seg(const double *p1s, const double *p1e, const double *p2s, const double *p2e, double *points);
void orc_rollout(int H, int nj, double dt, const double *xR1, const double *u, double *x_);
double orc_get_cost(int nn, const double *QQ, const double *ff, double caug, const double *u);

/* ---- derivest defaults (derivest.m:192-203): DerivativeOrder 1, MethodOrder 4, central, RombergTerms 2 ---------- */
#define DV_NDEL 26
#define DV_NE 23                      /* ndel + 1 - nfda - RombergTerms, :415 */
#define DV_NEST 19                    /* ne - (nexpon + 2), :520 */
typedef struct {
    double delta[DV_NDEL];            /* MaxStep * StepRatio.^(0:-1:-25), :238 */
    double fdarule[2];                /* [1 0]/fdamat(sr,1,2), :282 */
    double rmat[4][3];                /* :478-503 */
    double pinv[3][4];                /* rromb \ qromb.' : least-squares solution operator, :512-521 */
    double cov_scale;                 /* 12.7062047361747 * sqrt(cov1(1)), :526-528 */
} derivest_tab;

void orc_derivest_setup(derivest_tab *t)
{
    const double sr = 2.0000001, srinv = 1.0 / sr;
    for (int k = 0; k < DV_NDEL; ++k) t->delta[k] = 100.0 * pow(sr, -(double)k);
    /* fdamat(sr, 1, 2) (:551-572): mat(i,j) = c(j) * srinv^((i-1)(2j-1)), c = 1./factorial([1 3]) */
    const double m11 = 1.0, m12 = 1.0 / 6.0, m21 = srinv, m22 = srinv * srinv * srinv / 6.0;
    const double det = m11 * m22 - m12 * m21;
    t->fdarule[0] = m22 / det;        /* [1 0] * inv(mat) */
    t->fdarule[1] = -m12 / det;
    const double ex[2] = {4.0, 6.0};  /* rombexpon = 2*(1:2) + 4 - 2, :431 */
    for (int i = 0; i < 4; ++i) {
        t->rmat[i][0] = 1.0;
        for (int j = 0; j < 2; ++j) t->rmat[i][1 + j] = i == 0 ? 1.0 : pow(srinv, i * ex[j]);
    }
    /* (R'R)^-1 R' in long double: the same operator as rromb \ (qromb.' * rhs) */
    long double N[3][3], Ni[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            long double s = 0;
            for (int i = 0; i < 4; ++i) s += (long double)t->rmat[i][a] * t->rmat[i][b];
            N[a][b] = s;
        }
    const long double dt = N[0][0] * (N[1][1] * N[2][2] - N[1][2] * N[2][1]) - N[0][1] * (N[1][0] * N[2][2] - N[1][2] * N[2][0])
                         + N[0][2] * (N[1][0] * N[2][1] - N[1][1] * N[2][0]);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const int a1 = (a + 1) % 3, a2 = (a + 2) % 3, b1 = (b + 1) % 3, b2 = (b + 2) % 3;
            Ni[b][a] = (N[a1][b1] * N[a2][b2] - N[a1][b2] * N[a2][b1]) / dt;   /* cofactor transpose */
        }
    for (int a = 0; a < 3; ++a)
        for (int i = 0; i < 4; ++i) {
            long double s = 0;
            for (int b = 0; b < 3; ++b) s += Ni[a][b] * t->rmat[i][b];
            t->pinv[a][i] = (double)s;
        }
    t->cov_scale = 12.7062047361747 * sqrt((double)Ni[0][0]);   /* cov1(1) = sum(rinv(1,:).^2) = ((R'R)^-1)(1,1) */
}

/* derivest.m:353-468 for one scalar x0; fun(x, ctx) */
double orc_derivest(const derivest_tab *t, double (*fun)(double, void *), void *ctx, double x0, double *errest_out)
{
    const double h = x0 > 0.02 ? x0 : 0.02;                  /* NominalStep = max(x0, 0.02), :229 */
    double f_del[DV_NDEL], der_init[DV_NE], der_romb[DV_NEST], errors[DV_NEST];
    for (int j = 0; j < DV_NDEL; ++j)                         /* :366-371, :376 */
        f_del[j] = (fun(x0 + h * t->delta[j], ctx) - fun(x0 - h * t->delta[j], ctx)) / 2;
    for (int i = 0; i < DV_NE; ++i)                           /* vec2mat(f_del,ne,nfda)*fdarule.' ./ (h*delta), :419-422 */
        der_init[i] = (f_del[i] * t->fdarule[0] + f_del[i + 1] * t->fdarule[1]) / (h * t->delta[i]);
    for (int j = 0; j < DV_NEST; ++j) {                       /* rombextrap, :475-530 */
        double c[3], s2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            c[a] = 0.0;
            for (int i = 0; i < 4; ++i) c[a] += t->pinv[a][i] * der_init[i + j];
        }
        for (int i = 0; i < 4; ++i) {
            const double r = der_init[i + j] - ((t->rmat[i][0] * c[0] + t->rmat[i][1] * c[1]) + t->rmat[i][2] * c[2]);
            s2 += r * r;
        }
        der_romb[j] = c[0];
        errors[j] = sqrt(s2) * t->cov_scale;
    }
    /* :439-461: stable ascending sort with tags, drop the two smallest and the two largest, best error estimate wins */
    int tags[DV_NEST];
    for (int j = 0; j < DV_NEST; ++j) tags[j] = j;
    for (int a = 1; a < DV_NEST; ++a) {
        const int tg = tags[a];
        int b = a - 1;
        while (b >= 0 && der_romb[tags[b]] > der_romb[tg]) { tags[b + 1] = tags[b]; --b; }
        tags[b + 1] = tg;
    }
    int best = -1;
    for (int a = 2; a < DV_NEST - 2; ++a)
        if (best < 0 || errors[tags[a]] < errors[best]) best = tags[a];
    if (errest_out) *errest_out = errors[best];
    return der_romb[best];
}

/* ---- CHOMP_FANUC ------------------------------------------------------------------------------------------------- */
typedef struct {
    const orc_robot *rb;
    int nj, s, linkid;             /* joint being varied, link measured (1-based) */
    double theta[ORC_MAXLINK];
    const double *obs;             /* 6 */
} link_ctx;

/* dist_link_200i / dist_link_Heu (Lib/200i/dist_link_200i.m:1-27, Lib/M16iB/dist_link_Heu.m:1-27) as a function of joint s */
static double dist_link_fun(double x, void *vctx)
{
    const link_ctx *c = (const link_ctx *)vctx;
    double DH[ORC_MAXLINK * 4], pos[ORC_MAXLINK * 6], pts[6];
    memcpy(DH, c->rb->DH, sizeof(double) * 4 * c->nj);                 /* the caller passes DH(1:njoint,:) (CHOMP_FANUC.m:141) */
    for (int i = 0; i < c->nj; ++i) DH[i * 4] = i == c->s ? x : c->theta[i];
    if (c->rb->kind == ORC_ROBOT_M200I) DH[1 * 4] = DH[1 * 4] - M_PI / 2;   /* dist_link_200i.m:8 */
    orc_cap_pos(c->rb->base, DH, c->rb->cap, c->nj, pos);
    const int i = c->linkid - 1;
    double dis = orc_dist_lin_seg(pos + i * 6, pos + i * 6 + 3, c->obs, c->obs + 3, pts);
    if (fabs(dis) < 0.0001) {                                          /* :19-21 */
        const double e0 = pts[0] - pos[i * 6 + 3], e1 = pts[1] - pos[i * 6 + 4], e2 = pts[2] - pos[i * 6 + 5];
        dis = -sqrt(e0 * e0 + e1 * e1 + e2 * e2);
    }
    return dis;
}

/* dm_f (CHOMP_FANUC.m:105-126): per-link distance minus obs.D, NO joint offset */
void orc_chomp_dm(const orc_robot *rb, int nj, const double *theta, const double *obs, double D, double *d)
{
    double DH[ORC_MAXLINK * 4], pos[ORC_MAXLINK * 6], pts[6];
    memcpy(DH, rb->DH, sizeof(double) * 4 * rb->nlink);
    for (int i = 0; i < nj; ++i) DH[i * 4] = theta[i];
    orc_cap_pos(rb->base, DH, rb->cap, rb->nlink, pos);              /* all of robot.DH / robot.cap (:117) */
    for (int i = 0; i < nj; ++i) {
        double dis = orc_dist_lin_seg(pos + i * 6, pos + i * 6 + 3, obs, obs + 3, pts);
        if (fabs(dis) < 0.0001) {
            const double e0 = pts[0] - pos[i * 6 + 3], e1 = pts[1] - pos[i * 6 + 4], e2 = pts[2] - pos[i * 6 + 5];
            dis = -sqrt(e0 * e0 + e1 * e1 + e2 * e2);
        }
        d[i] = dis - D;
    }
}

/* fobs_m (:87-103) */
double orc_chomp_fobs(const orc_robot *rb, int H, int nj, const double *x_, int nobs, const double *obs, const double *D, const double *eps)
{
    double c_all = 0.0, Dfx[ORC_MAXLINK];
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < nobs; ++j) {
            orc_chomp_dm(rb, nj, x_ + 2 * nj * i, obs + 6 * j, D[j], Dfx);
            for (int s = 0; s < nj; ++s) {
                double c_x;
                if (Dfx[s] < 0) c_x = -Dfx[s] + (1.0 / 2) * eps[j];
                else if (Dfx[s] <= eps[j]) c_x = (1 / (2 * eps[j])) * ((Dfx[s] - eps[j]) * (Dfx[s] - eps[j]));
                else c_x = 0;
                c_all = c_all + c_x;
            }
        }
    return c_all;
}

/* dcostObs_f (:128-158).  Baug: (H*ns) x nn column-major. */
void orc_chomp_dcost_obs(const orc_robot *rb, const derivest_tab *t, int H, int nj, const double *x_, int nobs, const double *obs,
                         const double *D, const double *eps, const double *Baug, double *dc_all)
{
    const int nn = H * nj, ns = 2 * nj, ldb = H * ns;
    memset(dc_all, 0, sizeof(double) * nn);
    for (int i = 0; i < H; ++i) {
        const double *theta = x_ + ns * i;
        for (int j = 0; j < nobs; ++j) {
            double Dfx[ORC_MAXLINK], dD[ORC_MAXLINK];
            orc_chomp_dm(rb, nj, theta, obs + 6 * j, D[j], Dfx);
            int linkid = 1;
            for (int s = 1; s < nj; ++s) if (Dfx[s] < Dfx[linkid - 1]) linkid = s + 1;   /* [dis, linkid] = min(Dfx), first minimum */
            const double dmin = Dfx[linkid - 1];
            double coef;
            if (dmin < 0) coef = -1.0;                                         /* :139-144 */
            else if (dmin <= eps[j]) coef = (1 / eps[j]) * (dmin - eps[j]);   /* :145-149 */
            else continue;                                                    /* :150-152 */
            link_ctx c;
            c.rb = rb; c.nj = nj; c.linkid = linkid; c.obs = obs + 6 * j;
            memcpy(c.theta, theta, sizeof(double) * nj);
            for (int s = 0; s < nj; ++s) { c.s = s; dD[s] = orc_derivest(t, dist_link_fun, &c, theta[s], 0); }
            for (int k = 0; k < nn; ++k) {                                    /* dDfx' * Baug((i-1)*njoint+1:i*njoint,:) */
                double g = 0.0;
                for (int s = 0; s < nj; ++s) g += dD[s] * Baug[(i * nj + s) + (size_t)k * ldb];
                dc_all[k] = dc_all[k] + coef * g;
            }
        }
    }
}

/* optimizer (:55-69) + CHOMP_update_arm (:73-85); returns the number of iterations run */
int orc_chomp_optimizer(const orc_robot *rb, int H, int nj, double dt, const double *x_init, const double *xR1, const double *u0,
                        const double *QQ, const double *ff, double caug, const double *Baug,
                        int nobs, const double *obs, const double *D, const double *eps,
                        double epsilon_O, int max_o_iter, double alpha,
                        double *u, double *x_, double *cost_all, double *e_cost_all, double *e_u_all)
{
    const int nn = H * nj, nx = 2 * nj * H;
    derivest_tab t;
    orc_derivest_setup(&t);
    double *u_old = (double *)malloc(sizeof(double) * nn), *dc = (double *)malloc(sizeof(double) * nn);
    memcpy(u, u0, sizeof(double) * nn);                      /* self.u = uu (:47) */
    memcpy(x_, x_init, sizeof(double) * nx);                 /* self.x_ = sys_info.x_ (:46) */
    double d2 = 0.0;                                          /* stop_outer: eval.x_ and eval.x_old never change (EVAL.m:46-47) */
    for (int i = 0; i < nx; ++i) d2 += (x_init[i] - 1.0) * (x_init[i] - 1.0);
    double cost_new = orc_get_cost(nn, QQ, ff, caug, u), cost_old;   /* :56 */
    int iter_O = 1;
    while (!(sqrt(d2) < epsilon_O) && !(iter_O > max_o_iter)) {
        memcpy(u_old, u, sizeof(double) * nn);
        cost_old = cost_new;
        orc_chomp_dcost_obs(rb, &t, H, nj, x_, nobs, obs, D, eps, Baug, dc);
        for (int k = 0; k < nn; ++k) {                        /* u = u_ - alpha*3*(dcostArm_f + 2000*dcostObs_f), :75 */
            double s = 0.0;
            for (int c = 0; c < nn; ++c) s += QQ[k + (size_t)c * nn] * u_old[c];
            u[k] = u_old[k] - alpha * 3 * ((s + ff[k]) + 2000 * dc[k]);
        }
        orc_rollout(H, nj, dt, xR1, u, x_);                   /* :77-83 */
        cost_new = orc_get_cost(nn, QQ, ff, caug, u) + orc_chomp_fobs(rb, H, nj, x_, nobs, obs, D, eps);   /* :64 */
        double du2 = 0.0;
        for (int k = 0; k < nn; ++k) du2 += (u_old[k] - u[k]) * (u_old[k] - u[k]);
        cost_all[iter_O - 1] = cost_new;                      /* EVAL.m:55-59 */
        e_cost_all[iter_O - 1] = fabs(cost_old - cost_new);
        e_u_all[iter_O - 1] = sqrt(du2);
        ++iter_O;
    }
    free(u_old); free(dc);
    return iter_O;
}
