// cfs_chomp.hip -- CHOMP_FANUC (Lib/CHOMP_FANUC.m:34-165) on the path's distance kernels:
// one workgroup owns one problem for its whole loop.  Literal restatement, including
//   * dm_f (:105-126) without the M200i joint offset, its derivative taken of dist_link_200i (with the offset);
//   * the joint-space gradient mapped through Baug((i-1)*njoint+1 : i*njoint, :)  (:143, :148): stride njoint, not nstate;
//   * eval.x_ / eval.x_old never refreshed (:55-69): the loop always runs MAX_O_ITER updates;
//   * step 3*alpha (:75), which exceeds 2/lambda_max(QQ) when alpha = 1/sigma_max(QQ) as the drivers set it: the
//     iteration of the reference diverges along the stiff directions, and so does this one.
// The derivatives are derivest(fun, x, 'Vectorized','no') with the suite's defaults (DERIVESTsuite/DERIVESTsuite/
// derivest.m:192-203, :353-468, :475-530): 26 step sizes, 4th-order central rule, 2 Romberg terms, trimmed selection.
#include "cfs_geom_dev.h"
#include "cfs_host.h"

namespace {

constexpr int CT = 256;                 // threads per workgroup
constexpr int DV_NDEL = 26, DV_NE = 23, DV_NEST = 19;
constexpr int CHUNK = 12;               // active (waypoint, obstacle) pairs differentiated at a time

__device__ __forceinline__ double wg_sum(double v, double *red, int tid)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double t = red[0];
    for (int w = 1; w < CT / 64; ++w) t += red[w];
    return t;
}

// distance of link `link` (0-based) to the obstacle axis at joint angles th (offset already applied), near-zero surrogate
// of dist_link_200i.m:19-21 / CHOMP_FANUC.m:121-123 (restated with points(1:3))
template <int NJ>
__device__ double link_dist(const DevRobot *rb, const double *th, int link, const double *o6)
{
    double M[12], Mn[12], e6[6];
    for (int k = 0; k <= link; ++k) {
        double sn, cs;
        sincos(th[k], &sn, &cs);
        fk_step(rb, k, sn, cs, k == 0 ? nullptr : M, Mn);
#pragma unroll
        for (int q = 0; q < 12; ++q) M[q] = Mn[q];
    }
    link_ends(rb, link, M, e6);
    return seg_seg_dist(e6, o6);
}

template <int NJ>
__global__ __launch_bounds__(CT) void cfs_chomp_kernel(ChompParams P)
{
    constexpr int NS = 2 * NJ;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int H = P.H, nobs = P.nobs, HN = H * NJ, nn = HN, NX = H * NS, np = H * nobs;
    const double dt = P.dt;
    DevRobot *rb = reinterpret_cast<DevRobot *>(lds);
    double *s_ob = lds + sizeof(DevRobot) / 8;          // [nobs][6]
    double *s_x = s_ob + nobs * 6;                       // [H][NS]
    double *s_u = s_x + NX, *s_uo = s_u + HN, *s_q = s_uo + HN, *s_G = s_q + HN;   // u, u_old, QQ*u, G (gradient per literal Baug row)
    double *s_E = s_G + HN;                              // [H][nobs][NJ]  coef * dDfx of the active pairs, 0 otherwise
    double *s_coef = s_E + np * NJ;                      // [H][nobs]
    double *s_fd = s_coef + np;                          // [CHUNK][NJ][26] f_del of the chunk being differentiated
    double *red = s_fd + CHUNK * NJ * DV_NDEL;           // 8
    int *s_link = reinterpret_cast<int *>(red + 8);      // [H][nobs] closest link (0-based), -1: inactive
    int *s_act = s_link + np;                            // [H*nobs] list of active pairs
    int *s_nact = s_act + np;

    {
        const double *src = reinterpret_cast<const double *>(P.rb);
        for (int e = tid; e < (int)(sizeof(DevRobot) / 8); e += CT) lds[e] = src[e];
        for (int e = tid; e < nobs * 6; e += CT) s_ob[e] = P.obs[(size_t)b * nobs * 6 + e];
        for (int e = tid; e < NX; e += CT) s_x[e] = P.x_init[(size_t)b * NX + e];          // self.x_ = sys_info.x_ (:46)
        for (int e = tid; e < HN; e += CT) s_u[e] = P.u0[(size_t)b * nn + e];               // self.u = uu (:47)
    }
    __syncthreads();
    double d2 = 0.0;                                     // stop_outer: eval.x_ = sys_info.x_, eval.x_old = ones, never refreshed
    for (int e = tid; e < NX; e += CT) { const double v = s_x[e] - 1.0; d2 += v * v; }
    d2 = wg_sum(d2, red, tid);
    const bool never = sqrt(d2) < P.epsilon_O;           // EVAL.m:64

    // q = QQ*u, cost = 0.5 u'q + ff'u + caug (EVAL.m:52)
    auto matvec_cost = [&]() -> double {
        double cp = 0.0;
        for (int k = tid; k < HN; k += CT) {
            double s = 0.0;
            for (int c = 0; c < HN; ++c) s += P.QQ[k + (size_t)c * nn] * s_u[c];
            s_q[k] = s;
            cp += s_u[k] * (0.5 * s + P.ff[(size_t)b * nn + k]);
        }
        return wg_sum(cp, red, tid) + P.caug[b];
    };
    // dm_f at the current x_ for every (waypoint, obstacle): closest link, regime (:134-152) and the obstacle cost (:87-103)
    auto measure = [&]() -> double {
        double cp = 0.0;
        for (int e = tid; e < np; e += CT) {
            const int i = e / nobs, j = e - i * nobs;
            double th[NJ], Dfx[NJ];
#pragma unroll
            for (int m = 0; m < NJ; ++m) th[m] = s_x[i * NS + m];                           // no joint offset in dm_f (:107-112)
            int lid = 0;
#pragma unroll
            for (int s = 0; s < NJ; ++s) {
                Dfx[s] = link_dist<NJ>(rb, th, s, s_ob + j * 6) - P.D[j];
                if (Dfx[s] < Dfx[lid]) lid = s;                                             // [dis, linkid] = min(Dfx)
                const double eps = P.eps[j];
                double c_x;
                if (Dfx[s] < 0.0) c_x = -Dfx[s] + (1.0 / 2) * eps;
                else if (Dfx[s] <= eps) c_x = (1 / (2 * eps)) * ((Dfx[s] - eps) * (Dfx[s] - eps));
                else c_x = 0.0;
                cp += c_x;
            }
            const double dmin = Dfx[lid], eps = P.eps[j];
            double coef = 0.0;
            int link = -1;
            if (dmin < 0.0) { coef = -1.0; link = lid; }                                    // :139-144
            else if (dmin <= eps) { coef = (1 / eps) * (dmin - eps); link = lid; }          // :145-149
            s_coef[e] = coef;
            s_link[e] = link;
        }
        return wg_sum(cp, red, tid);
    };

    double cost_new = matvec_cost();                     // eval.cost_new = get_cost(u) (:56)
    (void)measure();
    int iter_O = 1;
    while (!never && iter_O <= P.max_o_iter) {
        const double cost_old = cost_new;
        // ---- dcostObs_f (:128-158): derivest of dist_link w.r.t. every joint for the active pairs ---------------------
        if (tid == 0) *s_nact = 0;
        for (int e = tid; e < np * NJ; e += CT) s_E[e] = 0.0;
        __syncthreads();
        for (int e = tid; e < np; e += CT)
            if (s_link[e] >= 0) s_act[atomicAdd(s_nact, 1)] = e;
        __syncthreads();
        const int nact = *s_nact;
        for (int c0 = 0; c0 < nact; c0 += CHUNK) {
            const int cn = min(CHUNK, nact - c0);
            for (int e = tid; e < cn * NJ * DV_NDEL; e += CT) {
                const int jd = e % DV_NDEL, s = (e / DV_NDEL) % NJ, a = e / (DV_NDEL * NJ);
                const int pe = s_act[c0 + a], i = pe / nobs, j = pe - i * nobs;
                double th[NJ];
#pragma unroll
                for (int m = 0; m < NJ; ++m) th[m] = s_x[i * NS + m] - rb->th_off[m];      // dist_link_200i.m:8
                const double x0 = s_x[i * NS + s];
                const double h = x0 > 0.02 ? x0 : 0.02;                                     // derivest.m:229
                const double step = h * P.delta[jd];
                th[s] = (x0 + step) - rb->th_off[s];
                const double fp = link_dist<NJ>(rb, th, s_link[pe], s_ob + j * 6);
                th[s] = (x0 - step) - rb->th_off[s];
                const double fm = link_dist<NJ>(rb, th, s_link[pe], s_ob + j * 6);
                s_fd[(a * NJ + s) * DV_NDEL + jd] = (fp - fm) / 2;                          // :376
            }
            __syncthreads();
            for (int e = tid; e < cn * NJ; e += CT) {
                const int s = e % NJ, a = e / NJ;
                const int pe = s_act[c0 + a], i = pe / nobs;
                const double *fd = s_fd + (a * NJ + s) * DV_NDEL;
                const double x0 = s_x[i * NS + s];
                const double h = x0 > 0.02 ? x0 : 0.02;
                double der_init[DV_NE], der_romb[DV_NEST], errs[DV_NEST];
                for (int q = 0; q < DV_NE; ++q) der_init[q] = (fd[q] * P.fdarule[0] + fd[q + 1] * P.fdarule[1]) / (h * P.delta[q]);   // :419-422
                for (int q = 0; q < DV_NEST; ++q) {                                         // rombextrap, :475-530
                    double c[3], s2 = 0.0;
                    for (int r = 0; r < 3; ++r) {
                        c[r] = 0.0;
                        for (int w = 0; w < 4; ++w) c[r] += P.pinv[r * 4 + w] * der_init[w + q];
                    }
                    for (int w = 0; w < 4; ++w) {
                        const double rr = der_init[w + q] - ((P.rmat[w * 3] * c[0] + P.rmat[w * 3 + 1] * c[1]) + P.rmat[w * 3 + 2] * c[2]);
                        s2 += rr * rr;
                    }
                    der_romb[q] = c[0];
                    errs[q] = sqrt(s2) * P.cov_scale;
                }
                // :439-461: stable ascending sort, the two smallest and two largest dropped, smallest error estimate wins (first)
                int rank_lo[DV_NEST];
                for (int q = 0; q < DV_NEST; ++q) {
                    int r = 0;
                    for (int w = 0; w < DV_NEST; ++w) r += (der_romb[w] < der_romb[q]) || (der_romb[w] == der_romb[q] && w < q);
                    rank_lo[q] = r;                                                          // position of estimate q after the stable sort
                }
                int best = -1, best_rank = 0;
                for (int q = 0; q < DV_NEST; ++q) {
                    const int r = rank_lo[q];
                    if (r < 2 || r >= DV_NEST - 2) continue;
                    if (best < 0 || errs[q] < errs[best] || (errs[q] == errs[best] && r < best_rank)) { best = q; best_rank = r; }
                }
                s_E[pe * NJ + s] = s_coef[pe] * der_romb[best];
            }
            __syncthreads();
        }
        // G(r) = sum over obstacles of coef*dDfx for the literal Baug row r = i*njoint + s  (:143, :148)
        for (int r = tid; r < HN; r += CT) {
            const int i = r / NJ, s = r - i * NJ;
            double g = 0.0;
            for (int j = 0; j < nobs; ++j) g += s_E[(i * nobs + j) * NJ + s];
            s_G[r] = g;
        }
        for (int k = tid; k < HN; k += CT) s_uo[k] = s_u[k];
        __syncthreads();
        // dc(k) = sum_r G(r) * Baug(r, k): row r of Baug is (waypoint w = r / nstate, component q = r % nstate); for the
        // double integrator Baug(w*ns + c, k'*nj + c) = (w - k' + 1/2) dt^2 and Baug(w*ns + nj + c, k'*nj + c) = dt, k' <= w
        for (int k = tid; k < HN; k += CT) {
            const int kp = k / NJ, c = k - kp * NJ;
            double dc = 0.0;
            for (int w = kp; w < H; ++w) {
                const int rp = w * NS + c, rv = w * NS + NJ + c;                            // position / velocity rows of waypoint w
                if (rp < HN) dc += s_G[rp] * (((double)(w - kp) + 0.5) * dt * dt);
                if (rv < HN) dc += s_G[rv] * dt;
            }
            s_u[k] = s_uo[k] - P.alpha * 3 * ((s_q[k] + P.ff[(size_t)b * nn + k]) + 2000 * dc);   // :75
        }
        __syncthreads();
        // new reference (:77-83)
        for (int c = tid; c < NJ; c += CT) {
            double th = P.xR1[(size_t)b * NS + c], om = P.xR1[(size_t)b * NS + NJ + c];
            for (int i = 0; i < H; ++i) {
                const double uu = s_u[i * NJ + c];
                th = (th + dt * om) + (0.5 * dt * dt) * uu;
                om = om + dt * uu;
                s_x[i * NS + c] = th;
                s_x[i * NS + NJ + c] = om;
            }
        }
        __syncthreads();
        const double quad = matvec_cost();
        const double fobs = measure();
        cost_new = quad + fobs;                          // :64
        double du2 = 0.0;
        for (int k = tid; k < HN; k += CT) { const double e = s_uo[k] - s_u[k]; du2 += e * e; }
        du2 = wg_sum(du2, red, tid);
        if (tid == 0) {
            const size_t o = (size_t)b * P.max_o_iter + (iter_O - 1);
            P.cost_all[o] = cost_new;                    // EVAL.m:55-59
            P.e_cost_all[o] = fabs(cost_old - cost_new);
            P.e_u_all[o] = sqrt(du2);
        }
        ++iter_O;
    }
    __syncthreads();
    for (int e = tid; e < HN; e += CT) P.u[(size_t)b * nn + e] = s_u[e];
    for (int e = tid; e < NX; e += CT) P.x_[(size_t)b * NX + e] = s_x[e];
    if (tid == 0) {
        P.iter_O[b] = iter_O;
        if (P.total_iter) P.total_iter[b] = 0;
        if (P.status) P.status[b] = never ? CFS_OK_CONVERGED : CFS_OK_MAXITER;
    }
}

size_t chomp_lds_doubles(int nj, int H, int nobs)
{
    const size_t np = (size_t)H * nobs, HN = (size_t)H * nj;
    return sizeof(DevRobot) / 8 + (size_t)nobs * 6 + (size_t)H * 2 * nj + 4 * HN + np * nj + np + (size_t)CHUNK * nj * DV_NDEL + 8 + (2 * np + 2 + 1) / 2 + 2;
}

}  // namespace

bool chomp_fits(int nj, int H, int nobs) { return chomp_lds_doubles(nj, H, nobs) * 8 <= 64 * 1024; }

hipError_t launch_chomp(int nj, const ChompParams &p, hipStream_t s)
{
    const size_t lds = chomp_lds_doubles(nj, p.H, p.nobs) * 8;
    switch (nj) {
    case 2: hipLaunchKernelGGL(cfs_chomp_kernel<2>, dim3(p.B), dim3(CT), lds, s, p); break;
    case 3: hipLaunchKernelGGL(cfs_chomp_kernel<3>, dim3(p.B), dim3(CT), lds, s, p); break;
    case 4: hipLaunchKernelGGL(cfs_chomp_kernel<4>, dim3(p.B), dim3(CT), lds, s, p); break;
    case 5: hipLaunchKernelGGL(cfs_chomp_kernel<5>, dim3(p.B), dim3(CT), lds, s, p); break;
    case 6: hipLaunchKernelGGL(cfs_chomp_kernel<6>, dim3(p.B), dim3(CT), lds, s, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// derivest's constant tables (see oracle/chomp_oracle.c for the same derivation): derivest.m:238, :282, :478-503, :512-528
void chomp_derivest_tables(ChompParams &p)
{
    const double sr = 2.0000001, srinv = 1.0 / sr;
    for (int k = 0; k < DV_NDEL; ++k) p.delta[k] = 100.0 * pow(sr, -(double)k);
    const double m12 = 1.0 / 6.0, m21 = srinv, m22 = srinv * srinv * srinv / 6.0;
    const double det = m22 - m12 * m21;
    p.fdarule[0] = m22 / det;
    p.fdarule[1] = -m12 / det;
    const double ex[2] = {4.0, 6.0};
    for (int i = 0; i < 4; ++i) {
        p.rmat[i * 3] = 1.0;
        for (int j = 0; j < 2; ++j) p.rmat[i * 3 + 1 + j] = i == 0 ? 1.0 : pow(srinv, i * ex[j]);
    }
    long double N[3][3], Ni[3][3];
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
            long double s = 0;
            for (int i = 0; i < 4; ++i) s += (long double)p.rmat[i * 3 + a] * p.rmat[i * 3 + c];
            N[a][c] = s;
        }
    const long double dtm = N[0][0] * (N[1][1] * N[2][2] - N[1][2] * N[2][1]) - N[0][1] * (N[1][0] * N[2][2] - N[1][2] * N[2][0])
                          + N[0][2] * (N[1][0] * N[2][1] - N[1][1] * N[2][0]);
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) {
            const int a1 = (a + 1) % 3, a2 = (a + 2) % 3, c1 = (c + 1) % 3, c2 = (c + 2) % 3;
            Ni[c][a] = (N[a1][c1] * N[a2][c2] - N[a1][c2] * N[a2][c1]) / dtm;
        }
    for (int a = 0; a < 3; ++a)
        for (int i = 0; i < 4; ++i) {
            long double s = 0;
            for (int c = 0; c < 3; ++c) s += Ni[a][c] * p.rmat[i * 3 + c];
            p.pinv[a * 4 + i] = (double)s;
        }
    p.cov_scale = 12.7062047361747 * sqrt((double)Ni[0][0]);
}
