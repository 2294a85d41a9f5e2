// cfs_qp.hip -- K2: the strictly convex QP of one CFS / PSGCFS outer iteration for a whole batch,
// followed by the state rollout (rows a6, a7 and the second half of a5 in DESIGN.md).
//
// Reference behaviour restated:
//   Lib/CFS_FANUC.m:83-98      quadprog(QQ,ff,Ainq,binq,[],[],-MAX_input,MAX_input) + rollout
//   Lib/PSGCFS_FANUC.m:86-128  u_ = u - alpha*(QQ*u+ff+10*xi/(iter_O^2+1)); quadprog(I,-u_,Ainq,binq)
//   Lib/CFS_FANUC.m:119-129    the rows of Ainq/binq (never materialised here)
// quadprog is closed source; the minimiser of a strictly convex QP is unique, so the solver is a
// design choice: a Goldfarb-Idnani dual active-set method written for one 64-lane wavefront.
//
// MI355X mapping.  One wavefront per problem, lane i owns waypoint i: its NJ inputs u_i and the
// rolled-out joint velocity / position displacements live in registers.  The constraint rows are
// structured -- a collision row touches only the position of one waypoint, a velocity row one
// velocity component, a bound one input -- so Ainq is never formed: slacks are NJ-term dot
// products against registers, and H^{-1} times a constraint normal is a gather of <= NJ columns
// of three problem-family matrices (H^{-1}Bpos', H^{-1}Bvel', H^{-1}) that stay in L2.
// Rollouts are two wave prefix sums (the dynamics are the robot's double integrator,
// robotproperty2.m:136-139).  The active set is kept in range-space form: Y = H^{-1}N (q x nn) and
// a factor T with (N'H^{-1}N)^{-1} = T T'.  Adding a constraint borders T in O(q); dropping one is
// a Householder reflection -- there is no sequential triangular solve anywhere, every step is a
// small mat-vec or rank-1 update across the lanes.  Y and T live in LDS for q <= QP_QL; problems
// whose active set outgrows that (mostly infeasible linearisations, whose certificate needs up
// to nn constraints) are re-run by the same code instantiated with Y/T in global memory.
#include "cfs_device.h"

namespace {

constexpr int QP_QL = 16;        // active-set capacity of the LDS instantiation
constexpr int QP_MAXIT = 6000;   // cap on active-set steps per QP
constexpr double DEP_TOL = 1e-8;

enum { CT_COL = 0, CT_VELP = 1, CT_VELM = 2, CT_BNDP = 3, CT_BNDM = 4 };
__device__ __forceinline__ int mk_code(int type, int i, int jc) { return (type << 16) | (i << 8) | jc; }

__device__ __forceinline__ double wave_scan(double v, int lane)
{
#pragma unroll
    for (int off = 1; off < CFS_WAVE; off <<= 1) {
        const double t = __shfl_up(v, off, CFS_WAVE);
        if (lane >= off) v += t;
    }
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, CFS_WAVE);
    return v;
}

// all-lanes argmin of (v, id); ties broken towards the smaller id so that every lane agrees
__device__ __forceinline__ void wave_argmin(double &v, int &id)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = __shfl_xor(v, m, CFS_WAVE);
        const int oi = __shfl_xor(id, m, CFS_WAVE);
        if (ov < v || (ov == v && oi < id)) { v = ov; id = oi; }
    }
}

// velocity / position displacement of a per-lane input block: yv = Bvel*x, yp = Bpos*x
template <int NJ>
__device__ __forceinline__ void roll(const double *x, double *yv, double *yp, double dt, int lane)
{
#pragma unroll
    for (int c = 0; c < NJ; ++c) {
        yv[c] = dt * wave_scan(x[c], lane);
        yp[c] = dt * wave_scan(yv[c], lane) - 0.5 * dt * yv[c];
    }
}

// inward normal of constraint `code` applied to a vector given as (w, Bvel w, Bpos w) in LDS
template <int NJ>
__device__ __forceinline__ double ndot(int code, const double *buf, const double *g, int H)
{
    const int type = code >> 16, i = (code >> 8) & 0xff, jc = code & 0xff, HN = H * NJ;
    switch (type) {
    case CT_COL: {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < NJ; ++c) s += g[(jc * H + i) * NJ + c] * buf[2 * HN + i * NJ + c];
        return s;
    }
    case CT_VELP: return -buf[HN + i * NJ + jc];
    case CT_VELM: return buf[HN + i * NJ + jc];
    case CT_BNDP: return -buf[i * NJ + jc];
    default: return buf[i * NJ + jc];
    }
}

template <int NJ, bool BIG>
__global__ __launch_bounds__(CFS_WAVE) void cfs_qp_kernel(QpParams P)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int H = P.H, nobs = P.nobs, HN = H * NJ, nn = HN, NS = 2 * NJ;
    if (P.done && P.done[b]) return;
    if (BIG) { if (P.qp_status[b] != QP_OVERFLOW) return; }
    const bool wp = lane < H;
    const double dt = P.dt;
    const int QC = BIG ? nn : QP_QL;

    // ---- LDS carve-up -------------------------------------------------------------------------
    double *s_g = lds;                           // [nobs][H][NJ]
    double *s_rhs = s_g + nobs * HN;             // [nobs][H]
    double *s_w = s_rhs + nobs * H;              // [3][H][NJ]  w, Bvel w, Bpos w
    double *s_z = s_w;                           // z aliases w: w's LDS copy is dead once d = N'w is formed
    double *s_d = s_w + 3 * HN;                  // [QC]
    double *s_v = s_d + QC;                      // [QC]
    double *s_r = s_v + QC;                      // [QC]
    double *s_lam = s_r + QC;                    // [QC]
    int *s_act = reinterpret_cast<int *>(s_lam + QC);          // [QC]
    double *s_Y = reinterpret_cast<double *>(s_act + ((QC + 1) & ~1));
    double *Yst, *Tst;
    int ldy, ldt;
    if (BIG) {
        Yst = P.Yg + (size_t)b * nn * nn; ldy = nn;
        Tst = P.Tg + (size_t)b * nn * nn; ldt = nn;
    } else {
        Yst = s_Y; ldy = HN;                     // [QP_QL][H*NJ]
        Tst = s_Y + QP_QL * HN; ldt = QP_QL + 1; // [QP_QL][QP_QL+1]
    }

    // ---- prologue: per-lane state ---------------------------------------------------------------
    double ul[NJ], x[NJ], yv[NJ], yp[NJ], mx[NJ], v0[NJ], lm[NJ];
    const double *ub = P.u + (size_t)b * nn;
#pragma unroll
    for (int c = 0; c < NJ; ++c) {
        ul[c] = wp ? ub[lane * NJ + c] : 0.0;
        mx[c] = (wp && P.has_bounds) ? P.maxin[lane * NJ + c] : 0.0;
        v0[c] = P.xR1[(size_t)b * NS + NJ + c];
        lm[c] = P.lim[c];
    }
    int status = QP_OK, iters = 0;
    bool skip = false;
    if (P.mode == CFS_MODE_PSGCFS) {
        // stop_inner (PSGCFS_FANUC.m:136-142) with iter_I = 1: one PSG step unless the cost stalled
        // (the global-memory instantiation re-runs a step the LDS one already accounted for)
        const double cn = P.cost_new[b], co = P.cost_old_in[b];
        skip = BIG ? false : fabs(cn - co) < 1e-4;
        if (!skip) {
            const int k = P.iter_O[b];
            const double sc = (double)k * (double)k + 1.0;
            const int nr = BIG ? P.noise_row[b] - 1 : P.noise_row[b];
            const bool have = P.noise != nullptr && nr < P.noise_rows;
#pragma unroll
            for (int c = 0; c < NJ; ++c) {
                if (wp) {
                    const size_t e = (size_t)b * nn + lane * NJ + c;
                    const double nz = have ? P.noise[((size_t)b * P.noise_rows + nr) * nn + lane * NJ + c] : 0.0;
                    x[c] = ul[c] - P.alpha * ((P.qu[e] + P.ff[e]) + 10.0 * nz / sc);   // PSGCFS_FANUC.m:109
                } else x[c] = 0.0;
            }
            if (!BIG && lane == 0) { P.noise_row[b] = nr + 1; P.cost_old_out[b] = cn; }
        }
    } else {
#pragma unroll
        for (int c = 0; c < NJ; ++c) x[c] = wp ? P.x0[(size_t)b * nn + lane * NJ + c] : 0.0;
    }

    int q = 0;
    unsigned long long amask = 0ull;             // active constraints of this lane's waypoint
    if (!skip) {
        // linearisation data: g, and rhs = (d - margin) - g'*Bpos_i*u_lin   (CFS_FANUC.m:119-120)
        double plv[NJ], pl[NJ];
        roll<NJ>(ul, plv, pl, dt, lane);
        const double *gb = P.grad + (size_t)b * nobs * HN;
        const double *db = P.dist + (size_t)b * nobs * H;
        for (int e = lane; e < nobs * HN; e += CFS_WAVE) s_g[e] = gb[e];
        __syncthreads();
        if (wp)
            for (int j = 0; j < nobs; ++j) {
                double gp = 0.0;
#pragma unroll
                for (int c = 0; c < NJ; ++c) gp += s_g[(j * H + lane) * NJ + c] * pl[c];
                s_rhs[j * H + lane] = (db[j * H + lane] - P.margin[j]) - gp;
            }
        roll<NJ>(x, yv, yp, dt, lane);
        __syncthreads();
    }

    // ---- dual active-set iterations ---------------------------------------------------------------
    while (!skip) {
        // step 1: most violated constraint among this lane's rows, then across the wave
        double sbest = 0.0;
        int cbest = 0x7fffffff;
        if (wp) {
            for (int j = 0; j < nobs; ++j) {
                if ((amask >> j) & 1ull) continue;
                const double rh = s_rhs[j * H + lane];
                double s = rh;
#pragma unroll
                for (int c = 0; c < NJ; ++c) s += s_g[(j * H + lane) * NJ + c] * yp[c];
                if (s < -1e-11 * (1.0 + fabs(rh)) && s < sbest) { sbest = s; cbest = mk_code(CT_COL, lane, j); }
            }
#pragma unroll
            for (int c = 0; c < NJ; ++c) {
                const double bp = lm[c] - v0[c], bm = lm[c] + v0[c];   // CFS_FANUC.m:127,129
                const double sp_ = bp - yv[c], sm_ = bm + yv[c];
                if (!((amask >> (32 + c)) & 1ull) && sp_ < -1e-11 * (1.0 + fabs(bp)) && sp_ < sbest) { sbest = sp_; cbest = mk_code(CT_VELP, lane, c); }
                if (!((amask >> (40 + c)) & 1ull) && sm_ < -1e-11 * (1.0 + fabs(bm)) && sm_ < sbest) { sbest = sm_; cbest = mk_code(CT_VELM, lane, c); }
                if (P.has_bounds) {
                    const double tp = mx[c] - x[c], tm = mx[c] + x[c];
                    if (!((amask >> (48 + c)) & 1ull) && tp < -1e-11 * (1.0 + fabs(mx[c])) && tp < sbest) { sbest = tp; cbest = mk_code(CT_BNDP, lane, c); }
                    if (!((amask >> (56 + c)) & 1ull) && tm < -1e-11 * (1.0 + fabs(mx[c])) && tm < sbest) { sbest = tm; cbest = mk_code(CT_BNDM, lane, c); }
                }
            }
        }
        wave_argmin(sbest, cbest);
        if (cbest == 0x7fffffff) break;           // feasible: optimum reached
        const int pc = cbest, ptype = pc >> 16, pi = (pc >> 8) & 0xff, pj = pc & 0xff;
        double sp = sbest, lam_p = 0.0;

        // step 2: bring constraint p in, dropping blocking constraints on the way
        for (;;) {
            if (++iters > QP_MAXIT) { status = QP_NUMERIC; break; }
            // w = H^{-1} n_p : gather of <= NJ columns of the family matrices
            double w[NJ], wv[NJ], wpz[NJ];
#pragma unroll
            for (int c = 0; c < NJ; ++c) w[c] = 0.0;
            if (wp) {
                if (ptype == CT_COL) {
#pragma unroll
                    for (int cs = 0; cs < NJ; ++cs) {
                        const double gc = s_g[(pj * H + pi) * NJ + cs];
                        const double *col = P.M1 + ((size_t)(pi * NJ + cs) * NJ) * H + lane;
#pragma unroll
                        for (int c = 0; c < NJ; ++c) w[c] += gc * col[c * H];
                    }
                } else {
                    const double *Mx = (ptype == CT_VELP || ptype == CT_VELM) ? P.M2 : P.M3;
                    const double sg = (ptype == CT_VELP || ptype == CT_BNDP) ? -1.0 : 1.0;
                    const double *col = Mx + ((size_t)(pi * NJ + pj) * NJ) * H + lane;
#pragma unroll
                    for (int c = 0; c < NJ; ++c) w[c] = sg * col[c * H];
                }
            }
            roll<NJ>(w, wv, wpz, dt, lane);
            if (wp) {
#pragma unroll
                for (int c = 0; c < NJ; ++c) {
                    s_w[lane * NJ + c] = w[c];
                    s_w[HN + lane * NJ + c] = wv[c];
                    s_w[2 * HN + lane * NJ + c] = wpz[c];
                }
            }
            __syncthreads();
            const double spp = ndot<NJ>(pc, s_w, s_g, H);          // n_p' H^{-1} n_p
            for (int a = lane; a < q; a += CFS_WAVE) s_d[a] = ndot<NJ>(s_act[a], s_w, s_g, H);
            __syncthreads();
            // r = T (T' d)
            for (int bq = lane; bq < q; bq += CFS_WAVE) {
                double s = 0.0;
                for (int a = 0; a < q; ++a) s += Tst[a * ldt + bq] * s_d[a];
                s_v[bq] = s;
            }
            __syncthreads();
            for (int a = lane; a < q; a += CFS_WAVE) {
                double s = 0.0;
                for (int bq = 0; bq < q; ++bq) s += Tst[a * ldt + bq] * s_v[bq];
                s_r[a] = s;
            }
            __syncthreads();
            // z = w - Y' r  (primal step direction), and its rollout
            double z[NJ], zv[NJ], zp[NJ];
#pragma unroll
            for (int c = 0; c < NJ; ++c) z[c] = w[c];
            if (wp)
                for (int a = 0; a < q; ++a) {
                    const double ra = s_r[a];
#pragma unroll
                    for (int c = 0; c < NJ; ++c) z[c] -= ra * Yst[(size_t)a * ldy + lane * NJ + c];
                }
            roll<NJ>(z, zv, zp, dt, lane);
            if (wp) {
#pragma unroll
                for (int c = 0; c < NJ; ++c) {
                    s_z[lane * NJ + c] = z[c];
                    s_z[HN + lane * NJ + c] = zv[c];
                    s_z[2 * HN + lane * NJ + c] = zp[c];
                }
            }
            __syncthreads();
            const double delta = ndot<NJ>(pc, s_z, s_g, H);        // n_p' z = curvature along z
            const bool dependent = !(delta > DEP_TOL * spp);
            // step lengths
            double t1 = INFINITY;
            int l = 0x7fffffff;
            for (int a = lane; a < q; a += CFS_WAVE) {
                const double ra = s_r[a];
                if (ra > 0.0) {
                    const double tt = s_lam[a] / ra;
                    if (tt < t1) { t1 = tt; l = a; }
                }
            }
            wave_argmin(t1, l);
            const double t2 = dependent ? INFINITY : -sp / delta;
            const double t = fmin(t1, t2);
            if (!(t < INFINITY)) { status = QP_INFEASIBLE; break; }
            const bool full = !dependent && t2 <= t1;
            if (!dependent) {
#pragma unroll
                for (int c = 0; c < NJ; ++c) { x[c] += t * z[c]; yv[c] += t * zv[c]; yp[c] += t * zp[c]; }
            }
            for (int a = lane; a < q; a += CFS_WAVE) s_lam[a] -= t * s_r[a];
            lam_p += t;
            if (full) {
                if (q == QC) { status = BIG ? QP_NUMERIC : QP_OVERFLOW; break; }
                const double rho = sqrt(delta);
                for (int a = lane; a < q; a += CFS_WAVE) {
                    Tst[a * ldt + q] = -s_r[a] / rho;
                    Tst[q * ldt + a] = 0.0;
                }
                if (wp) {
#pragma unroll
                    for (int c = 0; c < NJ; ++c) Yst[(size_t)q * ldy + lane * NJ + c] = w[c];
                }
                if (lane == 0) { Tst[q * ldt + q] = 1.0 / rho; s_act[q] = pc; s_lam[q] = lam_p; }
                if (lane == pi) amask |= 1ull << (ptype == CT_COL ? pj : 24 + 8 * ptype + pj);
                ++q;
                __syncthreads();
                break;                                             // back to step 1
            }
            // partial step: drop blocking constraint l (Householder on T), keep working on p
            {
                const int last = q - 1;
                const int gone = s_act[l];
                for (int bq = lane; bq < q; bq += CFS_WAVE) s_v[bq] = Tst[l * ldt + bq];
                __syncthreads();
                double n2 = 0.0;
                for (int bq = 0; bq < q; ++bq) n2 += s_v[bq] * s_v[bq];
                const double nrm = sqrt(n2);
                const double clast = s_v[last] / nrm;
                const double sgn = clast >= 0.0 ? 1.0 : -1.0;
                const double beta = 1.0 / (1.0 + fabs(clast));
                __syncthreads();
                for (int bq = lane; bq < q; bq += CFS_WAVE) s_v[bq] = s_v[bq] / nrm + (bq == last ? sgn : 0.0);
                __syncthreads();
                for (int a = lane; a < q; a += CFS_WAVE) {
                    if (a == l) continue;
                    double tau = 0.0;
                    for (int bq = 0; bq < q; ++bq) tau += Tst[a * ldt + bq] * s_v[bq];
                    tau *= beta;
                    for (int bq = 0; bq < last; ++bq) Tst[a * ldt + bq] -= tau * s_v[bq];
                }
                __syncthreads();
                if (l != last) {
                    for (int bq = lane; bq < last; bq += CFS_WAVE) Tst[l * ldt + bq] = Tst[last * ldt + bq];
                    if (wp) {
#pragma unroll
                        for (int c = 0; c < NJ; ++c) Yst[(size_t)l * ldy + lane * NJ + c] = Yst[(size_t)last * ldy + lane * NJ + c];
                    }
                    if (lane == 0) { s_act[l] = s_act[last]; s_lam[l] = s_lam[last]; }
                }
                const int gt = gone >> 16, gi = (gone >> 8) & 0xff, gj = gone & 0xff;
                if (lane == gi) amask &= ~(1ull << (gt == CT_COL ? gj : 24 + 8 * gt + gj));
                q = last;
                __syncthreads();
            }
            // slack of p at the new x (held by lane pi)
            {
                double s = 0.0;
                if (lane == pi) {
                    if (ptype == CT_COL) {
                        s = s_rhs[pj * H + pi];
#pragma unroll
                        for (int c = 0; c < NJ; ++c) s += s_g[(pj * H + pi) * NJ + c] * yp[c];
                    } else {
#pragma unroll
                        for (int c = 0; c < NJ; ++c)
                            if (c == pj) {
                                if (ptype == CT_VELP) s = (lm[c] - v0[c]) - yv[c];
                                else if (ptype == CT_VELM) s = (lm[c] + v0[c]) + yv[c];
                                else if (ptype == CT_BNDP) s = mx[c] - x[c];
                                else s = mx[c] + x[c];
                            }
                    }
                }
                sp = __shfl(s, pi, CFS_WAVE);
            }
        }
        if (status != QP_OK) break;
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    if (lane == 0) {
        P.qp_status[b] = skip ? QP_SKIPPED : status;
        if (P.qp_iter) P.qp_iter[b] = BIG ? P.qp_iter[b] + iters : iters;
    }
    if (status != QP_OK) return;                  // u, x_ untouched (overflow: the big instantiation re-runs)
    if (skip) {
#pragma unroll
        for (int c = 0; c < NJ; ++c) x[c] = ul[c];
        roll<NJ>(x, yv, yp, dt, lane);
    }
    if (P.lambda) {
        double *lb = P.lambda + (size_t)b * (nobs * H + 4 * nn);
        for (int e = lane; e < nobs * H + 4 * nn; e += CFS_WAVE) lb[e] = 0.0;
        __syncthreads();
        for (int a = lane; a < q; a += CFS_WAVE) {
            const int code = s_act[a], ty = code >> 16, i = (code >> 8) & 0xff, jc = code & 0xff;
            const int idx = ty == CT_COL ? jc * H + i : nobs * H + (ty - 1) * nn + i * NJ + jc;
            lb[idx] = s_lam[a];
        }
    }
    // new u, rollout xR(:,i) = A xR(:,i-1) + B u_{i-1} (CFS_FANUC.m:90-94) in closed form, and the
    // two norms the outer loop needs (EVAL.m:58, :64)
    double du2 = 0.0, dx2 = 0.0;
    if (wp) {
        double *uo = P.u + (size_t)b * nn + lane * NJ;
#pragma unroll
        for (int c = 0; c < NJ; ++c) {
            const double e = ul[c] - x[c];
            du2 += e * e;
            uo[c] = x[c];
        }
        if (P.x_) {
            double *xo = P.x_ + (size_t)b * H * NS + lane * NS;
            const double *x1 = P.xR1 + (size_t)b * NS;
            const double tk = (double)(lane + 1) * dt;
#pragma unroll
            for (int c = 0; c < NJ; ++c) {
                const double th = (x1[c] + tk * v0[c]) + yp[c];
                const double om = v0[c] + yv[c];
                const double oth = (P.mode == CFS_MODE_PSGCFS) ? 1.0 : xo[c];       // EVAL.m:47, N1
                const double oom = (P.mode == CFS_MODE_PSGCFS) ? 1.0 : xo[NJ + c];
                dx2 += (th - oth) * (th - oth) + (om - oom) * (om - oom);
                xo[c] = th;
                xo[NJ + c] = om;
            }
        }
    }
    du2 = wave_sum(du2);
    dx2 = wave_sum(dx2);
    if (lane == 0) {
        if (P.e_u) P.e_u[b] = sqrt(du2);
        if (P.delta) P.delta[b] = sqrt(dx2);
    }
}

size_t qp_lds_bytes(int nj, int H, int nobs, bool big)
{
    const int HN = H * nj, QC = big ? HN : QP_QL;
    size_t dbl = (size_t)nobs * HN + (size_t)nobs * H + 3 * (size_t)HN + 4 * (size_t)QC + ((QC + 1) & ~1) / 2 + 1;
    if (!big) dbl += (size_t)QP_QL * HN + (size_t)QP_QL * (QP_QL + 1);
    return dbl * 8;
}

template <int NJ>
void launch_qp_nj(const QpParams &p, bool big, hipStream_t s)
{
    const dim3 grid(p.B), block(CFS_WAVE);
    const size_t lds = qp_lds_bytes(NJ, p.H, p.nobs, big);
    if (big) hipLaunchKernelGGL((cfs_qp_kernel<NJ, true>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((cfs_qp_kernel<NJ, false>), grid, block, lds, s, p);
}

}  // namespace

void launch_qp(int nj, const QpParams &p, bool big, hipStream_t s)
{
    switch (nj) {
    case 2: launch_qp_nj<2>(p, big, s); break;
    case 3: launch_qp_nj<3>(p, big, s); break;
    case 4: launch_qp_nj<4>(p, big, s); break;
    case 5: launch_qp_nj<5>(p, big, s); break;
    case 6: launch_qp_nj<6>(p, big, s); break;
    default: break;
    }
}
