// cfs_fused.hip -- the whole optimizer() loop of one problem in ONE persistent workgroup.
//
// Reference behaviour restated (row a8 of DESIGN.md ties a1-a7, a9 together):
//   Lib/CFS_FANUC.m:62-79      while ~stop_outer: get_con; Solve_QP; get_cost; store_result; iter_O++
//   Lib/PSGCFS_FANUC.m:65-103  same with one projected stochastic-gradient step per outer iteration
//   Lib/EVAL.m:51-73           cost, history, stop test
//
// MI355X mapping.  Problems of a batch are independent and very unequal (most converge in a few
// outer iterations with a handful of active constraints; infeasible linearisations need ~200
// active-set steps with up to nn active rows to be certified).  One 256-thread workgroup
// therefore owns one problem for its whole life: no host round trip, no inter-problem barrier,
// and the hardware's workgroup dispatcher load-balances the stragglers over the 256 CUs.
//   * linearisation (FK variants, segment pairs, literal num_jac) as in cfs_geom.hip, tiled over
//     waypoints, scratch aliased with the QP's Y storage;
//   * QP: Goldfarb-Idnani dual active set in range-space form.  The inverse Gram matrix
//     P = (N'H^{-1}N)^{-1} is kept EXPLICITLY, one row per thread in registers (it is symmetric,
//     so column access is never needed); adding a constraint is a bordered rank-1 update,
//     dropping one a rank-1 downdate + swap-with-last: every step is O(q) per thread with O(1)
//     depth.  Each direction is polished by iterative refinement against the TRUE Gram matrix
//     (structured dots n_a'z), which keeps the explicit inverse honest over hundreds of updates.
//     Y = H^{-1}N rows live in LDS (first QY rows) and spill to global memory beyond;
//   * rollouts are short prefix sums from LDS, wave reductions use DPP row shifts (no ds_bpermute).
#include "cfs_geom_dev.h"
#include <algorithm>
#include <atomic>

namespace {

#define STAMP(k) do { if (P.stamps) { const unsigned long long t_ = clock64(); if (tid == 0) s_acc[k] += t_ - t0_; t0_ = t_; } } while (0)

#ifndef CFS_CERT_AT
#define CFS_CERT_AT 6
#endif
constexpr int CERT_AT = CFS_CERT_AT;    // main-loop steps of a QP before the step-free infeasibility certificate is asked
#ifndef CFS_WG_PER_CU
#define CFS_WG_PER_CU 1                   // workgroups resident per CU (2: half the LDS and registers each)
#endif
#ifndef CFS_ANGLE_ADD
#define CFS_ANGLE_ADD 0                    // 1: sin/cos of theta +- eps/2 by angle addition (measured: no faster, and the 1e-16 differences it seeds cost parity on the chaotic minority)
#endif
#ifndef CFS_REF_A
#define CFS_REF_A 1e-4                   // refinement of a step direction continues while |r'rho| > A * max(|delta|, tol * n'H^-1 n) ...
#define CFS_REF_B 1e-9                   // ... or max|rho| > B * max|d|  (rho_a = n_a'z, zero in exact arithmetic)
#endif
#ifndef CFS_MV_BATCH
#define CFS_MV_BATCH 16                  // QQ*u: loads in flight per thread
#endif
#ifndef CFS_TU
#define CFS_TU 8                         // tail columns of P / global rows of Y loaded per batch (independent loads in flight)
#endif
#ifndef CFS_LIN_UNROLL
#define CFS_LIN_UNROLL 0                 // 1: two segment pairs per thread in flight in the distance loops of the linearisation
#endif
#ifndef CFS_PR
#define CFS_PR 64                        // columns of each inverse-Gram row kept in registers
#endif
// This file is compiled once per tier (Makefile): CFS_VARIANT names the exported launch_fused_<tier> / fused_fits_<tier>.
#ifndef CFS_VARIANT
#define CFS_VARIANT w1
#endif
#define CFS_CAT2(a, b) a##_##b
#define CFS_CAT(a, b) CFS_CAT2(a, b)
static_assert(CFS_PR % 8 == 0 && CFS_PR >= 8, "register-resident P columns come in chunks of 8");
constexpr int FT = 256;                  // threads per workgroup
constexpr double DEP_TOL_F = 1e-8;       // dependent if delta <= tol * n'H^{-1}n: above the eps*cond(H) noise floor of Y = H^{-1}N
enum { CT_COL = 0, CT_VELP = 1, CT_VELM = 2, CT_BNDP = 3, CT_BNDM = 4 };
__device__ __forceinline__ int mk_code(int type, int i, int jc) { return (type << 16) | (i << 8) | jc; }

// ---- DPP helpers (gfx9 row shifts / row broadcasts) ---------------------------------------------
template <int CTRL, int RM>
__device__ __forceinline__ double dpp_f64(double old, double src)
{
    const int rl = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, RM, 0xf, false);
    const int rh = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, RM, 0xf, false);
    return __hiloint2double(rh, rl);
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_min(double v)      // all lanes receive the minimum
{
    v = fmin(v, dpp_f64<0x111, 0xf>(v, v));   // row_shr:1
    v = fmin(v, dpp_f64<0x112, 0xf>(v, v));   // row_shr:2
    v = fmin(v, dpp_f64<0x114, 0xf>(v, v));   // row_shr:4
    v = fmin(v, dpp_f64<0x118, 0xf>(v, v));   // row_shr:8
    v = fmin(v, dpp_f64<0x142, 0xa>(v, v));   // row_bcast:15 -> rows 1,3
    v = fmin(v, dpp_f64<0x143, 0xc>(v, v));   // row_bcast:31 -> rows 2,3
    return readlane_f64(v, 63);
}
__device__ __forceinline__ double wave_add(double v)      // all lanes receive the sum
{
    v += dpp_f64<0x111, 0xf>(0.0, v);
    v += dpp_f64<0x112, 0xf>(0.0, v);
    v += dpp_f64<0x114, 0xf>(0.0, v);
    v += dpp_f64<0x118, 0xf>(0.0, v);
    v += dpp_f64<0x142, 0xa>(0.0, v);
    v += dpp_f64<0x143, 0xc>(0.0, v);
    return readlane_f64(v, 63);
}

// Block-wide reductions cost ONE barrier each: consecutive reductions alternate between two exchange buffers (the
// caller passes red_a / red_b in turn), so a wavefront may still be reading the previous result while the next one is
// being written; the buffer written two reductions ago is free because the reduction in between had a barrier.
// Ordering of LDS traffic among the active rows.  While every active row lives in wavefront 0 (qhi <= 64: the normal
// case, PSGCFS projections end with 2-3 active rows), writes and reads of the row vectors (d, r, rho, lambda, P
// bookkeeping) are ordered by the wavefront's own in-order LDS queue: a wavefront-scope fence instead of s_barrier.
__device__ __forceinline__ void sync_rows(bool one_wave)
{
    if (one_wave) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else __syncthreads();
}

// block-wide argmin of (v, id): every thread returns the same pair.  red: >= 2*(FT/64) doubles.
__device__ __forceinline__ void block_argmin(double &v, int &id, double *red, int tid)
{
    const double m = wave_min(v);
    const unsigned long long hit = __ballot(v == m);
    const int src = hit ? (int)__builtin_ctzll(hit) : 0;
    const int mid = __builtin_amdgcn_readlane(id, src);
    const int wv = tid >> 6;
    if ((tid & 63) == 0) { red[wv] = m; reinterpret_cast<int *>(red + FT / 64)[wv] = mid; }
    __syncthreads();
    double bm = red[0];
    int bi = reinterpret_cast<int *>(red + FT / 64)[0];
#pragma unroll
    for (int w = 1; w < FT / 64; ++w) {
        const double om = red[w];
        const int oi = reinterpret_cast<int *>(red + FT / 64)[w];
        if (om < bm) { bm = om; bi = oi; }
    }
    v = bm; id = bi;
}
__device__ __forceinline__ double block_sum(double v, double *red, int tid)
{
    const double s = wave_add(v);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    double t = red[0];
#pragma unroll
    for (int w = 1; w < FT / 64; ++w) t += red[w];
    return t;
}

// four sums through one exchange
__device__ __forceinline__ void block_sum4(double &a, double &b, double &c, double &d, double *red, int tid)
{
    const double sa = wave_add(a), sb = wave_add(b), sc = wave_add(c), sd = wave_add(d);
    if ((tid & 63) == 0) { const int w = tid >> 6; red[w] = sa; red[4 + w] = sb; red[8 + w] = sc; red[12 + w] = sd; }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]);
    b = (red[4] + red[5]) + (red[6] + red[7]);
    c = (red[8] + red[9]) + (red[10] + red[11]);
    d = (red[12] + red[13]) + (red[14] + red[15]);
}

// three sums through one exchange: a and b combined pairwise as block_sum4 does, c left to right as block_sum does (each keeps
// the rounding it had when it was reduced on its own)
__device__ __forceinline__ void block_sum3(double &a, double &b, double &c, double *red, int tid)
{
    const double sa = wave_add(a), sb = wave_add(b), sc = wave_add(c);
    if ((tid & 63) == 0) { const int w = tid >> 6; red[w] = sa; red[4 + w] = sb; red[8 + w] = sc; }
    __syncthreads();
    a = (red[0] + red[1]) + (red[2] + red[3]);
    b = (red[4] + red[5]) + (red[6] + red[7]);
    double t = red[8];
#pragma unroll
    for (int w = 1; w < FT / 64; ++w) t += red[8 + w];
    c = t;
}

// inclusive prefix sum over the 64 lanes of a wavefront (DPP row shifts + row broadcasts)
__device__ __forceinline__ double wave_scan_incl(double v)
{
    v += dpp_f64<0x111, 0xf>(0.0, v);   // row_shr:1
    v += dpp_f64<0x112, 0xf>(0.0, v);   // row_shr:2
    v += dpp_f64<0x114, 0xf>(0.0, v);   // row_shr:4
    v += dpp_f64<0x118, 0xf>(0.0, v);   // row_shr:8
    v += dpp_f64<0x142, 0xa>(0.0, v);   // row_bcast:15 -> rows 1,3
    v += dpp_f64<0x143, 0xc>(0.0, v);   // row_bcast:31 -> rows 2,3
    return v;
}

// the same over each half of the wavefront separately (lanes 0-31 and 32-63): two joints per wavefront when H <= 32
__device__ __forceinline__ double half_scan_incl(double v)
{
    v += dpp_f64<0x111, 0xf>(0.0, v);   // row_shr:1
    v += dpp_f64<0x112, 0xf>(0.0, v);   // row_shr:2
    v += dpp_f64<0x114, 0xf>(0.0, v);   // row_shr:4
    v += dpp_f64<0x118, 0xf>(0.0, v);   // row_shr:8
    v += dpp_f64<0x142, 0xa>(0.0, v);   // row_bcast:15 -> rows 1,3
    return v;
}

// (Bvel v, Bpos v) of the vector in buf[0..HN) -> buf[HN..2HN), buf[2HN..3HN).  Lane i of a wavefront
// owns waypoint i (H <= 64); the workgroup's wavefronts share the joints.  Two DPP prefix sums per joint:
// Bvel v = dt*cumsum(v), Bpos v = dt*cumsum(Bvel v) - dt/2 * Bvel v  (double integrator, robotproperty2.m:136-139).
template <int NJ>
__device__ __forceinline__ void roll_lds(double *buf, int H, double dt, int tid)
{
    const int HN = H * NJ;
    if (H <= 32) {                              // two joints per wavefront, one per half: all NJ <= 6 joints in one round of 3 wavefronts
        const int lane = tid & 31;
        for (int c = tid >> 5; c < NJ; c += 2 * (FT / 64)) {
            const double x = lane < H ? buf[lane * NJ + c] : 0.0;
            const double sv = dt * half_scan_incl(x);
            const double sp = dt * half_scan_incl(sv) - (0.5 * dt) * sv;
            if (lane < H) { buf[HN + lane * NJ + c] = sv; buf[2 * HN + lane * NJ + c] = sp; }
        }
        return;
    }
    const int lane = tid & 63, wv = tid >> 6;
    for (int c = wv; c < NJ; c += FT / 64) {
        const double x = lane < H ? buf[lane * NJ + c] : 0.0;
        const double sv = dt * wave_scan_incl(x);
        const double sp = dt * wave_scan_incl(sv) - (0.5 * dt) * sv;
        if (lane < H) { buf[HN + lane * NJ + c] = sv; buf[2 * HN + lane * NJ + c] = sp; }
    }
}

// inward normal of constraint `code` applied to a vector given as (v, Bvel v, Bpos v) in LDS
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

// slack b - a'x of constraint `code` at the iterate xs = (x, Bvel x, Bpos x); *bnd = its right-hand side
template <int NJ>
__device__ __forceinline__ double slack_of(int code, const double *xs, const double *g, const double *rhs,
                                           const double *lim, const double *v0, const double *maxin, int H, double *bnd)
{
    const int type = code >> 16, i = (code >> 8) & 0xff, jc = code & 0xff, HN = H * NJ;
    switch (type) {
    case CT_COL: {
        const double rh = rhs[jc * H + i];
        double s = rh;
#pragma unroll
        for (int c = 0; c < NJ; ++c) s += g[(jc * H + i) * NJ + c] * xs[2 * HN + i * NJ + c];
        *bnd = rh;
        return s;
    }
    case CT_VELP: { const double bb = lim[jc] - v0[jc]; *bnd = bb; return bb - xs[HN + i * NJ + jc]; }   // CFS_FANUC.m:127
    case CT_VELM: { const double bb = lim[jc] + v0[jc]; *bnd = bb; return bb + xs[HN + i * NJ + jc]; }   // CFS_FANUC.m:129
    case CT_BNDP: { const double bb = maxin[i * NJ + jc]; *bnd = bb; return bb - xs[i * NJ + jc]; }
    default: { const double bb = maxin[i * NJ + jc]; *bnd = bb; return bb + xs[i * NJ + jc]; }
    }
}


// ---- H = I (the PSGCFS projection, PSGCFS_FANUC.m:117): constraint normals in closed form ----------------
// A collision row at waypoint i with gradient g is n[k,c] = g_c ((i-k)+1/2) dt^2 for k <= i (CFS_FANUC.m:121 with
// the double integrator of robotproperty2.m:136-139), a velocity row +-dt for k <= i on its joint.  Products of
// two normals are sums of small half-integers: formed in integer arithmetic, exact in fp64 and symmetric by
// construction, so the active set needs neither H^{-1}N in memory nor the family matrices.
__device__ __forceinline__ double ramp_sum(int a, int m)          // sum_{k=0..m} ((a-k)+1/2)
{
    const int M = m + 1;
    return 0.5 * (double)(M * (2 * a + 1) - m * M);
}
__device__ __forceinline__ double ramp_dot(int a, int b, int m)   // sum_{k=0..m} ((a-k)+1/2)((b-k)+1/2)
{
    const int M = m + 1;
    return 0.25 * (double)(M * (2 * a + 1) * (2 * b + 1) - 2 * (a + b + 1) * m * M + 2 * (m * M * (2 * m + 1) / 3));
}
template <int NJ>
__device__ __forceinline__ double gram_ident(int ca, int cp, const double *g, int H, double dt)   // n_a' n_p
{
    const int ta = ca >> 16, ia = (ca >> 8) & 0xff, ja = ca & 0xff;
    const int tp = cp >> 16, ip = (cp >> 8) & 0xff, jp = cp & 0xff;
    const int m = min(ia, ip);
    if (ta == CT_COL && tp == CT_COL) {
        double dot = 0.0;
#pragma unroll
        for (int c = 0; c < NJ; ++c) dot += g[(ja * H + ia) * NJ + c] * g[(jp * H + ip) * NJ + c];
        return dot * ((dt * dt) * (dt * dt) * ramp_dot(ia, ip, m));
    }
    if (ta == CT_COL) return g[(ja * H + ia) * NJ + jp] * ((tp == CT_VELP ? -1.0 : 1.0) * (dt * dt * dt) * ramp_sum(ia, m));
    if (tp == CT_COL) return g[(jp * H + ip) * NJ + ja] * ((ta == CT_VELP ? -1.0 : 1.0) * (dt * dt * dt) * ramp_sum(ip, m));
    if (ja != jp) return 0.0;
    return ((ta == tp) ? 1.0 : -1.0) * (dt * dt) * (double)(m + 1);
}

// ---- one row of the symmetric inverse Gram matrix P: columns [0,PR) in registers, [PR,QB) in global ---
// tail layout: ptail[(b-PR)*QB + a] (thread a reads consecutive addresses).  Active constraints occupy
// SLOTS of [0,qhi); a freed slot keeps a (numerically) zero row and column, so whole 8-column chunks are
// processed unguarded and no column is ever written through a runtime register index.
template <int PR, int QB>
struct PRow {
    static constexpr int TU = CFS_TU;    // tail columns loaded per batch
    double v[PR];
    __device__ __forceinline__ void zero(double *ptail, int a, int q)
    {
#pragma unroll
        for (int b = 0; b < PR; ++b) v[b] = 0.0;
        for (int b = PR; b < q; ++b) ptail[(b - PR) * QB + a] = 0.0;
    }
    // sum_b P[a][b] * vec[b]
    __device__ __forceinline__ double dot(const double *vec, const double *ptail, int a, int q) const
    {
        double s = 0.0;
#pragma unroll
        for (int b0 = 0; b0 < PR; b0 += 8)
            if (b0 < q) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) s += v[b0 + jj] * vec[b0 + jj];
                __builtin_amdgcn_sched_barrier(0);
            }
        // tail columns live in global scratch (L2): keep TU independent loads in flight, a one-at-a-time loop pays the
        // L2 round trip per column (measured: 45-60 us per step at 120 active rows)
        int b = PR;
        for (; b + TU <= q; b += TU) {
            double t[TU];
#pragma unroll
            for (int jj = 0; jj < TU; ++jj) t[jj] = ptail[(b + jj - PR) * QB + a];
#pragma unroll
            for (int jj = 0; jj < TU; ++jj) s += t[jj] * vec[b + jj];
        }
        for (; b < q; ++b) s += ptail[(b - PR) * QB + a] * vec[b];
        return s;
    }
    // P[a][b] += alpha * vec[b]   (vec zero beyond q)
    __device__ __forceinline__ void axpy(double alpha, const double *vec, double *ptail, int a, int q)
    {
#pragma unroll
        for (int b0 = 0; b0 < PR; b0 += 8)
            if (b0 < q) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[b0 + jj] += alpha * vec[b0 + jj];
                __builtin_amdgcn_sched_barrier(0);
            }
        int b = PR;
        for (; b + TU <= q; b += TU) {
            double t[TU];
#pragma unroll
            for (int jj = 0; jj < TU; ++jj) t[jj] = ptail[(b + jj - PR) * QB + a];
#pragma unroll
            for (int jj = 0; jj < TU; ++jj) ptail[(b + jj - PR) * QB + a] = t[jj] + alpha * vec[b + jj];
        }
        for (; b < q; ++b) ptail[(b - PR) * QB + a] += alpha * vec[b];
    }
    // P[a][b] = (P[a][b] + alpha * vec[b]) * mask[b]   (mask is exactly 0 or 1: annihilates a freed column)
    __device__ __forceinline__ void axpy_mask(double alpha, const double *vec, const double *mask, double *ptail, int a, int q)
    {
#pragma unroll
        for (int b0 = 0; b0 < PR; b0 += 8)
            if (b0 < q) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[b0 + jj] = (v[b0 + jj] + alpha * vec[b0 + jj]) * mask[b0 + jj];
                __builtin_amdgcn_sched_barrier(0);
            }
        int b = PR;
        for (; b + TU <= q; b += TU) {
            double t[TU];
#pragma unroll
            for (int jj = 0; jj < TU; ++jj) t[jj] = ptail[(b + jj - PR) * QB + a];
#pragma unroll
            for (int jj = 0; jj < TU; ++jj) ptail[(b + jj - PR) * QB + a] = (t[jj] + alpha * vec[b + jj]) * mask[b + jj];
        }
        for (; b < q; ++b) ptail[(b - PR) * QB + a] = (ptail[(b - PR) * QB + a] + alpha * vec[b]) * mask[b];
    }
    // P[a][b] = alpha * vec[b]
    __device__ __forceinline__ void set_scaled(double alpha, const double *vec, double *ptail, int a, int q)
    {
#pragma unroll
        for (int b0 = 0; b0 < PR; b0 += 8)
            if (b0 < q) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[b0 + jj] = alpha * vec[b0 + jj];
                __builtin_amdgcn_sched_barrier(0);
            }
        for (int b = PR; b < q; ++b) ptail[(b - PR) * QB + a] = alpha * vec[b];
    }
    // row -> LDS vector (entries [0,q))
    __device__ __forceinline__ void store(double *dst, const double *ptail, int a, int q) const
    {
#pragma unroll
        for (int b0 = 0; b0 < PR; b0 += 8)
            if (b0 < q) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) dst[b0 + jj] = v[b0 + jj];
                __builtin_amdgcn_sched_barrier(0);
            }
        int b = PR;
        for (; b + TU <= q; b += TU) {
            double t[TU];
#pragma unroll
            for (int jj = 0; jj < TU; ++jj) t[jj] = ptail[(b + jj - PR) * QB + a];
#pragma unroll
            for (int jj = 0; jj < TU; ++jj) dst[b + jj] = t[jj];
        }
        for (; b < q; ++b) dst[b] = ptail[(b - PR) * QB + a];
    }
};

struct FusedLayout {      // LDS offsets in doubles, computed identically on host and device
    int rb, ob, x, u, qu, g, rhs, xs, up, wb, zb, d, r, rho, lam, prow, act, fre, prev, flag, slot, code, red, small, mx, racc, cost, ptail, lin, y, total_fixed;
};
__host__ __device__ inline FusedLayout fused_layout(int NJ, int H, int nobs, int QB, int PR)
{
    FusedLayout L;
    const int HN = H * NJ, NS = 2 * NJ;
    int o = 0;
    L.rb = o; o += (int)(sizeof(DevRobot) / 8);
    L.ob = o; o += nobs * 6;
    L.x = o; o += H * NS;
    L.u = o; o += HN;
    L.qu = o; o += HN;
    L.g = o; o += nobs * HN;
    L.rhs = o; o += nobs * H;
    L.xs = o; o += 3 * HN;
    L.up = o; o += HN;                     // Bpos*u of the linearisation point (right sides of the collision rows)
    L.act = o; o += (QB + 1) / 2;
    L.fre = o; o += (QB + 1) / 2;          // stack of freed slots
    L.prev = o; o += (QB + 1) / 2;         // active rows (codes per slot) at the end of the previous QP: the warm start
    L.flag = o; o += (nobs * H + 4 * HN + 7) / 8;
    L.code = o; o += (nobs * H + 4 * HN + 1) / 2;
    L.red = o; o += 64;                    // two exchange buffers of 24 doubles + 12 stamp accumulators
    L.small = o; o += 4 * NJ + nobs;       // lim, v0, theta0 (2NJ), margin
    L.mx = o; o += HN;                     // MAX_input
    L.racc = o; o += HN;                   // how far the input bounds let waypoint (i, c) move: dt^2 sum_k ((i-k)+1/2) MAX_input(k, c)
    L.cost = o; o += (int)(sizeof(DevCost) / 8);   // structure of QQ (handles created from the cost weights)
    o = (o + 1) & ~1;
    L.lin = o;                             // linearisation scratch starts here: it may overwrite the QP's work vectors below
    L.wb = o; o += 3 * HN;
    L.zb = o; o += 3 * HN;
    L.d = o; o += QB;
    L.r = o; o += QB;
    L.rho = o; o += QB;
    L.lam = o; o += QB;
    L.prow = o; o += QB;
    L.slot = o; o += (nobs * H + 4 * HN + 3) / 4;   // constraint -> slot + 1 (0: inactive), ushort; reset at every QP setup
    L.ptail = o;                           // (tail columns of P live in global scratch)
    o = (o + 1) & ~1;
    L.y = o;
    L.total_fixed = o;
    return L;
}

// ------------------------------------------------------------------------------------------------
template <int NJ, int QB, bool IDENT>      // IDENT: the QP Hessian is the identity (PSGCFS projection)
__global__ __launch_bounds__(FT, CFS_WG_PER_CU) void cfs_solve_fused_kernel(FusedParams P)
{
    constexpr int NS = 2 * NJ, NVT = nvt(NJ), NE = 2 * NJ + 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = P.order ? P.order[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;   // workgroups are dispatched in blockIdx order: longest problems first when the caller knows them
    const int H = P.H, nobs = P.nobs, HN = H * NJ, nn = HN, NX = H * NS;
    const double dt = P.dt;
    constexpr int PR = QB < CFS_PR ? QB : CFS_PR;   // register-resident columns of each P row (rest: global scratch)
    const FusedLayout L = fused_layout(NJ, H, nobs, QB, PR);
    DevRobot *rb = reinterpret_cast<DevRobot *>(lds + L.rb);
    double *s_ob = lds + L.ob, *s_x = lds + L.x, *s_u = lds + L.u, *s_qu = lds + L.qu, *s_g = lds + L.g;
    double *s_rhs = lds + L.rhs, *xs = lds + L.xs, *s_up = lds + L.up, *wb = lds + L.wb, *zb = lds + L.zb;
    double *s_d = lds + L.d, *s_r = lds + L.r, *s_rho = lds + L.rho, *s_lam = lds + L.lam, *s_prow = lds + L.prow;
    int *s_act = reinterpret_cast<int *>(lds + L.act);
    int *s_free = reinterpret_cast<int *>(lds + L.fre);
    int *s_prev = reinterpret_cast<int *>(lds + L.prev);
    unsigned char *s_flag = reinterpret_cast<unsigned char *>(lds + L.flag);
    unsigned short *s_slot = reinterpret_cast<unsigned short *>(lds + L.slot);
    int *s_code = reinterpret_cast<int *>(lds + L.code);
    double *red_base = lds + L.red;
    int red_sel = 0;
#define red (red_base + 24 * (red_sel ^= 1))      /* every use is one block-wide reduction: alternate the buffer */
    double *s_lim = lds + L.small, *s_v0 = s_lim + NJ, *s_th0 = s_v0 + NJ, *s_margin = s_th0 + 2 * NJ;
    double *s_mx = lds + L.mx;
    double *s_racc = lds + L.racc;
    // Spill space (columns [PR,QB) of P as [b-PR][a], rows of Y beyond the LDS capacity) comes from a POOL of P.pool_n slots, one per
    // workgroup that can be resident at once (2 per compute unit), not one per problem of the batch: a workgroup takes a free slot
    // when it starts (one atomicCAS by thread 0) and gives it back when it ends.  With fewer slots than resident workgroups a
    // workgroup would only wait for another one to finish (holders never wait for anything).
    int pool_slot;
    {
        int *pubp = reinterpret_cast<int *>(red_base + 63);
        if (tid == 0) {
            // flags are 64 bytes apart (one L2 line each: atomics on neighbours do not serialise); the probe starts at a hash of the
            // workgroup index so that workgroups starting together do not walk the same run of taken slots
            int sidx = (int)(((blockIdx.x * 0x9E3779B1u) >> 8) % (unsigned)P.pool_n);
            // (bounded: there is a slot for every workgroup that can be resident, so a miss means a neighbour has not STARTED releasing
            // yet; a pool whose flags were left set by a launch that never finished must not turn into a spin the GPU never leaves)
            int probes = 0;
            while (atomicCAS(&P.pool_flag[sidx * 16], 0, 1) != 0) {
                sidx = sidx + 1 == P.pool_n ? 0 : sidx + 1;
                __builtin_amdgcn_s_sleep(2);
                if (++probes > (1 << 22)) { sidx = -1; break; }
            }
            pubp[0] = sidx;
        }
        __syncthreads();
        pool_slot = pubp[0];
        if (pool_slot < 0) {                               // reported (status NUMERIC), never waited out
            if (tid == 0 && P.piece == 0) { P.status[b] = CFS_NUMERIC; P.iter_O[b] = 1; P.total_iter[b] = 0; }
            return;
        }
    }
    double *const s_pt = P.Pt + (size_t)pool_slot * P.pt_stride;
    double *s_Y = lds + L.y;                        // QY rows of HN doubles; linearisation scratch in between
    const int QY = P.qy;
    double *const Yg = P.Yg + (size_t)pool_slot * nn * nn;
    auto drop_pool = [&]() {
        __syncthreads();                                   // every thread is done with the slot
        if (tid == 0) { __threadfence(); atomicExch(&P.pool_flag[pool_slot * 16], 0); }
    };
    const int ncon = nobs * H + (P.has_bounds ? 4 : 2) * HN;
    const int maxit = 8 * nn + 200;                  // the oracle's longest certificates take ~2 nn steps
    // constraint codes never change: decoded once into LDS (no integer divisions in the step loop)
    for (int e = tid; e < ncon; e += FT) {
        int code;
        if (e < nobs * H) { const int j = e / H; code = mk_code(CT_COL, e - j * H, j); }
        else {
            const int f2 = e - nobs * H, ty = f2 / HN, k = f2 - ty * HN, i = k / NJ;
            code = mk_code(1 + ty, i, k - i * NJ);
        }
        s_code[e] = code;
    }

    // ---- constructor state (CFS_FANUC.m:55-58, EVAL.m:40-48) ------------------------------------
    {
        const double *src = reinterpret_cast<const double *>(P.rb);
        for (int e = tid; e < (int)(sizeof(DevRobot) / 8); e += FT) lds[L.rb + e] = src[e];
        for (int e = tid; e < nobs * 6; e += FT) s_ob[e] = P.obs[(size_t)b * nobs * 6 + e];
        for (int e = tid; e < NX; e += FT) s_x[e] = P.x_init[(size_t)b * NX + e];
        for (int e = tid; e < HN; e += FT) { s_u[e] = 0.0; s_qu[e] = 0.0; s_mx[e] = P.has_bounds ? P.maxin[e] : 0.0; }
        if (tid < NJ) {
            s_lim[tid] = P.lim[tid];
            s_v0[tid] = P.xR1[(size_t)b * NS + NJ + tid];
            s_th0[tid] = P.xR1[(size_t)b * NS + tid];
        }
        if (tid < nobs) s_margin[tid] = P.margin[tid];
        if (P.cost) {
            const double *csrc = reinterpret_cast<const double *>(P.cost);
            for (int e = tid; e < (int)(sizeof(DevCost) / 8); e += FT) lds[L.cost + e] = csrc[e];
        }
    }
    __syncthreads();
    if (P.has_bounds)
        for (int e = tid; e < HN; e += FT) {
            const int i = e / NJ, c = e - i * NJ;
            double acc = 0.0;
            for (int k = 0; k <= i; ++k) acc += ((double)(i - k) + 0.5) * s_mx[k * NJ + c];
            s_racc[e] = (dt * dt) * acc;
        }
    unsigned long long *s_acc = reinterpret_cast<unsigned long long *>(red_base + 48);   // 12 phase accumulators (developer aid)
    if (tid < 12) s_acc[tid] = 0ull;
    unsigned long long t0_ = P.stamps ? clock64() : 0ull;
    double cost_new = P.caug[b], cost_old = 100000.0;      // get_cost(zeros) = caug; EVAL.m:29
    int iter_O = 1, total_iter = 0, noise_row = 0, status = CFS_OK_MAXITER;
    bool done = false;
    const int nseg = nobs - P.nmesh;                       // obstacles [nseg, nobs) are meshes, linearised by cfs_mesh.hip
    int launched = 0;
    if (P.resume) {                                        // continue a solve that is driven one outer iteration per launch
        for (int e = tid; e < NX; e += FT) s_x[e] = P.x_[(size_t)b * NX + e];
        for (int e = tid; e < HN; e += FT) { s_u[e] = P.u[(size_t)b * nn + e]; s_qu[e] = P.st_qu[(size_t)b * nn + e]; }
        cost_new = P.st_cost[2 * b]; cost_old = P.st_cost[2 * b + 1];
        iter_O = P.iter_O[b]; total_iter = P.total_iter[b]; status = P.status[b];
        noise_row = P.st_noise[b]; done = P.st_done[b] != 0;
        __syncthreads();
    } else if (P.piece == 0) {
        double d2 = 0.0;
        for (int e = tid; e < NX; e += FT) { const double v = s_x[e] - 1.0; d2 += v * v; }   // x_old = ones (EVAL.m:47)
        d2 = block_sum(d2, red, tid);
        if (sqrt(d2) < P.epsilon_O) { done = true; status = CFS_OK_CONVERGED; }
        else if (iter_O > P.max_o_iter) { done = true; status = CFS_OK_MAXITER; }
    } else if (P.piece == 2) {                             // one QP at the caller's linearisation point
        for (int e = tid; e < HN; e += FT) s_u[e] = P.u[(size_t)b * nn + e];
        __syncthreads();
    }

    PRow<PR, QB> Pr;                                       // row `tid` of P = (N'H^{-1}N)^{-1}
    int prev_q = 0;                                        // slots [0, prev_q) of s_prev hold the previous QP's final active rows

    while (!done) {
        // =========================================================================================
        // Start vector of this iteration's QP and the two rollouts it needs -- Bpos*u for the right sides of the collision rows
        // (CFS_FANUC.m:119-120) and (x, Bvel x, Bpos x) of the start point -- depend only on the previous iterate, not on the
        // linearisation: they are computed in four steps that ride along with the first four phases of the linearisation
        // (no barrier of their own; `prep(ph)` below), or in four phases of their own when there is nothing to ride on.
        // =========================================================================================
        bool skip = false;
        if (P.piece == 0 && P.mode == CFS_MODE_PSGCFS) {
            skip = fabs(cost_new - cost_old) < 1e-4;        // stop_inner, MAX_I_ITER = 1 (PSGCFS_FANUC.m:136-142)
            if (!skip) cost_old = cost_new;                 // PSGCFS_FANUC.m:89
        } else if (P.piece == 0) cost_old = cost_new;       // CFS_FANUC.m:67
        const int noise_row_now = noise_row;
        if (P.piece == 0 && P.mode == CFS_MODE_PSGCFS && !skip) ++noise_row;
        auto prep = [&](int ph) {
            if (P.piece == 1) return;
            if (ph == 0) { for (int k = tid; k < HN; k += FT) xs[k] = s_u[k]; }
            else if (ph == 1) roll_lds<NJ>(xs, H, dt, tid);
            else if (ph == 2) {
                const bool psg = P.piece == 0 && P.mode == CFS_MODE_PSGCFS;
                const double sc = (double)iter_O * (double)iter_O + 1.0;
                const bool have = P.noise != nullptr && noise_row_now < P.noise_rows;
                for (int k = tid; k < HN; k += FT) {
                    s_up[k] = xs[2 * HN + k];
                    double v;
                    if (psg && skip) v = s_u[k];            // no step: x_ is the rollout of the unchanged u
                    else if (psg) {
                        const double nz = have ? P.noise[((size_t)b * P.noise_rows + noise_row_now) * nn + k] : 0.0;
                        v = s_u[k] - P.alpha * ((s_qu[k] + P.ff[(size_t)b * nn + k]) + 10.0 * nz / sc);   // PSGCFS_FANUC.m:109
                    } else v = P.x0[(size_t)b * nn + k];    // CFS: -H^{-1} ff; cfs_qp: the caller's start point
                    xs[k] = v;
                }
            } else roll_lds<NJ>(xs, H, dt, tid);
        };
        bool prepped = false;
        // =========================================================================================
        // get_con, distance half (CFS_FANUC.m:110-118): dist -> s_rhs, Diff -> s_g
        // =========================================================================================
        if (nseg > 0) {
            const int W = P.lin_w;
            double *s_sc = lds + L.lin;                      // [W][NJ][3][2] sin, cos of theta, theta+eps/2, theta-eps/2 (minus the joint offset)
            double *s_en = s_sc + W * NJ * 6;                // [W][NVT][6]  capsule end points of every link variant
            double *s_bd = s_en + W * NVT * 6;               // [W][NJ][nobs] base-pose distance of every link
            double *s_dv = s_bd + W * NJ * nseg;             // [W][nobs][NE] min over the links at every evaluation point of num_jac
            unsigned short *s_list = reinterpret_cast<unsigned short *>(s_dv + W * nseg * NE);    // [NJ][W*nobs] (wi << 8 | obstacle)
            int *s_cnt = s_free;                             // candidates per link (the QP's free-slot stack is idle here)
            for (int w0 = 0; w0 < H; w0 += W) {
                const int Wc = min(W, H - w0);
                if (w0 == 0) prep(0);
#if CFS_ANGLE_ADD
                const double ch = cos(FD_EPS / 2), sh = sin(FD_EPS / 2);
                for (int e = tid; e < Wc * NJ; e += FT) {
                    const int m = e % NJ, wi = e / NJ;
                    double sn, cs;
                    sincos(s_x[(w0 + wi) * NS + m] - rb->th_off[m], &sn, &cs);   // dist_arm_3D_200i_2.m:11
                    s_sc[e * 6] = sn;
                    s_sc[e * 6 + 1] = cs;
                    s_sc[e * 6 + 2] = sn * ch + cs * sh; s_sc[e * 6 + 3] = cs * ch - sn * sh;
                    s_sc[e * 6 + 4] = sn * ch - cs * sh; s_sc[e * 6 + 5] = cs * ch + sn * sh;
                }
#else
                for (int e = tid; e < Wc * NJ * 3; e += FT) {
                    const int var = e % 3, m = (e / 3) % NJ, wi = e / (3 * NJ);
                    double x = s_x[(w0 + wi) * NS + m];
                    if (var == 1) x = x + FD_EPS / 2;        // num_jac.m:11
                    else if (var == 2) x = x - FD_EPS / 2;   // num_jac.m:13
                    x = x - rb->th_off[m];                   // dist_arm_3D_200i_2.m:11
                    double sn, cs;
                    sincos(x, &sn, &cs);
                    s_sc[((wi * NJ + m) * 3 + var) * 2] = sn;
                    s_sc[((wi * NJ + m) * 3 + var) * 2 + 1] = cs;
                }
#endif
                __syncthreads();
                // one thread per (waypoint, evaluation point of num_jac): the whole kinematic chain in registers.
                // Evaluation point ev has joint m at +eps/2 if ev == 2m-1, at -eps/2 if ev >= 2m (num_jac.m:8-14:
                // xp is never restored); the shifted sin/cos come from the base pair by angle addition.  Link k of
                // evaluation ev is variant min(ev, 2k); the thread with ev <= 2k is the one that stores it.
                for (int e = tid; e < Wc * NE; e += FT) {
                    const int ev = e % NE, wi = e / NE;
                    double M[12], Mn[12], e6[6];
#pragma unroll
                    for (int k1 = 1; k1 <= NJ; ++k1) {
                        const int avar = (ev == 2 * k1 - 1) ? 1 : (ev >= 2 * k1 ? 2 : 0);
                        const double sn = s_sc[((wi * NJ + k1 - 1) * 3 + avar) * 2], cs = s_sc[((wi * NJ + k1 - 1) * 3 + avar) * 2 + 1];
                        fk_step(rb, k1 - 1, sn, cs, k1 == 1 ? nullptr : M, Mn);
#pragma unroll
                        for (int qq = 0; qq < 12; ++qq) M[qq] = Mn[qq];
                        if (ev <= 2 * k1) {
                            link_ends(rb, k1 - 1, M, e6);
                            double *dstE = s_en + (wi * NVT + kvoff(k1) + ev) * 6;
#pragma unroll
                            for (int qq = 0; qq < 6; ++qq) dstE[qq] = e6[qq];
                        }
                    }
                }
                if (w0 == 0) prep(1);
                __syncthreads();
                STAMP(10);                                  // 10: sincos + link transforms
                // Base-pose distance of every link (dist_arm_3D_200i_2.m:16-26), link index slow so the point /
                // segment branch of distLinSeg is wave-uniform.
#if CFS_LIN_UNROLL
#pragma unroll 2
#endif
                for (int e = tid; e < NJ * Wc * nseg; e += FT) {
                    const int j = e % nseg, wi = (e / nseg) % Wc, k0 = e / (nseg * Wc);
                    s_bd[(wi * NJ + k0) * nseg + j] = seg_seg_dist(s_en + (wi * NVT + kvoff(k0 + 1)) * 6, s_ob + j * 6);
                }
                if (tid < NJ) s_cnt[tid] = 0;
                if (w0 == 0) prep(2);
                __syncthreads();
                // num_jac only needs min over the links at 2nj shifted poses.  A link whose base distance exceeds
                // max(min, 1e-4) by prune_tol can neither become the minimum nor reach the near-zero surrogate at any
                // of them (see DevRobot::prune_tol), so only the other links are evaluated there: same minima, bit for bit.
                for (int e = tid; e < Wc * nseg; e += FT) {
                    const int j = e % nseg, wi = e / nseg;
                    double bk[NJ], m0 = INFINITY;
                    int lk = 0;
#pragma unroll
                    for (int k1 = 1; k1 <= NJ; ++k1) { bk[k1 - 1] = s_bd[(wi * NJ + k1 - 1) * nseg + j]; if (bk[k1 - 1] < m0) { m0 = bk[k1 - 1]; lk = k1; } }   // first minimum wins (dist_arm_3D_200i_2.m:25)
                    if (P.dump_linkid && launched == 0) P.dump_linkid[((size_t)b * nobs + j) * H + w0 + wi] = lk;
                    const double thr = P.no_prune ? INFINITY : fmax(m0, 0.0001) + rb->prune_tol;
#pragma unroll
                    for (int k1 = 1; k1 <= NJ; ++k1)
                        if (bk[k1 - 1] < thr) s_list[(k1 - 1) * W * nseg + atomicAdd(&s_cnt[k1 - 1], 1)] = (unsigned short)((wi << 8) | j);
                    s_dv[e * NE] = m0;
#pragma unroll
                    for (int ev = 1; ev < NE; ++ev) s_dv[e * NE + ev] = INFINITY;
                }
                if (w0 == 0) { prep(3); prepped = true; }
                __syncthreads();
                STAMP(0);                                   // 0: base distances, candidate lists (+ the minima / differences below)
                {
                    int offs[NJ + 1];
                    offs[0] = 0;
#pragma unroll
                    for (int k1 = 1; k1 <= NJ; ++k1) offs[k1] = offs[k1 - 1] + s_cnt[k1 - 1] * 2 * k1;
#if CFS_LIN_UNROLL
#pragma unroll 2
#endif
                    for (int e = tid; e < offs[NJ]; e += FT) {
                        int k1 = 1, ent = 0, v = 1;
#pragma unroll
                        for (int kk = 1; kk <= NJ; ++kk)
                            if (e >= offs[kk - 1] && e < offs[kk]) {
                                const int r_ = e - offs[kk - 1];
                                k1 = kk; ent = r_ / (2 * kk); v = r_ - ent * (2 * kk) + 1;
                            }
                        const int item = s_list[(k1 - 1) * W * nseg + ent], wi = item >> 8, j = item & 255;
                        const double dis = seg_seg_dist(s_en + (wi * NVT + kvoff(k1) + v) * 6, s_ob + j * 6);
                        // link k1 is at variant min(ev, 2 k1) at evaluation point ev: v < 2 k1 serves ev = v, v = 2 k1 every ev >= v
                        double *dv = s_dv + (wi * nseg + j) * NE;
                        const int evhi = (v == 2 * k1) ? NE - 1 : v;
                        for (int ev = v; ev <= evhi; ++ev)
                            __hip_atomic_fetch_min(dv + ev, dis, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_min_f64
                    }
                }
                __syncthreads();
                STAMP(11);                                  // 11: segment pairs of the shifted poses
                for (int e = tid; e < Wc * nseg; e += FT) {
                    const int j = e % nseg, wi = e / nseg;
                    double dev[NE];
#pragma unroll
                    for (int ev = 0; ev < NE; ++ev) dev[ev] = s_dv[e * NE + ev];
                    s_rhs[j * H + w0 + wi] = dev[0];
#pragma unroll
                    for (int m = 0; m < NJ; ++m) s_g[(j * H + w0 + wi) * NJ + m] = (dev[2 * m + 1] - dev[2 * m + 2]) / FD_EPS;
                }
                // (no barrier between tiles: the next tile's first phase only writes its sin / cos table, which these differences do
                // not read; its barrier covers both)
                if (w0 + W >= H) __syncthreads();
            }
        }
        if (!prepped && P.piece != 1) {                     // nothing to ride on (mesh obstacles only, or the QP piece): four phases of their own
            prep(0); __syncthreads(); prep(1); __syncthreads(); prep(2); __syncthreads(); prep(3); __syncthreads();
        }

        if (P.nmesh > 0) {   // rows of the mesh obstacles: linearised at this iterate by cfs_linearize_mesh_kernel (cfs_mesh.hip)
            for (int e = tid; e < P.nmesh * H; e += FT) s_rhs[nseg * H + e] = P.ext_dist[(size_t)b * P.nmesh * H + e];
            for (int e = tid; e < P.nmesh * HN; e += FT) s_g[nseg * HN + e] = P.ext_grad[(size_t)b * P.nmesh * HN + e];
            __syncthreads();
        }

        if (P.dump_dist && launched == 0) {                 // the linearisation as cfs_linearize / cfs_get_con return it
            for (int e = tid; e < nobs * H; e += FT) P.dump_dist[(size_t)b * nobs * H + e] = s_rhs[e];
            for (int e = tid; e < nobs * HN; e += FT) P.dump_grad[(size_t)b * nobs * HN + e] = s_g[e];
        }
        if (P.piece == 1) { drop_pool(); return; }

        // =========================================================================================
        // the QP of this outer iteration (CFS_FANUC.m:85 | PSGCFS_FANUC.m:106-128)
        // =========================================================================================
        STAMP(0);                                           // 0: linearisation
        int qp_status = QP_OK, iters = 0;
        int qhi = 0, nfree = 0;                             // slots in use: [0,qhi) minus the free stack
        int npolish = 0;
        // zb = base - sum_a coef[a] * Y[a]   (Y rows [0,QY) in LDS, the rest in global scratch; base == nullptr: zero)
        // IDENT: zb = (z, Bvel z, Bpos z) with z = base - N coef, N coef in closed form: the active rows of waypoint i
        // contribute Rp[i] = sum_j coef g_j (position space) and Rv[i] = -+dt coef (velocity space); N coef =
        // Bpos' Rp + Rv summed over i >= k = suffix sums, done as DPP prefix scans over reversed lanes (lane <-> waypoint
        // H-1-lane); base == nullptr means w = n_p, evaluated on the fly.  No barrier inside: the caller syncs before and after.
        auto n_combine = [&](const double *coef, const double *base, int pc_) {
            const int ptype_ = pc_ >> 16, pi_ = (pc_ >> 8) & 0xff, pj_ = pc_ & 0xff;
            const bool two = H <= 32;                                   // two joints per wavefront, one per half
            const int lane = two ? (tid & 31) : (tid & 63);             // waypoint of the rollout part
            const int iR = H - 1 - lane;                                // waypoint of the suffix part (reversed lanes)
            const int cstep = two ? 2 * (FT / 64) : FT / 64;
            for (int c = two ? (tid >> 5) : (tid >> 6); c < NJ; c += cstep) {
                double Rp = 0.0, Rv = 0.0;
                if (lane < H) {
                    const int sp_ = s_slot[nobs * H + iR * NJ + c], sm_ = s_slot[nobs * H + HN + iR * NJ + c];
                    for (int j0 = 0; j0 < nobs; j0 += 8) {              // slot lookups of eight obstacles in flight
                        int sl[8];
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) sl[jj] = j0 + jj < nobs ? s_slot[(j0 + jj) * H + iR] : 0;
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj)
                            if (sl[jj]) Rp += coef[sl[jj] - 1] * s_g[((j0 + jj) * H + iR) * NJ + c];
                    }
                    if (sp_) Rv -= dt * coef[sp_ - 1];
                    if (sm_) Rv += dt * coef[sm_ - 1];
                }
                // (N coef)[k] = sum_{i>=k} (((i-k)+1/2) dt^2 Rp[i] + Rv[i]) = dt^2 (S2 - S1/2) + V1 with S1 = suffix(Rp), S2 = suffix(S1),
                // V1 = suffix(Rv): two scans, the second one carries dt^2 S1 + Rv
                const double S1 = two ? half_scan_incl(Rp) : wave_scan_incl(Rp);
                const double cmb = (dt * dt) * S1 + Rv;
                const double S2V = two ? half_scan_incl(cmb) : wave_scan_incl(cmb);
                const double nr_rev = S2V - (0.5 * dt * dt) * S1;       // (N coef)[H-1-lane, c]
                const int src = (lane < H ? H - 1 - lane : lane) + (two ? (tid & 32) : 0);
                const double nr = __shfl(nr_rev, src, 64);
                double z0 = 0.0;
                if (lane < H) {
                    if (base) z0 = base[lane * NJ + c];
                    else if (ptype_ == CT_COL) z0 = lane <= pi_ ? s_g[(pj_ * H + pi_) * NJ + c] * ((dt * dt) * ((double)(pi_ - lane) + 0.5)) : 0.0;
                    else z0 = (c == pj_ && lane <= pi_) ? (ptype_ == CT_VELP ? -dt : dt) : 0.0;
                    z0 -= nr;
                }
                const double sv = dt * (two ? half_scan_incl(z0) : wave_scan_incl(z0));
                const double spo = dt * (two ? half_scan_incl(sv) : wave_scan_incl(sv)) - (0.5 * dt) * sv;
                if (lane < H) { zb[lane * NJ + c] = z0; zb[HN + lane * NJ + c] = sv; zb[2 * HN + lane * NJ + c] = spo; }
            }
        };
        auto y_combine = [&](const double *coef, const double *base) {
            const int qa = min(qhi, QY);
            for (int k = tid; k < HN; k += FT) {
                double z0 = base ? base[k] : 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
                int a = 0;
                for (; a + 4 <= qa; a += 4) {               // four independent chains keep the LDS pipe full
                    z0 -= coef[a] * s_Y[a * HN + k];
                    z1 -= coef[a + 1] * s_Y[(a + 1) * HN + k];
                    z2 -= coef[a + 2] * s_Y[(a + 2) * HN + k];
                    z3 -= coef[a + 3] * s_Y[(a + 3) * HN + k];
                }
                for (; a < qa; ++a) z0 -= coef[a] * s_Y[a * HN + k];
                for (a = qa; a + CFS_TU <= qhi; a += CFS_TU) {   // rows beyond the LDS capacity: CFS_TU L2 loads in flight
                    double t[CFS_TU];
#pragma unroll
                    for (int jj = 0; jj < CFS_TU; ++jj) t[jj] = Yg[(size_t)(a + jj) * nn + k];
#pragma unroll
                    for (int jj = 0; jj < CFS_TU; jj += 4) {
                        z0 -= coef[a + jj] * t[jj]; z1 -= coef[a + jj + 1] * t[jj + 1]; z2 -= coef[a + jj + 2] * t[jj + 2]; z3 -= coef[a + jj + 3] * t[jj + 3];
                    }
                }
                for (; a < qhi; ++a) z0 -= coef[a] * Yg[(size_t)a * nn + k];
                zb[k] = (z0 + z1) + (z2 + z3);
            }
        };
        // the same with the rollouts (Bvel z, Bpos z) fused in: thread = (waypoint, joint), two joints per wavefront, scans in
        // registers -- one barrier less per pass.  Only while every Y row is in LDS (the strided mapping would uncoalesce Yg).
        auto y_combine_roll = [&](const double *coef, const double *base) {
            const int lane = tid & 31;
            for (int c = tid >> 5; c < NJ; c += 2 * (FT / 64)) {
                const int k = lane * NJ + c;
                double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
                if (lane < H) {
                    z0 = base ? base[k] : 0.0;
                    int a = 0;
                    for (; a + 4 <= qhi; a += 4) {
                        z0 -= coef[a] * s_Y[a * HN + k];
                        z1 -= coef[a + 1] * s_Y[(a + 1) * HN + k];
                        z2 -= coef[a + 2] * s_Y[(a + 2) * HN + k];
                        z3 -= coef[a + 3] * s_Y[(a + 3) * HN + k];
                    }
                    for (; a < qhi; ++a) z0 -= coef[a] * s_Y[a * HN + k];
                }
                const double z = (z0 + z1) + (z2 + z3);
                const double sv = dt * half_scan_incl(z);
                const double spo = dt * half_scan_incl(sv) - (0.5 * dt) * sv;
                if (lane < H) { zb[k] = z; zb[HN + k] = sv; zb[2 * HN + k] = spo; }
            }
        };
        // wb = (w, Bvel w, Bpos w) for w = H^{-1} n_code (H = QQ): a gather of <= NJ columns of the family matrices and of their
        // precomputed rollouts -- no prefix sums; the caller puts a barrier behind it
        auto gather_w = [&](int code) {
            const int ptype_ = code >> 16, pi_ = (code >> 8) & 0xff, pj_ = code & 0xff;
            for (int k = tid; k < HN; k += FT) {
                double w0, w1, w2;
                if (ptype_ == CT_COL) {
                    w0 = w1 = w2 = 0.0;
#pragma unroll
                    for (int cs = 0; cs < NJ; ++cs) {
                        const double gc = s_g[(pj_ * H + pi_) * NJ + cs];
                        const size_t o = (size_t)(pi_ * NJ + cs) * nn + k;
                        w0 += gc * P.M1[o];
                        if (!(P.opt & 1)) { w1 += gc * P.M1v[o]; w2 += gc * P.M1p[o]; }
                    }
                } else {
                    const bool vel = ptype_ == CT_VELP || ptype_ == CT_VELM;
                    const double sg = (ptype_ == CT_VELP || ptype_ == CT_BNDP) ? -1.0 : 1.0;
                    const size_t o = (size_t)(pi_ * NJ + pj_) * nn + k;
                    w0 = sg * (vel ? P.M2 : P.M3)[o];
                    w1 = w2 = 0.0;
                    if (!(P.opt & 1)) { w1 = sg * (vel ? P.M2v : P.M3v)[o]; w2 = sg * (vel ? P.M2p : P.M3p)[o]; }
                }
                wb[k] = w0; wb[HN + k] = w1; wb[2 * HN + k] = w2;
            }
        };
        if (!skip) {
            // rhs = (d - margin) - Diff'*Bj(1:nj,:)*u   (CFS_FANUC.m:119-120), with Bpos*u from the rollout of u (s_up)
            for (int e = tid; e < ncon; e += FT) { s_flag[e] = 0; s_slot[e] = 0; }
            Pr.zero(s_pt, tid, 0);
            if (tid < QB) { s_prev[tid] = s_act[tid]; s_d[tid] = 0.0; s_r[tid] = 0.0; s_rho[tid] = 0.0; s_prow[tid] = 0.0; s_lam[tid] = 0.0; s_act[tid] = -1; }
            for (int e = tid; e < nobs * H; e += FT) {
                const int j = e / H, i = e - j * H;
                double gp = 0.0;
#pragma unroll
                for (int c = 0; c < NJ; ++c) gp += s_g[e * NJ + c] * s_up[i * NJ + c];
                s_rhs[e] = (s_rhs[e] - s_margin[j]) - gp;
            }
            // (the sums of the infeasibility bound below share this phase: their exchange is the phase's one barrier)
            // Rigorous early infeasibility test.  At step 1 the iterate minimises the objective over a
            // subset of the constraints, so its objective value never exceeds the optimum, which in turn
            // is at most max{f(v): v in the box} <= f(x0) + lambda_max/2 (R + |x0|)^2, the box being
            // +-MAX_input (CFS) or what the velocity limits allow per step (PSGCFS).  A diverging dual
            // iterate (infeasible QP) crosses that bound long before fp64 loses the plot.
            double fgain = 0.0, fbound;
            {
                // (i) input box (CFS) / per-step velocity increments (PSGCFS) in u-space with lambda_max(H);
                // (ii) velocity box in s = Bvel*u space, where f - f(x0) = 1/2 (s-s0)'G(s-s0), G = D'HD/dt^2 and every
                //      feasible s obeys |v0 + s| <= lim  (CFS_FANUC.m:126-129) -- ~100x tighter for the drivers' cost
                double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
                for (int k = tid; k < HN; k += FT) {
                    const int c = k % NJ;
                    const double rb_ = P.has_bounds ? s_mx[k] : (2.0 * s_lim[c] + fabs(s_v0[c])) / dt;
                    const double rs_ = s_lim[c] + fabs(s_v0[c]);
                    p0 += rb_ * rb_;
                    p1 += xs[k] * xs[k];
                    p2 += rs_ * rs_;
                    p3 += xs[HN + k] * xs[HN + k];
                }
                block_sum4(p0, p1, p2, p3, red, tid);
                const double lmax = P.lmax_H;
                const double ru = sqrt(p0) + sqrt(p1), rs = sqrt(p2) + sqrt(p3);
                fbound = 1.0001 * 0.5 * fmin(lmax * ru * ru, P.lmax_vel * rs * rs);
            }
            // Infeasibility certificate without a step (CFS_FANUC).  The velocity rows |v0 + Bvel u| <= lim (CFS_FANUC.m:126-129)
            // confine the position offset of waypoint i to the box dt (i + 1/2) [-lim - v0, lim - v0] (trapezoid rule of the double
            // integrator), the input bounds to +-racc.  Two collision rows of one waypoint whose normalised sum (weights 1 : w)
            // cannot reach its right side anywhere in that box -- an arm squeezed between two obstacles -- prove the QP
            // infeasible (Farkas, multipliers on the two rows and on box faces); so do a row of waypoint i and one of waypoint
            // i + 1 against one trapezoid step.  Config 3: 218 of the 327 infeasible linearisations, a quarter of all QP steps,
            // +20 % throughput.  Asked only once a QP has taken CERT_AT steps without finishing (ordinary QPs end in 1-5 steps
            // and never pay for it).  H = I (PSGCFS; no input bounds, so the boxes are the velocity ones alone): +-0 when it was first
            // tried in round 2, +1.3 % on the round-3 kernel (1.435 -> 1.416 ms per solve, same-call A/B, results bit-identical but for
            // the step counts; same registers and scratch), so it is compiled in for both Hessians.
            const bool cert_on = !(P.opt & 16) && nobs > 1;   // whole solves and the QP piece (cfs_qp) alike
            auto certificate = [&]() -> bool {
                int hit = 0;
                for (int e = tid; e < H * nobs; e += FT) {
                    const int i = e / nobs, a = e - i * nobs;
                    double ga[NJ], cen[NJ], rad[NJ], na2 = 0.0;
#pragma unroll
                    for (int c = 0; c < NJ; ++c) { ga[c] = s_g[(a * H + i) * NJ + c]; na2 += ga[c] * ga[c]; }
                    if (!(na2 > 0.0)) continue;
                    const double ia = 1.0 / sqrt(na2), ra = -s_rhs[a * H + i] * ia;
#pragma unroll
                    for (int c = 0; c < NJ; ++c) {
                        const double f = dt * ((double)i + 0.5);
                        double lo = f * (-s_lim[c] - s_v0[c]), hi = f * (s_lim[c] - s_v0[c]);
                        if (P.has_bounds) { lo = fmax(lo, -s_racc[i * NJ + c]); hi = fmin(hi, s_racc[i * NJ + c]); }
                        cen[c] = 0.5 * (lo + hi); rad[c] = fmax(0.5 * (hi - lo), 0.0);
                    }
                    for (int b2 = a + 1; b2 < nobs; ++b2) {
                        double gb[NJ], nb2 = 0.0;
#pragma unroll
                        for (int c = 0; c < NJ; ++c) { gb[c] = s_g[(b2 * H + i) * NJ + c]; nb2 += gb[c] * gb[c]; }
                        if (!(nb2 > 0.0)) continue;
                        const double ib = 1.0 / sqrt(nb2), rb = -s_rhs[b2 * H + i] * ib;
#pragma unroll
                        for (int wk = 0; wk < 3; ++wk) {
                            const double w = wk == 0 ? 1.0 : (wk == 1 ? 0.5 : 2.0);
                            double lhs = 0.0, mag = 0.0;
#pragma unroll
                            for (int c = 0; c < NJ; ++c) {
                                const double cc = ga[c] * ia + w * (gb[c] * ib);
                                lhs += cc * cen[c] + fabs(cc) * rad[c];
                                mag += fabs(cc) * (fabs(cen[c]) + rad[c]);
                            }
                            const double rhs_ = ra + w * rb;
                            if (lhs < rhs_ - 1e-9 * (1.0 + fabs(rhs_) + mag)) hit = 1;
                        }
                    }
                    // ... and a row of the NEXT waypoint: P_{i+1} = P_i + D with D in dt [-lim - v0, lim - v0] (one trapezoid step), so
                    // n_a.P_i + w n_b.P_{i+1} <= max over the box of (n_a + w n_b).P_i + max over the step of w n_b.D.  An arm that
                    // would have to jump between two consecutive waypoints: 57 % of config 3's infeasible linearisations.
                    if (i + 1 < H)
                        for (int b2 = 0; b2 < nobs; ++b2) {
                            double gb[NJ], nb2 = 0.0;
#pragma unroll
                            for (int c = 0; c < NJ; ++c) { gb[c] = s_g[(b2 * H + i + 1) * NJ + c]; nb2 += gb[c] * gb[c]; }
                            if (!(nb2 > 0.0)) continue;
                            const double ib = 1.0 / sqrt(nb2), rb = -s_rhs[b2 * H + i + 1] * ib;
#pragma unroll
                            for (int wk = 0; wk < 3; ++wk) {
                                const double w = wk == 0 ? 1.0 : (wk == 1 ? 0.5 : 2.0);
                                double lhs = 0.0, mag = 0.0;
#pragma unroll
                                for (int c = 0; c < NJ; ++c) {
                                    const double cb = w * (gb[c] * ib), cc = ga[c] * ia + cb;
                                    lhs += (cc * cen[c] + fabs(cc) * rad[c]) + (cb * (-dt * s_v0[c]) + fabs(cb) * (dt * s_lim[c]));
                                    mag += fabs(cc) * (fabs(cen[c]) + rad[c]) + fabs(cb) * dt * (fabs(s_v0[c]) + s_lim[c]);
                                }
                                const double rhs_ = ra + w * rb;
                                if (lhs < rhs_ - 1e-9 * (1.0 + fabs(rhs_) + mag)) hit = 1;
                            }
                        }
                }
                return block_sum((double)hit, red, tid) > 0.0;     // (no __syncthreads_or: it would add static LDS to a kernel sized to the byte)
            };
            STAMP(1);                                       // 1: QP setup
            // ---- warm start (H = I): begin at the S-pair of the previous outer iteration's active rows ------------------------
            // Consecutive outer iterations linearise nearly the same trajectory, so the optimal active set barely changes (the
            // long problems of config 3: 8-12 rows, 80-100 % kept), yet a cold dual active set re-adds every row one full step
            // at a time.  Goldfarb-Idnani may start from ANY S-pair (x minimises over the equality-constrained rows A, their
            // multipliers are >= 0): build P for the previous rows -- Gram products only, wavefront 0 alone, no scan, no combine,
            // no block barrier --, take lambda = -P s(x0), drop rows with lambda <= 0 until none is left, set x = x0 + N lambda.
            // The optimum is the same (strictly convex QP); what changes is the number of steps.
            if (P.piece == 0 && !(P.opt & 8) && prev_q > 0 && prev_q <= min(64, P.warm_max > 0 ? P.warm_max : min(PR, IDENT ? 64 : 24))) {
                int *pub = reinterpret_cast<int *>(red_base + 62);
                int q = 0;                                   // H = QQ: tracked by every thread (one barrier per row); H = I: by wavefront 0
                for (int s0 = 0; s0 < (IDENT ? 1 : prev_q); ++s0) {
                    int c_all = -1;
                    if (!IDENT) {                            // every thread gathers w = H^{-1} n_c; wavefront 0 decides; Y[slot] = w if taken
                        c_all = s_prev[s0];
                        if (c_all < 0) continue;
                        gather_w(c_all);
                        __syncthreads();
                        if (P.opt & 1) { roll_lds<NJ>(wb, H, dt, tid); __syncthreads(); }
                    }
                if (tid < 64) {
                    for (int s1 = IDENT ? 0 : s0; s1 < (IDENT ? prev_q : s0 + 1); ++s1) {
                        const int c = s_prev[s1];
                        if (c < 0) continue;
                        const int mine = tid < q ? s_act[tid] : -1;
                        const double dv = mine >= 0 ? (IDENT ? gram_ident<NJ>(mine, c, s_g, H, dt) : ndot<NJ>(mine, wb, s_g, H)) : 0.0;
                        s_d[tid] = dv;
                        const double spp_ = IDENT ? gram_ident<NJ>(c, c, s_g, H, dt) : ndot<NJ>(c, wb, s_g, H);
                        if (!IDENT && tid == 0) pub[0] = 0;
                        sync_rows(true);
                        const double rv = mine >= 0 ? Pr.dot(s_d, s_pt, tid, q) : 0.0;
                        const double delta_ = spp_ - wave_add(dv * rv);
                        if (!(delta_ > 1e-6 * spp_)) { sync_rows(true); continue; }   // (nearly) dependent on the rows taken so far: left to the main loop
                        const int slot = q, qn = q + 1;
                        const double inv = 1.0 / delta_;
                        s_r[tid] = tid == slot ? -1.0 : rv;
                        if (slot >= PR) s_pt[(slot - PR) * QB + tid] = 0.0;            // fresh tail column, this wavefront's rows
                        sync_rows(true);
                        if (tid < q) Pr.axpy(rv * inv, s_r, s_pt, tid, qn);
                        else if (tid == slot) Pr.set_scaled(-inv, s_r, s_pt, tid, qn);
                        if (tid == 0) {
                            const int ct = c >> 16, ci = (c >> 8) & 0xff, cj = c & 0xff;
                            const int cidx = ct == CT_COL ? cj * H + ci : nobs * H + (ct - 1) * HN + ci * NJ + cj;
                            s_act[slot] = c; s_flag[cidx] = 1; s_slot[cidx] = (unsigned short)(slot + 1);
                            if (!IDENT) pub[0] = 1;
                        }
                        if (IDENT) q = qn;
                        sync_rows(true);
                    }
                }
                    if (!IDENT) {
                        __syncthreads();
                        if (pub[0]) {
                            if (q < QY) { for (int k = tid; k < HN; k += FT) s_Y[q * HN + k] = wb[k]; }
                            else { for (int k = tid; k < HN; k += FT) Yg[(size_t)q * nn + k] = wb[k]; }
                            ++q;
                        }
                    }
                }
                if (tid < 64) {
                    int nf = 0;
                    double lamv = 0.0, s0v = 0.0;
                    bool ok = true;
                    for (int round = 0;; ++round) {
                        const int mine = tid < q ? s_act[tid] : -1;
                        double bb;
                        s0v = mine >= 0 ? slack_of<NJ>(mine, xs, s_g, s_rhs, s_lim, s_v0, s_mx, H, &bb) : 0.0;
                        s_d[tid] = s0v;
                        sync_rows(true);
                        lamv = mine >= 0 ? -Pr.dot(s_d, s_pt, tid, q) : 0.0;
                        const double mn = wave_min(mine >= 0 ? lamv : INFINITY);
                        if (!(mn <= 0.0)) break;                                       // every multiplier positive (or no row left)
                        if (round == 12) { ok = false; break; }
                        const unsigned long long hit = __ballot(mine >= 0 && lamv == mn);
                        const int l = (int)__builtin_ctzll(hit);
                        sync_rows(true);
                        if (tid == l) Pr.store(s_prow, s_pt, tid, q);
                        s_rho[tid] = tid == l ? 0.0 : 1.0;
                        sync_rows(true);
                        const double pll = s_prow[l];
                        if (mine >= 0 && tid != l) Pr.axpy_mask(-(s_prow[tid] / pll), s_prow, s_rho, s_pt, tid, q);
                        if (tid == l) Pr.zero(s_pt, tid, q);
                        if (tid == 0) {
                            const int gone = s_act[l], gt = gone >> 16, gi = (gone >> 8) & 0xff, gj = gone & 0xff;
                            const int gidx = gt == CT_COL ? gj * H + gi : nobs * H + (gt - 1) * HN + gi * NJ + gj;
                            s_act[l] = -1; s_free[nf] = l; s_flag[gidx] = 0; s_slot[gidx] = 0;
                        }
                        ++nf;
                        sync_rows(true);
                    }
                    if (!ok) {                                                         // give up: cold start
                        const int mine = tid < q ? s_act[tid] : -1;
                        if (mine >= 0) {
                            const int gt = mine >> 16, gi = (mine >> 8) & 0xff, gj = mine & 0xff;
                            const int gidx = gt == CT_COL ? gj * H + gi : nobs * H + (gt - 1) * HN + gi * NJ + gj;
                            s_flag[gidx] = 0; s_slot[gidx] = 0; s_act[tid] = -1;
                        }
                        Pr.zero(s_pt, tid, 0);
                        q = 0; nf = 0; lamv = 0.0;
                    }
                    const int mine = tid < q ? s_act[tid] : -1;
                    s_lam[tid] = mine >= 0 ? lamv : 0.0;
                    s_r[tid] = mine >= 0 ? -lamv : 0.0;                                // combine below: zb = x0 - N(-lambda)
                    const double fg = -0.5 * wave_add(mine >= 0 ? lamv * s0v : 0.0);  // f(x) - f(x0) = 1/2 lambda'G lambda = -1/2 lambda's(x0)
                    if (tid == 0) { pub[0] = q; pub[1] = nf; red_base[61] = fg; }
                }
                __syncthreads();
                qhi = pub[0]; nfree = pub[1]; fgain = red_base[61];
                iters = qhi - nfree;
                if (qhi > nfree) {
                    if (IDENT) n_combine(s_r, xs, 0);        // zb = (x, Bvel x, Bpos x), x = x0 + N lambda
                    else {                                   // x = x0 + Y lambda
                        y_combine(s_r, xs);
                        __syncthreads();
                        roll_lds<NJ>(zb, H, dt, tid);
                    }
                    __syncthreads();
                    for (int k = tid; k < 3 * HN; k += FT) xs[k] = zb[k];
                    __syncthreads();
                }
            }
            STAMP(3);                                       // 3: warm start (H = I) | w gather + rollout (H = QQ, inside the steps)
            int nloop = 0;
            for (;;) {
                if (fgain > fbound) { qp_status = QP_INFEASIBLE; break; }
                if (cert_on && nloop == CERT_AT && certificate()) { qp_status = QP_INFEASIBLE; break; }
                ++nloop;
                // step 1: most violated constraint (constraints are strided over the threads; the codes of a
                // thread's constraints never change, so they are decoded once per kernel -- no integer divisions here)
                double sbest = 0.0;
                int cbest = 0x7fffffff;
                for (int e = tid; e < nobs * H; e += FT) {      // collision rows: e = j*H + i
                    if (s_flag[e]) continue;
                    const int code = s_code[e], i = (code >> 8) & 0xff;
                    const double rh = s_rhs[e];
                    double sl = rh;
#pragma unroll
                    for (int c = 0; c < NJ; ++c) sl += s_g[e * NJ + c] * xs[2 * HN + i * NJ + c];
                    if (sl < -1e-11 * (1.0 + fabs(rh)) && sl < sbest) { sbest = sl; cbest = code; }
                }
                for (int k = tid; k < HN; k += FT) {            // velocity rows (and input bounds) of (waypoint, joint) k: both signs at once
                    const int i = k / NJ, c = k - i * NJ;
                    const double v = xs[HN + k], bp = s_lim[c] - s_v0[c], bm = s_lim[c] + s_v0[c];   // CFS_FANUC.m:127, :129
                    const double sp_ = bp - v, sm_ = bm + v;
                    if (!s_flag[nobs * H + k] && sp_ < -1e-11 * (1.0 + fabs(bp)) && sp_ < sbest) { sbest = sp_; cbest = mk_code(CT_VELP, i, c); }
                    if (!s_flag[nobs * H + HN + k] && sm_ < -1e-11 * (1.0 + fabs(bm)) && sm_ < sbest) { sbest = sm_; cbest = mk_code(CT_VELM, i, c); }
                    if (P.has_bounds) {
                        const double bb = s_mx[k], u_ = xs[k];
                        const double s1 = bb - u_, s2 = bb + u_;
                        if (!s_flag[nobs * H + 2 * HN + k] && s1 < -1e-11 * (1.0 + fabs(bb)) && s1 < sbest) { sbest = s1; cbest = mk_code(CT_BNDP, i, c); }
                        if (!s_flag[nobs * H + 3 * HN + k] && s2 < -1e-11 * (1.0 + fabs(bb)) && s2 < sbest) { sbest = s2; cbest = mk_code(CT_BNDM, i, c); }
                    }
                }
                block_argmin(sbest, cbest, red, tid);
                STAMP(2);                                   // 2: step 1 (slack scan + argmin)
                bool polish = false;
                if (cbest == 0x7fffffff) {                   // primal feasible: optimum, up to the drift of its active rows
                    // Every step keeps x - x0 = H^{-1} N lambda exactly, but the refined direction z still leaks ~1e-9 |d| into
                    // the active normals, and t reaches 1e5..1e8: over ~100 steps of a near-degenerate QP the active rows drift
                    // off their bounds (problem 939 of config 3: 3e-4 relative in u).  Project back, G c = -s through the
                    // running inverse: x <- x - Y (P s), lambda <- lambda - P s.  Stationarity is untouched (x moves along Y,
                    // lambda with it).  The projection runs through the refinement code below (same P row product, same
                    // combination with Y, same rollout), then the rows are scanned again.
                    if (qhi == nfree || npolish >= 3) break;
                    double drift = 0.0;
                    if (tid < qhi) {
                        const int ac = s_act[tid];
                        double sa = 0.0, bb = 0.0;
                        if (ac >= 0) { sa = slack_of<NJ>(ac, xs, s_g, s_rhs, s_lim, s_v0, s_mx, H, &bb); drift = fabs(sa) / (1.0 + fabs(bb)); }
                        s_prow[tid] = sa;
                    }
                    { double nd = -drift; int dm = 0; block_argmin(nd, dm, red, tid); drift = -nd; }   // block-wide maximum
                    if (!(drift > P.polish_tol)) break;
                    for (int k = tid; k < HN; k += FT) zb[k] = xs[k];
                    __syncthreads();
                    polish = true;
                }
                const int pc = cbest, ptype = pc >> 16, pi = (pc >> 8) & 0xff, pj = pc & 0xff;
                const int pidx = ptype == CT_COL ? pj * H + pi : nobs * H + (ptype - 1) * HN + pi * NJ + pj;
                double sp = sbest, lam_p = 0.0;
                bool have_w = false;                        // wb = (w, Bvel w, Bpos w) of the entering row survives a partial step: gathered once per row

                // step 2
                for (;;) {
                    if (!polish && ++iters > maxit) { qp_status = QP_NUMERIC; break; }
                    double spp = 0.0;
                    const int myact = tid < qhi ? s_act[tid] : -1;
                    if (!polish && IDENT) {
                        // H = I: w = n_p; d = N'n_p and n_p'n_p straight from the closed form (exactly symmetric products)
                        spp = gram_ident<NJ>(pc, pc, s_g, H, dt);
                        if (tid < qhi) s_d[tid] = myact >= 0 ? gram_ident<NJ>(myact, pc, s_g, H, dt) : 0.0;
                        sync_rows(qhi <= 64);
                        if (tid < qhi) s_r[tid] = myact >= 0 ? Pr.dot(s_d, s_pt, tid, qhi) : 0.0;
                        __syncthreads();
                    }
                    if (!polish && !IDENT) {
                    if (!have_w) {
                    gather_w(pc);                          // w = H^{-1} n_p with its rollouts
                    __syncthreads();
                    if (P.opt & 1) { roll_lds<NJ>(wb, H, dt, tid); __syncthreads(); }
                    have_w = true;
                    }
                    STAMP(3);                               // 3: w gather + rollout
                    spp = ndot<NJ>(pc, wb, s_g, H);
                    if (tid < qhi) s_d[tid] = myact >= 0 ? ndot<NJ>(myact, wb, s_g, H) : 0.0;
                    sync_rows(qhi <= 64);
                    // r = P d
                    if (tid < qhi) s_r[tid] = myact >= 0 ? Pr.dot(s_d, s_pt, tid, qhi) : 0.0;
                    __syncthreads();
                    }
                    STAMP(4);                               // 4: d = N'w, r = P d
                    // z = w - Y'r, rollout, then iterative refinement against the true Gram matrix
                    double delta = 0.0, t1 = INFINITY;
                    int l = 0x7fffffff;
                    bool enter_at_correction = polish;                 // the projection starts at "dr = P rho" with rho = active slacks
                    for (int pass = 0; pass < 4; ++pass) {
                        if (!enter_at_correction) {
                        if (IDENT) {
                            n_combine(pass == 0 ? s_r : s_rho, pass == 0 ? nullptr : zb, pc);
                            __syncthreads();
                        } else if (H <= 32 && qhi <= QY) {
                            y_combine_roll(pass == 0 ? s_r : s_rho, pass == 0 ? wb : zb);
                            __syncthreads();
                        } else {
                        y_combine(pass == 0 ? s_r : s_rho, pass == 0 ? wb : zb);       // pass>0: correction dr held in s_rho
                        __syncthreads();
                        roll_lds<NJ>(zb, H, dt, tid);
                        __syncthreads();
                        }
                        if (polish) break;                             // zb = (x, Bvel x, Bpos x) after the projection
                        delta = ndot<NJ>(pc, zb, s_g, H);              // n_p'z
                        if (qhi == nfree) break;                       // empty active set: nothing to refine, t1 = inf
                        // one exchange carries the refinement diagnostics (r'rho, max|rho|, max|d|; rho_a = n_a'z is
                        // zero in exact arithmetic) and the dual step length t1 = min{lambda_a / r_a : r_a > 0}
                        double rr = 0.0, rmax = 0.0, dmax = 0.0, t1c = INFINITY;
                        if (tid < qhi && myact >= 0) {
                            const double ra = ndot<NJ>(myact, zb, s_g, H), rv = s_r[tid];
                            s_prow[tid] = ra;
                            rr = rv * ra;
                            rmax = fabs(ra);
                            dmax = fabs(s_d[tid]);
                            if (rv > 0.0) t1c = s_lam[tid] / rv;
                        } else if (tid < qhi) s_prow[tid] = 0.0;
                        rr = wave_add(rr);
                        rmax = -wave_min(-rmax);
                        dmax = -wave_min(-dmax);
                        const double t1w = wave_min(t1c);
                        const unsigned long long hit = __ballot(t1c == t1w);
                        const int lw = (tid & ~63) + (hit ? (int)__builtin_ctzll(hit) : 0);
                        double *rx = red;                              // one exchange buffer for this reduction (alternating)
                        if ((tid & 63) == 0) {
                            const int wv = tid >> 6;
                            rx[wv] = rr; rx[4 + wv] = rmax; rx[8 + wv] = dmax; rx[12 + wv] = t1w;
                            reinterpret_cast<int *>(rx + 16)[wv] = lw;
                        }
                        __syncthreads();
                        rr = (rx[0] + rx[1]) + (rx[2] + rx[3]);
                        rmax = fmax(fmax(rx[4], rx[5]), fmax(rx[6], rx[7]));
                        dmax = fmax(fmax(rx[8], rx[9]), fmax(rx[10], rx[11]));
                        t1 = rx[12]; l = reinterpret_cast<int *>(rx + 16)[0];
#pragma unroll
                        for (int w = 1; w < 4; ++w)
                            if (rx[12 + w] < t1) { t1 = rx[12 + w]; l = reinterpret_cast<int *>(rx + 16)[w]; }
                        const double ref = fmax(fabs(delta), DEP_TOL_F * spp);
                        if (pass == 3 || (P.opt & 2) || !(fabs(rr) > CFS_REF_A * ref || rmax > CFS_REF_B * (dmax + 1e-300))) break;
                        }
                        enter_at_correction = false;
                        if (tid < qhi) {                                // dr = P rho ; r += dr   (projection: lambda -= P s)
                            const double dr = myact >= 0 ? Pr.dot(s_prow, s_pt, tid, qhi) : 0.0;
                            s_rho[tid] = dr;
                            if (polish) s_lam[tid] -= dr; else s_r[tid] += dr;
                        }
                        __syncthreads();
                    }
                    if (polish) {
                        for (int k = tid; k < 3 * HN; k += FT) xs[k] = zb[k];
                        __syncthreads();
                        ++npolish;
                        break;                                         // back to step 1: scan again
                    }
                    STAMP(5);                               // 5: z, rollout, refinement, dual step length
                    const bool dependent = !(delta > DEP_TOL_F * spp);
                    const double t2 = dependent ? INFINITY : -sp / delta;
                    const double t = fmin(t1, t2);
                    if (P.dbg && b == P.dbg_b && tid == 0) {
                        const int n = (int)P.dbg[0];
                        if (n < P.dbg_cap) {
                            double *o = P.dbg + 8 + (size_t)n * 8;
                            o[0] = iter_O * 100000.0 + iters; o[1] = qhi - nfree; o[2] = pidx; o[3] = sp;
                            o[4] = delta / spp; o[5] = t1; o[6] = t2; o[7] = fgain / fbound;
                            P.dbg[0] = n + 1;
                        }
                    }
                    if (!(t < INFINITY)) { qp_status = QP_INFEASIBLE; break; }
                    const bool full = !dependent && t2 <= t1;
                    // The primal step is taken even when n_p is classified as dependent (t2 = inf, t = t1): z is then tiny but
                    // not zero, t1 = lambda_l / r_l can be 1e5..1e8, and x - x0 = H^{-1} N lambda holds only if x moves with
                    // exactly the r that moves the multipliers.  (Skipping it cost 1e-5 rad on ~1 % of the config-3 batch.)
                    for (int k = tid; k < 3 * HN; k += FT) xs[k] += t * zb[k];
                    fgain += t * delta * (0.5 * t + lam_p);              // Goldfarb-Idnani step (c)(ii)
                    if (tid < qhi && myact >= 0) s_lam[tid] -= t * s_r[tid];
                    lam_p += t;
                    STAMP(6);                               // 6: step length, trace, x/lambda update
                    if (full) {
                        // P <- [P + r r'/delta, -r/delta; -r'/delta, 1/delta] in a free slot: with r[slot] := -1
                        // the bordered update is one axpy per old row and one scaled copy for the new row
                        int slot;
                        if (nfree > 0) slot = s_free[nfree - 1];
                        else slot = qhi;
                        if (slot >= QB || qhi - nfree >= nn) { qp_status = QP_NUMERIC; break; }
                        const int qn = max(qhi, slot + 1);
                        const double inv = 1.0 / delta;
                        sync_rows(qn <= 64);
                        if (tid == 0) s_r[slot] = -1.0;
                        if (slot >= PR && slot == qhi && tid < QB) s_pt[(slot - PR) * QB + tid] = 0.0;   // fresh tail column: each row's own entry
                        sync_rows(qn <= 64);
                        if (tid < qhi && myact >= 0) Pr.axpy(s_r[tid] * inv, s_r, s_pt, tid, qn);
                        else if (tid == slot) Pr.set_scaled(-inv, s_r, s_pt, tid, qn);
                        if (!IDENT) {
                            if (slot < QY) { for (int k = tid; k < HN; k += FT) s_Y[slot * HN + k] = wb[k]; }
                            else { for (int k = tid; k < HN; k += FT) Yg[(size_t)slot * nn + k] = wb[k]; }
                        }
                        if (tid == 0) { s_act[slot] = pc; s_lam[slot] = lam_p; s_flag[pidx] = 1; s_slot[pidx] = (unsigned short)(slot + 1); }
                        if (nfree > 0) --nfree;
                        qhi = qn;
                        __syncthreads();
                        STAMP(7);                           // 7: add
                        break;                                          // back to step 1
                    }
                    // partial step: drop blocking constraint l: rank-1 downdate (which also annihilates
                    // column l up to rounding), zero row l, push the slot on the free stack
                    {
                        const int gone = s_act[l];
                        const bool one_wave = qhi <= 56;                     // the mask below covers whole 8-column chunks
                        sync_rows(one_wave);
                        if (tid == l) Pr.store(s_prow, s_pt, tid, qhi);
                        if (tid < QB && tid < qhi + 8) s_rho[tid] = tid == l ? 0.0 : 1.0;     // exact column mask
                        sync_rows(one_wave);
                        const double pll = s_prow[l];
                        if (tid < qhi && myact >= 0 && tid != l) Pr.axpy_mask(-(s_prow[tid] / pll), s_prow, s_rho, s_pt, tid, qhi);
                        if (tid == l) Pr.zero(s_pt, tid, qhi);
                        if (tid == 0) {
                            s_act[l] = -1;
                            s_lam[l] = 0.0;
                            s_free[nfree] = l;
                            const int gt = gone >> 16, gi = (gone >> 8) & 0xff, gj = gone & 0xff;
                            const int gidx = gt == CT_COL ? gj * H + gi : nobs * H + (gt - 1) * HN + gi * NJ + gj;
                            s_flag[gidx] = 0;
                            s_slot[gidx] = 0;
                        }
                        ++nfree;
                        __syncthreads();
                        STAMP(8);                           // 8: drop
                    }
                    { double bb; sp = slack_of<NJ>(pc, xs, s_g, s_rhs, s_lim, s_v0, s_mx, H, &bb); }
                }
                if (qp_status != QP_OK) break;
            }
        }
        total_iter += iters;
        if (!skip) prev_q = qp_status == QP_OK ? qhi : 0;
        if (P.dump_lambda) {
            const int nlam = nobs * H + 4 * HN;
            __syncthreads();
            for (int e = tid; e < nlam; e += FT) P.dump_lambda[(size_t)b * nlam + e] = 0.0;
            __syncthreads();
            if (tid < qhi && s_act[tid] >= 0 && qp_status == QP_OK) {
                const int ac = s_act[tid], at = ac >> 16, ai = (ac >> 8) & 0xff, aj = ac & 0xff;
                P.dump_lambda[(size_t)b * nlam + (at == CT_COL ? aj * H + ai : nobs * H + (at - 1) * HN + ai * NJ + aj)] = s_lam[tid];
            }
        }
        if (P.piece == 2) {
            for (int e = tid; e < HN; e += FT) P.u[(size_t)b * nn + e] = xs[e];
            if (tid == 0) { P.total_iter[b] = iters; P.status[b] = qp_status == QP_OK ? CFS_OK_CONVERGED : (qp_status == QP_INFEASIBLE ? CFS_QP_INFEASIBLE : CFS_NUMERIC); }
            drop_pool();
            return;
        }
        if (qp_status != QP_OK) {        // the reference would crash here (CFS_FANUC.m:92); report instead
            status = qp_status == QP_INFEASIBLE ? CFS_QP_INFEASIBLE : CFS_NUMERIC;
            done = true;
            break;
        }

        // =========================================================================================
        // new u, rollout (CFS_FANUC.m:86-95), get_cost, store_result, iter_O++, stop_outer
        // =========================================================================================
        // One phase, one barrier: the new state, the three sums (|du|^2, |dx|^2, cost) and the history.  xs = (u, Bvel u, Bpos u) is
        // final here (a skipped PSG step left the rollout of the unchanged u in it), so everything below reads xs; s_u is refreshed
        // at the end of the phase for the next iteration.
        double du2 = 0.0, dx2 = 0.0, cpart = 0.0;
        for (int k = tid; k < HN; k += FT) {
            const int i = k / NJ, c = k - i * NJ;
            const double un = xs[k], e = s_u[k] - un;
            du2 += e * e;
            const double th = (s_th0[c] + ((double)(i + 1) * dt) * s_v0[c]) + xs[2 * HN + k];
            const double om = s_v0[c] + xs[HN + k];
            const double oth = P.mode == CFS_MODE_PSGCFS ? 1.0 : s_x[i * NS + c];          // EVAL.m:47, SURVEY N1
            const double oom = P.mode == CFS_MODE_PSGCFS ? 1.0 : s_x[i * NS + NJ + c];
            dx2 += (th - oth) * (th - oth) + (om - oom) * (om - oom);
            s_x[i * NS + c] = th;
            s_x[i * NS + NJ + c] = om;
            s_u[k] = un;
            if (P.u_log) P.u_log[((size_t)b * P.max_o_iter + (iter_O - 1)) * nn + k] = un;   // test aid: the iterate of every outer iteration
            if (P.u_hist) P.u_hist[((size_t)b * P.max_o_iter + (iter_O - 1)) * nn + k] = un;
        }
        // cost = 0.5*u'*QQ*u + ff'*u + caug (EVAL.m:52).  CFS: the stop test does not depend on it, so u is
        // logged and the whole cost history is one batched MFMA product after the solve (cfs_gemm.hip).
        // PSGCFS: QQ*u is needed in the loop (stop_inner and the next gradient step).
        double cost = cost_new;
        if (!P.u_hist) {
            if (P.cost) {
                // QQ = Baug'*Qaug*Baug + cR*(R+R') (main_FANUC.m:96-97) applied through its factors: with (p, v) = (Bpos u, Bvel u)
                // already in xs, QQ*u = Bpos'(w_i (Qp p_i + qc v_i)) + Bvel'(w_i (qc p_i + Qv v_i)) + Rs u_i; Bpos', Bvel' are suffix
                // sums along the horizon (DPP scans over reversed lanes) -- no pass over the dense nn x nn matrix
                const DevCost *ck = reinterpret_cast<const DevCost *>(lds + L.cost);
                const bool two = H <= 32;
                const int lane = two ? (tid & 31) : (tid & 63), iR = H - 1 - lane;
                const int cstep = two ? 2 * (FT / 64) : FT / 64;
                for (int c = two ? (tid >> 5) : (tid >> 6); c < NJ; c += cstep) {
                    double Rp = 0.0, Rv = 0.0;
                    if (lane < H) {
                        const double wi = iR == H - 1 ? ck->wt : ck->ws;
                        double ap = 0.0, av = 0.0;
#pragma unroll
                        for (int c2 = 0; c2 < NJ; ++c2) {
                            ap += ck->Qp[c + c2 * NJ] * xs[2 * HN + iR * NJ + c2];
                            av += ck->Qv[c + c2 * NJ] * xs[HN + iR * NJ + c2];
                        }
                        Rp = wi * (ap + ck->qc * xs[HN + iR * NJ + c]);
                        Rv = dt * (wi * (ck->qc * xs[2 * HN + iR * NJ + c] + av));
                    }
                    const double S1 = two ? half_scan_incl(Rp) : wave_scan_incl(Rp);
                    const double cmb = (dt * dt) * S1 + Rv;
                    const double S2V = two ? half_scan_incl(cmb) : wave_scan_incl(cmb);
                    const double y_rev = S2V - (0.5 * dt * dt) * S1;
                    const int src = (lane < H ? H - 1 - lane : lane) + (two ? (tid & 32) : 0);
                    const double y = __shfl(y_rev, src, 64);
                    if (lane < H) {
                        const int k = lane * NJ + c;
                        double ru = 0.0;
#pragma unroll
                        for (int c2 = 0; c2 < NJ; ++c2) ru += ck->Rs[c + c2 * NJ] * xs[lane * NJ + c2];
                        const double sq = y + ru;
                        s_qu[k] = sq;
                        cpart += xs[k] * (0.5 * sq + P.ff[(size_t)b * nn + k]);
                    }
                }
            } else
            for (int k = tid; k < HN; k += FT) {
                double sa[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                int c = 0;
                for (; c + CFS_MV_BATCH <= HN; c += CFS_MV_BATCH) {   // independent L2 loads in flight per thread
                    double ld[CFS_MV_BATCH];
#pragma unroll
                    for (int j = 0; j < CFS_MV_BATCH; ++j) ld[j] = P.QQ[k + (size_t)(c + j) * nn];
#pragma unroll
                    for (int j = 0; j < CFS_MV_BATCH; ++j) sa[j & 7] += ld[j] * xs[c + j];
                }
                for (; c < HN; ++c) sa[0] += P.QQ[k + (size_t)c * nn] * xs[c];
                const double s = ((sa[0] + sa[1]) + (sa[2] + sa[3])) + ((sa[4] + sa[5]) + (sa[6] + sa[7]));
                s_qu[k] = s;
                cpart += xs[k] * (0.5 * s + P.ff[(size_t)b * nn + k]);
            }
        }
        block_sum3(du2, dx2, cpart, red, tid);
        if (P.u_hist) {
            if (tid == 0) P.e_u_all[(size_t)b * P.max_o_iter + (iter_O - 1)] = sqrt(du2);
        } else {
            cost = cpart + P.caug[b];
            if (tid == 0) {
                const size_t o = (size_t)b * P.max_o_iter + (iter_O - 1);
                P.cost_all[o] = cost;                              // EVAL.m:56-58
                P.e_cost_all[o] = fabs(cost_old - cost);
                P.e_u_all[o] = sqrt(du2);
            }
        }
        STAMP(9);                                           // 9: rollout, cost, history
        cost_new = cost;
        ++iter_O;                                              // CFS_FANUC.m:77
        if (sqrt(dx2) < P.epsilon_O) { done = true; status = CFS_OK_CONVERGED; }     // EVAL.m:64-68
        else if (iter_O > P.max_o_iter) { done = true; status = CFS_OK_MAXITER; }    // EVAL.m:69-72
        if (P.max_launch_iters > 0 && ++launched >= P.max_launch_iters) break;        // the host relaunches (mesh obstacles)
    }

    // ---- results -----------------------------------------------------------------------------------
    drop_pool();
    __syncthreads();
    for (int e = tid; e < HN; e += FT) P.u[(size_t)b * nn + e] = s_u[e];
    for (int e = tid; e < NX; e += FT) P.x_[(size_t)b * NX + e] = s_x[e];
    if (tid == 0) { P.iter_O[b] = iter_O; P.total_iter[b] = total_iter; P.status[b] = status; }
    if (P.st_qu) {                                         // state for the next launch of a host-driven solve
        for (int e = tid; e < HN; e += FT) P.st_qu[(size_t)b * nn + e] = s_qu[e];
        if (tid == 0) { P.st_cost[2 * b] = cost_new; P.st_cost[2 * b + 1] = cost_old; P.st_noise[b] = noise_row; P.st_done[b] = done ? 1 : 0; }
    }
    if (P.stamps && tid == 0) for (int k = 0; k < 12; ++k) P.stamps[(size_t)b * 12 + k] = s_acc[k];
}

#undef red

template <int NJ, int QB, bool IDENT>
hipError_t launch_fused_inst2(const FusedParams &p, size_t lds, hipStream_t s)
{
    // the function attribute is per device (handles may live on several GPUs of one process: cfs_set_device)
    static std::atomic<unsigned long long> attr_set{0ull};
    auto kern = cfs_solve_fused_kernel<NJ, QB, IDENT>;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 64 || !((attr_set.load(std::memory_order_acquire) >> dev) & 1ull)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev < 64) attr_set.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3(p.B), dim3(FT), lds, s, p);
    return hipGetLastError();
}

// CFS_IDENT_SET: which Hessians this tier is compiled for (0: QQ only, 1: identity only, 2: both) -- compile time, not a dual path
#ifndef CFS_IDENT_SET
#define CFS_IDENT_SET 2
#endif
template <int NJ, int QB>
hipError_t launch_fused_inst(const FusedParams &p, size_t lds, hipStream_t s)
{
    const bool ident = p.mode == CFS_MODE_PSGCFS;
#if CFS_IDENT_SET != 0
    if (ident) return launch_fused_inst2<NJ, QB, true>(p, lds, s);
#endif
#if CFS_IDENT_SET != 1
    if (!ident) return launch_fused_inst2<NJ, QB, false>(p, lds, s);
#endif
    return hipErrorInvalidValue;
}

}  // namespace

// linearisation scratch per waypoint of a tile: sin/cos, link end points, base distances, minima per evaluation point, candidate lists (ushort)
static size_t lin_doubles_per_wp(int nj, int nobs)
{
    return (size_t)nj * 6 + (size_t)nvt(nj) * 6 + (size_t)nj * nobs + (size_t)nobs * (2 * nj + 1) + ((size_t)nj * nobs + 3) / 4;
}

// does the fused kernel's fixed LDS footprint (+ a minimal Y / linearisation region) fit a CU?
bool CFS_CAT(fused_fits, CFS_VARIANT)(int nj, int H, int nobs)
{
    const int nn = H * nj;
    const int QB = nn <= 96 ? 96 : (nn <= 160 ? 160 : 256);
    const FusedLayout L = fused_layout(nj, H, nobs, QB, QB < CFS_PR ? QB : CFS_PR);
    const size_t avail = (160 * 1024 / CFS_WG_PER_CU) / 8 - 64;
    const size_t per_wp = lin_doubles_per_wp(nj, nobs);
    return (size_t)L.total_fixed + (size_t)4 * nn <= avail && (size_t)L.lin + per_wp <= avail;
}

// host: choose the capacities, fill qy / lin_w, launch
hipError_t CFS_CAT(launch_fused, CFS_VARIANT)(int nj, FusedParams p, hipStream_t s)
{
    const int nn = p.H * nj;
    const int QB = nn <= 96 ? 96 : (nn <= 160 ? 160 : 256);
    const FusedLayout L = fused_layout(nj, p.H, p.nobs, QB, QB < CFS_PR ? QB : CFS_PR);
    const size_t avail = (160 * 1024 / CFS_WG_PER_CU) / 8 - 64;   // doubles per workgroup, small safety margin
    const bool ident = p.mode == CFS_MODE_PSGCFS;          // H = I: no Y rows at all (closed-form normals)
    if ((size_t)L.total_fixed + (ident ? 0 : 4 * nn) > avail) return hipErrorInvalidValue;
    const size_t region = avail - L.total_fixed;
    int qy = ident ? 0 : (int)(region / nn);
    if (qy > nn) qy = nn;
    const size_t per_wp = lin_doubles_per_wp(nj, p.nobs);
    int w = (int)((avail - L.lin) / per_wp);   // the linearisation may use the QP's work vectors too (layout)
    if (w < 1) return hipErrorInvalidValue;
    w = (p.H + (p.H + w - 1) / w - 1) / ((p.H + w - 1) / w);   // equal tiles: ceil(H / number of tiles)
    p.qy = qy;
    p.lin_w = w;
    const size_t need = std::max((size_t)L.total_fixed + (size_t)qy * nn, (size_t)L.lin + (size_t)w * per_wp);
    const size_t lds = need * 8;
    switch (nj * 1000 + QB) {
    case 2096: return launch_fused_inst<2, 96>(p, lds, s);
    case 3096: return launch_fused_inst<3, 96>(p, lds, s);
    case 4096: return launch_fused_inst<4, 96>(p, lds, s);
    case 5096: return launch_fused_inst<5, 96>(p, lds, s);
    case 6096: return launch_fused_inst<6, 96>(p, lds, s);
    case 2160: return launch_fused_inst<2, 160>(p, lds, s);
    case 4160: return launch_fused_inst<4, 160>(p, lds, s);
    case 5160: return launch_fused_inst<5, 160>(p, lds, s);
    case 6160: return launch_fused_inst<6, 160>(p, lds, s);
    case 3160: return launch_fused_inst<3, 160>(p, lds, s);
    case 3256: return launch_fused_inst<3, 256>(p, lds, s);
    case 4256: return launch_fused_inst<4, 256>(p, lds, s);
    case 5256: return launch_fused_inst<5, 256>(p, lds, s);
    case 6256: return launch_fused_inst<6, 256>(p, lds, s);
    default: return hipErrorInvalidValue;
    }
}
