// cfs_gemm.hip -- K3: the dense contractions of the path, on the fp64 matrix cores, and the
// per-iteration bookkeeping of EVAL (rows a8, a9 in DESIGN.md).
//
// The only dense products left on the path once the constraint rows are kept in structured form
// are "one nn x nn matrix times a batch of B vectors":
//   -H^{-1} * ff_b            unconstrained minimiser of the CFS QP (once per solve)
//   QQ * u_b                  get_cost (Lib/EVAL.m:51-53) and dcostArm_f (Lib/PSGCFS_FANUC.m:131-133)
// i.e. an (nn x nn) x (nn x B) GEMM; it runs on v_mfma_f64_16x16x4_f64.  Everything else on the
// path is not GEMM-shaped and stays on the vector ALU.
#include "cfs_device.h"

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int GEMV_RT = 2;       // 16-row output tiles per wavefront

// Y[b][r] = scale * sum_c M[r + c*nn] * X[b][c].  One wavefront: 16 problems x GEMV_RT*16 rows.
// MFMA operand maps (f64 16x16x4): lane l gives A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// D[row = (l>>4) + 4*reg][col = l&15].  Here D rows = output index r, D columns = problem b.
__global__ __launch_bounds__(CFS_WAVE) void cfs_batched_gemv_kernel(GemvParams P)
{
    const int lane = threadIdx.x, nn = P.nn;
    const int b0 = blockIdx.x * 16, r0 = blockIdx.y * (16 * GEMV_RT);
    const int lj = lane & 15, lk = lane >> 4;
    const int bj = b0 + lj;
    const bool bok = bj < P.B;
    const double *xrow = P.X + (size_t)(bok ? bj : 0) * nn;
    v4f64 acc[GEMV_RT];
#pragma unroll
    for (int t = 0; t < GEMV_RT; ++t) acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < nn; c0 += 4) {
        const int c = c0 + lk;
        const double bfrag = (bok && c < nn) ? xrow[c] : 0.0;
#pragma unroll
        for (int t = 0; t < GEMV_RT; ++t) {
            const int r = r0 + t * 16 + lj;
            const double afrag = (r < nn && c < nn) ? P.M[r + (size_t)c * nn] : 0.0;
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag, bfrag, acc[t], 0, 0, 0);
        }
    }
    if (!bok) return;
#pragma unroll
    for (int t = 0; t < GEMV_RT; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = r0 + t * 16 + lk + 4 * reg;
            if (r < nn) P.Y[(size_t)bj * nn + r] = P.scale * acc[t][reg];
        }
}

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, CFS_WAVE);
    return v;
}

// cost history of a CFS solve from the logged u and QQ*u (EVAL.m:51-59; CFS_FANUC.m:67,73-75): one
// wavefront per problem; cost_old of iteration k is the cost of iteration k-1, caug (= get_cost(0)) for k = 0.
__global__ __launch_bounds__(CFS_WAVE) void cfs_cost_history_kernel(CostHistParams P)
{
    const int b = blockIdx.x, lane = threadIdx.x, nn = P.nn, n_it = P.iter_O ? P.iter_O[b] - 1 : P.max_o_iter;   // iter_O == NULL: every logged row (cfs_cost_b)
    double prev = P.caug[b];
    for (int k = 0; k < n_it; ++k) {
        const double *u = P.u_hist + ((size_t)b * P.max_o_iter + k) * nn;
        const double *q = P.qu_hist + ((size_t)b * P.max_o_iter + k) * nn;
        double quad = 0.0, lin = 0.0;
        for (int e = lane; e < nn; e += CFS_WAVE) { quad += u[e] * q[e]; lin += P.ff[(size_t)b * nn + e] * u[e]; }
        quad = wsum(quad);
        lin = wsum(lin);
        const double cost = 0.5 * quad + lin + P.caug[b];
        if (lane == 0) {
            P.cost_all[(size_t)b * P.max_o_iter + k] = cost;
            if (P.e_cost_all) P.e_cost_all[(size_t)b * P.max_o_iter + k] = fabs(prev - cost);
        }
        prev = cost;
    }
}

// line reference + cost terms of B (start, goal) pairs (main_FANUC.m:38-49, :98-103); one wavefront per problem
__global__ __launch_bounds__(CFS_WAVE) void cfs_build_terms_kernel(TermsParams P)
{
    const int b = blockIdx.x, lane = threadIdx.x, nj = P.nj, H = P.H, nn = H * nj, ns = 2 * nj;
    const int nwp = P.nwp_b ? max(P.nwp_b[b], 1) : P.nwp;        // routes of different lengths: B x nwp_stride x nj, nwp_b[b] rows used
    const double *rt = P.route ? P.route + (size_t)b * (P.nwp_b ? P.nwp_stride : P.nwp) * nj : nullptr;
    const double *x0 = rt ? rt : P.x0 + (size_t)b * nj;
    const double *xg = rt ? rt + (size_t)(nwp - 1) * nj : P.xg + (size_t)b * nj;
    for (int k = lane; k < nn; k += CFS_WAVE) {
        double s = 0.0;
        for (int c = 0; c < nj; ++c) s += P.F1[k + (size_t)c * nn] * x0[c] - P.F2[k + (size_t)c * nn] * xg[c];
        P.ff[(size_t)b * nn + k] = s;
    }
    for (int e = lane; e < H * ns; e += CFS_WAVE) {            // linspace(x0, xg, H+1), waypoint 0 dropped, zero velocities
        const int i = e / ns, c = e - i * ns;
        double v = 0.0;
        if (c < nj && !rt) v = (i == H - 1) ? xg[c] : x0[c] + (double)(i + 1) * ((xg[c] - x0[c]) / (double)H);
        if (c < nj && rt && nwp < 2) v = rt[c];                  // a one-node route (start inside the goal region): stay there
        if (c < nj && rt && nwp >= 2) {
            // cubicpolytraj(route, (0:nwp-1)*dt, linspace(0, (nwp-1)*dt, H+1)) with zero waypoint velocities
            // (RRTstar_CFS.m:94-100): sample n = i+1 lies in segment k, tau in [0,1], value r_k + (3 tau^2 - 2 tau^3)(r_k+1 - r_k)
            const double T = (double)(nwp - 1) * P.dt;
            const double t = (i == H - 1) ? T : (double)(i + 1) * (T / (double)H);
            int k = (int)floor(t / P.dt);
            while ((double)(k + 1) * P.dt <= t) ++k;
            while (k > 0 && (double)k * P.dt > t) --k;
            k = min(max(k, 0), nwp - 2);
            const double t0 = (double)k * P.dt, t1 = (double)(k + 1) * P.dt;
            const double tau = (t - t0) / (t1 - t0);
            const double r0 = rt[(size_t)k * nj + c], r1 = rt[(size_t)(k + 1) * nj + c];
            v = r0 + (3 * tau * tau - 2 * tau * tau * tau) * (r1 - r0);
        }
        P.x_init[(size_t)b * H * ns + e] = v;
    }
    if (lane < ns) P.xR1[(size_t)b * ns + lane] = lane < nj ? x0[lane] : 0.0;
    if (lane == 0) {
        double q = 0.0;
        for (int r = 0; r < 2 * nj; ++r)
            for (int c = 0; c < 2 * nj; ++c)
                q += (r < nj ? x0[r] : xg[r - nj]) * P.Cq[r + (size_t)c * 2 * nj] * (c < nj ? x0[c] : xg[c - nj]);
        P.caug[b] = q;
    }
}

}  // namespace

void launch_batched_gemv(const GemvParams &p, hipStream_t s)
{
    const dim3 grid((p.B + 15) / 16, (p.nn + 16 * GEMV_RT - 1) / (16 * GEMV_RT)), block(CFS_WAVE);
    hipLaunchKernelGGL(cfs_batched_gemv_kernel, grid, block, 0, s, p);
}

void launch_cost_history(const CostHistParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(cfs_cost_history_kernel, dim3(p.B), dim3(CFS_WAVE), 0, s, p);
}

void launch_build_terms(const TermsParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(cfs_build_terms_kernel, dim3(p.B), dim3(CFS_WAVE), 0, s, p);
}
