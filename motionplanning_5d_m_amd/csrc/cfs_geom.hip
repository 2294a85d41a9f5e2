// cfs_geom.hip -- plain dist_arm (cfs_dist_arm, RRT feasibility) and the dense Ainq / binq writer.  The linearisation
// itself (distance + literal central-difference Jacobian) lives in cfs_fused.hip, where the solver runs it.
//
// Reference behaviour restated (not translated):
//   Lib/functions/CapPos.m:8-22, Lib/2L/CapPos2.m:1-31        forward kinematics
//   Lib/functions/distLinSeg.m:23-101                         Lumelsky segment-segment distance
//   Lib/200i/dist_arm_3D_200i_2.m:1-30 (+ M16iB, 2L variants) min over links, near-zero surrogate
//   Lib/functions/num_jac.m:1-17                              central difference, xp never restored
//
// MI355X mapping: one 256-thread workgroup handles LIN_W waypoints of one problem.  The 2*NJ+1
// evaluation points of num_jac differ from the base pose in a prefix of the joints only, so link
// k has just 2k+1 distinct world transforms ("variants"); the kernel evaluates those NJ(NJ+2)
// link variants once (35 instead of 55 for NJ=5), stages their end points in LDS, runs every
// (variant x obstacle) segment pair on its own lane, and recombines the 2*NJ+1 minima per
// (waypoint, obstacle).  Robot constants are staged in LDS because the link index is per-lane.
#include "cfs_geom_dev.h"

namespace {

// plain dist_arm for N configurations x nobs obstacles (API entry cfs_dist_arm; RRT feasibility)
__global__ __launch_bounds__(256) void cfs_dist_arm_kernel(DistArmParams P)
{
    __shared__ __attribute__((aligned(16))) double s_rb[sizeof(DevRobot) / 8];
    {
        const double *src = reinterpret_cast<const double *>(P.rb);
        for (int e = threadIdx.x; e < (int)(sizeof(DevRobot) / 8); e += blockDim.x) s_rb[e] = src[e];
    }
    __syncthreads();
    const DevRobot *rb = reinterpret_cast<const DevRobot *>(s_rb);
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= P.N) return;
    double ends[CFS_MAX_LINKS * 6];
    double M[12], Mn[12];
    for (int k = 0; k < P.nj; ++k) {
        double sn, cs;
        sincos(P.theta[(size_t)n * P.nj + k] - rb->th_off[k], &sn, &cs);
        fk_step(rb, k, sn, cs, k == 0 ? nullptr : M, Mn);
        for (int q = 0; q < 12; ++q) M[q] = Mn[q];
        link_ends(rb, k, M, ends + k * 6);
        if (P.pos) for (int q = 0; q < 6; ++q) P.pos[((size_t)n * P.nj + k) * 6 + q] = ends[k * 6 + q];
    }
    for (int j = 0; j < P.nobs; ++j) {
        double o6[6];
        for (int q = 0; q < 6; ++q) o6[q] = P.obs[j * 6 + q];
        double d = INFINITY;
        int lid = 0;
        for (int k = 0; k < P.nj; ++k) {
            const double dis = seg_seg_dist(ends + k * 6, o6);
            if (dis < d) { d = dis; lid = k + 1; }
        }
        P.d[(size_t)n * P.nobs + j] = d;
        if (P.linkid) P.linkid[(size_t)n * P.nobs + j] = lid;
    }
}

// key[b] = number of (waypoint, line obstacle) pairs of problem b whose clearance on the initial trajectory is below the
// margin.  Eight problems per 256-thread workgroup, 32 lanes per problem, one waypoint per lane: the pre-pass runs on the
// solve's stream while ANOTHER solve's fused kernel holds every compute unit (two workgroups of 256 VGPRs per lane fill
// the register file), so each of its workgroups waits for a fused workgroup to retire -- 128 workgroups instead of 1 024 per
// batch of 1 024: config 3 PSGCFS 1.467 -> 1.451 ms per solve, CFS 1.598 -> 1.586 (same-call A/B, two rounds; 16 grid-striding
// workgroups instead: PSGCFS 1.489, CFS 1.578 -- the longer pre-pass costs the short solve more than the waiting saved).
// __launch_bounds__(256, 4): 128 VGPRs (18 spilled) so that two of these workgroups fit the registers one retired fused workgroup
// leaves: PSGCFS 1.4415 -> 1.4355 ms per solve.
constexpr int ORDER_PB = 8;
__global__ __launch_bounds__(256, 4) void cfs_order_key_kernel(OrderParams P)
{
    __shared__ __attribute__((aligned(16))) double s_rb[sizeof(DevRobot) / 8];
    __shared__ int s_cnt[ORDER_PB];
    {
        const double *src = reinterpret_cast<const double *>(P.rb);
        for (int e = threadIdx.x; e < (int)(sizeof(DevRobot) / 8); e += blockDim.x) s_rb[e] = src[e];
    }
    if (threadIdx.x < ORDER_PB) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const DevRobot *rb = reinterpret_cast<const DevRobot *>(s_rb);
    const int sub = threadIdx.x >> 5, lane = threadIdx.x & 31;
    const int b = blockIdx.x * ORDER_PB + sub, ns = 2 * P.nj;
    int cnt = 0;
    if (b < P.B)
    for (int i = lane; i < P.H; i += 32) {
        double ends[CFS_MAX_LINKS * 6];
        double M[12], Mn[12];
        for (int k = 0; k < P.nj; ++k) {
            double sn, cs;
            sincos(P.x_init[((size_t)b * P.H + i) * ns + k] - rb->th_off[k], &sn, &cs);
            fk_step(rb, k, sn, cs, k == 0 ? nullptr : M, Mn);
            for (int q = 0; q < 12; ++q) M[q] = Mn[q];
            link_ends(rb, k, M, ends + k * 6);
        }
        for (int j = 0; j < P.nobs; ++j) {
            double o6[6];
            for (int q = 0; q < 6; ++q) o6[q] = P.obs[((size_t)b * P.obs_stride + j) * 6 + q];
            double d = INFINITY;
            for (int k = 0; k < P.nj; ++k) d = fmin(d, seg_seg_dist(ends + k * 6, o6));
            cnt += d < P.margin[j];
        }
    }
    if (cnt) atomicAdd(&s_cnt[sub], cnt);
    __syncthreads();
    if (lane == 0 && b < P.B) P.key[b] = s_cnt[sub];
}

// order = problems by descending key, ties by index (a stable rank: no atomics, the same order on every run)
__global__ __launch_bounds__(256) void cfs_order_rank_kernel(OrderParams P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int ki = i < P.B ? P.key[i] : 0;
    int rank = 0;
    for (int j = 0; j < P.B; ++j) {          // j is uniform: scalar loads
        const int kj = P.key[j];
        rank += (kj > ki) || (kj == ki && j < i);
    }
    if (i < P.B) P.order[rank] = i;
}

// dense self.Ainq / self.binq in the reference's row order (CFS_FANUC.m:119-129), HBM-bound writer:
// one workgroup per (problem, column), threads along the contiguous row index.
__global__ __launch_bounds__(256) void cfs_dense_con_kernel(DenseConParams P)
{
    const int nj = P.nj, H = P.H, nn = H * nj, ns = 2 * nj;
    const int per = 1 + 2 * nj, rows = P.nobs * H * per;
    const int b = blockIdx.y, col = blockIdx.x;           // col = k*nj + cc
    const int k = col / nj, cc = col % nj;
    const double dt = P.dt;
    double *A = P.Ainq + ((size_t)b * rows * nn) + (size_t)col * rows;
    const double *g = P.grad + (size_t)b * P.nobs * H * nj;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const int blk = r / per, t = r % per, j = blk / H, i = blk % H;
        double v = 0.0;
        if (k <= i) {
            if (t == 0) v = -(g[((size_t)j * H + i) * nj + cc] * ((double)(i - k) + 0.5) * dt * dt);
            else if (t <= nj) v = (t - 1 == cc) ? dt : 0.0;
            else v = (t - 1 - nj == cc) ? -dt : 0.0;
        }
        if (v == 0.0) v = 0.0;                             // no negative zeros in the dense output
        A[r] = v;
    }
    if (col == 0) {
        const double *u = P.u + (size_t)b * nn;
        const double *x1 = P.xR1 + (size_t)b * ns;
        for (int r = threadIdx.x; r < rows; r += blockDim.x) {
            const int blk = r / per, t = r % per, j = blk / H, i = blk % H;
            double s;
            if (t == 0) {
                // s = (d - margin) - Diff'*Bj(1:nj,:)*u     (CFS_FANUC.m:119-120)
                double gbu = 0.0;
                for (int c = 0; c < nj; ++c) {
                    double pu = 0.0;
                    for (int kk = 0; kk <= i; ++kk) pu += ((double)(i - kk) + 0.5) * dt * dt * u[kk * nj + c];
                    gbu += g[((size_t)j * H + i) * nj + c] * pu;
                }
                s = (P.dist[((size_t)b * P.nobs + j) * H + i] - P.margin[j]) - gbu;
            } else {
                const int c = (t - 1) % nj;
                const double v0 = x1[nj + c];              // Aaug(vel rows)*xR(:,1) = initial velocity
                s = (t <= nj) ? P.lim[c] - v0 : P.lim[c] + v0;
            }
            P.binq[(size_t)b * rows + r] = s;
        }
    }
}

}  // namespace

void launch_dist_arm(const DistArmParams &p, hipStream_t s)
{
    const dim3 grid((p.N + 255) / 256), block(256);
    hipLaunchKernelGGL(cfs_dist_arm_kernel, grid, block, 0, s, p);
}

void launch_order(const OrderParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(cfs_order_key_kernel, dim3((p.B + ORDER_PB - 1) / ORDER_PB), dim3(256), 0, s, p);
    hipLaunchKernelGGL(cfs_order_rank_kernel, dim3((p.B + 255) / 256), dim3(256), 0, s, p);
}

void launch_dense_con(const DenseConParams &p, hipStream_t s)
{
    const dim3 grid(p.H * p.nj, p.B), block(256);
    hipLaunchKernelGGL(cfs_dense_con_kernel, grid, block, 0, s, p);
}
