// cfs_geom.hip -- K1: forward kinematics of the capsule axes, segment-segment distance and the
// literal central-difference Jacobian (rows a1-a4 and the distance half of a5 in DESIGN.md).
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

constexpr int LIN_W = 5;         // waypoints per workgroup
constexpr int LIN_THREADS = 256;

// ------------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(LIN_THREADS) void cfs_linearize_kernel(LinParams P)
{
    constexpr int NS = 2 * NJ, NVT = nvt(NJ), NE = 2 * NJ + 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tiles = (P.H + LIN_W - 1) / LIN_W;
    const int b = blockIdx.x / tiles;
    const int w0 = (blockIdx.x % tiles) * LIN_W;
    if (P.done && P.done[b]) return;
    const int W = min(LIN_W, P.H - w0);
    const int tid = threadIdx.x;
    const int nobs = P.nobs;

    // LDS carve-up (all doubles)
    DevRobot *rb = reinterpret_cast<DevRobot *>(lds);
    double *s_th = lds + sizeof(DevRobot) / 8;            // [LIN_W][NJ]
    double *s_sc = s_th + LIN_W * NJ;                     // [LIN_W][NJ][3][2]  (sin, cos) of base,+h,-h
    double *s_tm = s_sc + LIN_W * NJ * 6;                 // [LIN_W][NVT][12]
    double *s_en = s_tm + LIN_W * NVT * 12;               // [LIN_W][NVT][6]
    double *s_ob = s_en + LIN_W * NVT * 6;                // [nobs][6]
    double *s_dt = s_ob + nobs * 6;                       // [LIN_W][NVT][nobs]

    // stage robot constants, obstacle axes and the stacked trajectory block (coalesced reads)
    {
        const double *src = reinterpret_cast<const double *>(P.rb);
        for (int e = tid; e < (int)(sizeof(DevRobot) / 8); e += LIN_THREADS) lds[e] = src[e];
        const double *ob = P.obs + (size_t)b * nobs * 6;
        for (int e = tid; e < nobs * 6; e += LIN_THREADS) s_ob[e] = ob[e];
        const double *xb = P.x_ + (size_t)b * P.H * NS + (size_t)w0 * NS;
        for (int e = tid; e < W * NS; e += LIN_THREADS) {
            const int wi = e / NS, c = e % NS;
            if (c < NJ) s_th[wi * NJ + c] = xb[e];
        }
    }
    __syncthreads();

    // P1: sin/cos of (theta, theta+eps/2, theta-eps/2) minus the model's joint offset
    for (int e = tid; e < W * NJ * 3; e += LIN_THREADS) {
        const int var = e % 3, m = (e / 3) % NJ, wi = e / (3 * NJ);
        double x = s_th[wi * NJ + m];
        if (var == 1) x = x + FD_EPS / 2;                 // num_jac.m:11
        else if (var == 2) x = x - FD_EPS / 2;            // num_jac.m:13
        x = x - rb->th_off[m];                            // dist_arm_3D_200i_2.m:11
        double sn, cs;
        sincos(x, &sn, &cs);
        s_sc[((wi * NJ + m) * 3 + var) * 2 + 0] = sn;
        s_sc[((wi * NJ + m) * 3 + var) * 2 + 1] = cs;
    }
    __syncthreads();

    // P2: link transforms level by level; link k (1-based) has variants v = 0..2k:
    //   v <= 2(k-1): joints k..: base angle, parent variant v
    //   v == 2k-1 : joint k at +h, joints <k at -h  (parent variant 2(k-1))
    //   v == 2k   : joint k at -h, joints <k at -h  (parent variant 2(k-1))
    for (int k1 = 1; k1 <= NJ; ++k1) {
        const int nv = 2 * k1 + 1;
        for (int e = tid; e < W * nv; e += LIN_THREADS) {
            const int v = e % nv, wi = e / nv;
            const int avar = (v == 2 * k1 - 1) ? 1 : (v == 2 * k1 ? 2 : 0);
            const int pv = min(v, 2 * (k1 - 1));
            const double sn = s_sc[((wi * NJ + (k1 - 1)) * 3 + avar) * 2 + 0];
            const double cs = s_sc[((wi * NJ + (k1 - 1)) * 3 + avar) * 2 + 1];
            const double *par = (k1 == 1) ? nullptr : s_tm + (wi * NVT + kvoff(k1 - 1) + pv) * 12;
            double M[12], e6[6];
            fk_step(rb, k1 - 1, sn, cs, par, M);
            link_ends(rb, k1 - 1, M, e6);
            double *dstM = s_tm + (wi * NVT + kvoff(k1) + v) * 12;
            double *dstE = s_en + (wi * NVT + kvoff(k1) + v) * 6;
#pragma unroll
            for (int q = 0; q < 12; ++q) dstM[q] = M[q];
#pragma unroll
            for (int q = 0; q < 6; ++q) dstE[q] = e6[q];
        }
        __syncthreads();
    }

    // P3: every (link variant, waypoint, obstacle) segment pair on its own lane; kv is the slow
    // index so that a wavefront mostly shares the link (uniform point/segment branch)
    for (int e = tid; e < NVT * W * nobs; e += LIN_THREADS) {
        const int j = e % nobs, wi = (e / nobs) % W, kv = e / (nobs * W);
        s_dt[(wi * NVT + kv) * nobs + j] = seg_seg_dist(s_en + (wi * NVT + kv) * 6, s_ob + j * 6);
    }
    __syncthreads();

    // P4: the 2NJ+1 evaluations of dist_arm per (waypoint, obstacle) and the literal num_jac
    for (int e = tid; e < W * nobs; e += LIN_THREADS) {
        const int j = e % nobs, wi = e / nobs;
        const double *tab = s_dt + (wi * NVT) * nobs + j;
        double dev[NE];
        int lid = 0;
#pragma unroll
        for (int ev = 0; ev < NE; ++ev) {
            double d = INFINITY;
#pragma unroll
            for (int k1 = 1; k1 <= NJ; ++k1) {
                const int v = min(ev, 2 * k1);
                const double dis = tab[(kvoff(k1) + v) * nobs];
                if (dis < d) { d = dis; if (ev == 0) lid = k1; }   // first minimum wins (:25-28)
            }
            dev[ev] = d;
        }
        const size_t o = ((size_t)b * nobs + j) * P.H + (w0 + wi);
        P.dist[o] = dev[0];
        if (P.linkid) P.linkid[o] = lid;
#pragma unroll
        for (int m = 0; m < NJ; ++m) P.grad[o * NJ + m] = (dev[2 * m + 1] - dev[2 * m + 2]) / FD_EPS;  // num_jac.m:15
    }
}

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

size_t linearize_lds_bytes(int nj, int nobs)
{
    const int NVT = nvt(nj);
    size_t dbl = sizeof(DevRobot) / 8 + LIN_W * nj + LIN_W * nj * 6 + (size_t)LIN_W * NVT * 12 + (size_t)LIN_W * NVT * 6 +
                 (size_t)nobs * 6 + (size_t)LIN_W * NVT * nobs;
    return dbl * 8;
}

void launch_linearize(int nj, const LinParams &p, hipStream_t s)
{
    const int tiles = (p.H + LIN_W - 1) / LIN_W;
    const dim3 grid(p.B * tiles), block(LIN_THREADS);
    const size_t lds = linearize_lds_bytes(nj, p.nobs);
    switch (nj) {
    case 2: hipLaunchKernelGGL(cfs_linearize_kernel<2>, grid, block, lds, s, p); break;
    case 3: hipLaunchKernelGGL(cfs_linearize_kernel<3>, grid, block, lds, s, p); break;
    case 4: hipLaunchKernelGGL(cfs_linearize_kernel<4>, grid, block, lds, s, p); break;
    case 5: hipLaunchKernelGGL(cfs_linearize_kernel<5>, grid, block, lds, s, p); break;
    case 6: hipLaunchKernelGGL(cfs_linearize_kernel<6>, grid, block, lds, s, p); break;
    default: break;   // validated by the caller
    }
}

void launch_dist_arm(const DistArmParams &p, hipStream_t s)
{
    const dim3 grid((p.N + 255) / 256), block(256);
    hipLaunchKernelGGL(cfs_dist_arm_kernel, grid, block, 0, s, p);
}

void launch_dense_con(const DenseConParams &p, hipStream_t s)
{
    const dim3 grid(p.H * p.nj, p.B), block(256);
    hipLaunchKernelGGL(cfs_dense_con_kernel, grid, block, 0, s, p);
}
