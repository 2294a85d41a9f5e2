// cfs_geom_dev.h -- device functions shared by the linearisation kernels: forward kinematics of one
// link (Lib/functions/CapPos.m:13-20, Lib/2L/CapPos2.m:19-28), Lumelsky segment-segment distance
// (Lib/functions/distLinSeg.m:23-91) with the near-zero surrogate of dist_arm_3D_200i_2.m:22-24.
#pragma once
#include "cfs_device.h"

constexpr double FD_EPS = 1e-5;  // num_jac.m:6

__host__ __device__ constexpr int nvt(int nj) { return nj * (nj + 2); }   // sum_{k=1..nj} (2k+1)
__device__ __forceinline__ int kvoff(int k1) { return k1 * k1 - 1; }      // offset of link k1 (1-based)

// one homogeneous link transform A_k(angle) appended to parent (3x4 row-major), CapPos.m:13-17
__device__ __forceinline__ void fk_step(const DevRobot *rb, int k, double st, double ct,
                                        const double *par, double *out)
{
    double R[12];
    if (rb->kind == CFS_ROBOT_2L) {           // CapPos2.m:19-25
        R[0] = ct;  R[1] = -st; R[2] = 0.0;  R[3] = rb->t2l[k * 3 + 0];
        R[4] = st;  R[5] = ct;  R[6] = 0.0;  R[7] = rb->t2l[k * 3 + 1];
        R[8] = 0.0; R[9] = 0.0; R[10] = 1.0; R[11] = rb->t2l[k * 3 + 2];
    } else {
        const double ca = rb->ca[k], sa = rb->sa[k], a = rb->dh_a[k], d = rb->dh_d[k];
        R[0] = ct;  R[1] = -st * ca; R[2] = st * sa;   R[3] = a * ct;
        R[4] = st;  R[5] = ct * ca;  R[6] = -ct * sa;  R[7] = a * st;
        R[8] = 0.0; R[9] = sa;       R[10] = ca;       R[11] = d;
    }
    if (par == nullptr) {                      // M{1} = eye(4)
#pragma unroll
        for (int e = 0; e < 12; ++e) out[e] = R[e];
        return;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double p0 = par[r * 4 + 0], p1 = par[r * 4 + 1], p2 = par[r * 4 + 2], p3 = par[r * 4 + 3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double s = p0 * R[c] + p1 * R[4 + c] + p2 * R[8 + c];
            if (c == 3) s += p3;
            out[r * 4 + c] = s;
        }
    }
}

// capsule axis end points in the world frame, CapPos.m:18-20
__device__ __forceinline__ void link_ends(const DevRobot *rb, int k, const double *M, double *e6)
{
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const double *p = rb->cap + k * 6 + kk * 3;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            e6[kk * 3 + r] = (M[r * 4 + 0] * p[0] + M[r * 4 + 1] * p[1] + M[r * 4 + 2] * p[2]) + M[r * 4 + 3] + rb->base[r];
    }
}

// a/b without the IEEE special-case scaffolding (v_div_scale/fmas/fixup): v_rcp_f64 seed, two Newton
// steps, one residual correction.  Result within 1 ulp of the correctly rounded quotient for the finite,
// well-scaled operands of this routine (lengths and dot products of link / obstacle axes in metres).
#ifndef CFS_FAST_DIV
#define CFS_FAST_DIV 1
#endif
__device__ __forceinline__ double fdiv(double a, double b)
{
#if !CFS_FAST_DIV
    return a / b;
#endif
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

__device__ __forceinline__ double fixbound(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// distLinSeg.m:23-91 followed by the near-zero surrogate of dist_arm_3D_200i_2.m:22-24.
// a6 = link axis [p1s;p1e], o6 = obstacle axis [p2s;p2e].
__device__ __forceinline__ double seg_seg_dist(const double *a6, const double *o6)
{
    const double d1x = a6[3] - a6[0], d1y = a6[4] - a6[1], d1z = a6[5] - a6[2];
    const double d2x = o6[3] - o6[0], d2y = o6[4] - o6[1], d2z = o6[5] - o6[2];
    const double d12x = o6[0] - a6[0], d12y = o6[1] - a6[1], d12z = o6[2] - a6[2];
    const double D1 = d1x * d1x + d1y * d1y + d1z * d1z;
    const double D2 = d2x * d2x + d2y * d2y + d2z * d2z;
    const double S1 = d1x * d12x + d1y * d12y + d1z * d12z;
    const double S2 = d2x * d12x + d2y * d12y + d2z * d12z;
    const double R = d1x * d2x + d1y * d2y + d1z * d2z;
    const double den = D1 * D2 - R * R;
    double t, u;
    if (D1 == 0.0 || D2 == 0.0) {
        if (D1 != 0.0) { u = 0.0; t = fixbound(fdiv(S1, D1)); }
        else if (D2 != 0.0) { t = 0.0; u = fixbound(fdiv(-S2, D2)); }
        else { t = 0.0; u = 0.0; }
    } else if (den == 0.0) {
        t = 0.0;
        u = fdiv(-S2, D2);
        const double uf = fixbound(u);
        if (uf != u) { t = fixbound(fdiv(uf * R + S1, D1)); u = uf; }
    } else {
        t = fixbound(fdiv(S1 * D2 - S2 * R, den));
        u = fdiv(t * R - S2, D2);
        const double uf = fixbound(u);
        if (uf != u) { t = fixbound(fdiv(uf * R + S1, D1)); u = uf; }
    }
    const double ex = d1x * t - d2x * u - d12x, ey = d1y * t - d2y * u - d12y, ez = d1z * t - d2z * u - d12z;
    double dis = sqrt(ex * ex + ey * ey + ez * ez);
    if (fabs(dis) < 0.0001) {
        const double qx = (a6[0] + d1x * t) - a6[3], qy = (a6[1] + d1y * t) - a6[4], qz = (a6[2] + d1z * t) - a6[5];
        dis = -sqrt(qx * qx + qy * qy + qz * qz);
    }
    return dis;
}

