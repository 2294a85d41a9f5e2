// cfs_mesh.hip -- mesh obstacles: binary-STL loader, BVH, segment-to-mesh distance
// (the `point2surface_dis` the reference calls but does not contain: M200i/dist_arm_surf_200i.m:21,
// Lib/functions/dist_arm_surface.m:43), dist_arm over a mesh, and the linearisation (distance + literal
// num_jac gradient) of mesh obstacles that feeds the fused solver.
//
// Contract (the build's own, DESIGN.md "Mesh obstacles"): dis = min over triangles of the Euclidean distance
// between the link axis and the triangle (0 when they intersect); points = [closest point on the axis; closest
// point on the mesh]; equal distances resolve to the smaller parameter along the axis, so the answer does not
// depend on the traversal order.  The hierarchy only prunes: a node is skipped when a rigorous lower bound of
// its distance exceeds the incumbent.
#include "cfs_geom_dev.h"
#include "cfs_host.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>

namespace {

constexpr int MESH_THREADS = 128;
constexpr int MESH_STACK = 20;              // >= depth of the balanced hierarchy + 2 (checked at build time)
#ifndef CFS_LEAF_TRIS
#define CFS_LEAF_TRIS 2
#endif
constexpr int LEAF_TRIS = CFS_LEAF_TRIS;

// ---- device geometry -------------------------------------------------------------------------------
__device__ __forceinline__ double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void sub3(const double *a, const double *b, double *c) { c[0] = a[0] - b[0]; c[1] = a[1] - b[1]; c[2] = a[2] - b[2]; }
__device__ __forceinline__ void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

// Lib/functions/distLinSeg.m:23-91 with both outputs: distance, parameter t on the first segment, closest points
__device__ double seg_seg_full(const double *p1s, const double *p1e, const double *p2s, const double *p2e, double *t_out, double *pts)
{
    double d1[3], d2[3], d12[3];
    sub3(p1e, p1s, d1); sub3(p2e, p2s, d2); sub3(p2s, p1s, d12);
    const double D1 = dot3(d1, d1), D2 = dot3(d2, d2), S1 = dot3(d1, d12), S2 = dot3(d2, d12), R = dot3(d1, d2);
    const double den = D1 * D2 - R * R;
    double t, u;
    if (D1 == 0.0 || D2 == 0.0) {
        if (D1 != 0.0) { u = 0.0; t = clamp01(S1 / D1); }
        else if (D2 != 0.0) { t = 0.0; u = clamp01(-S2 / D2); }
        else { t = 0.0; u = 0.0; }
    } else if (den == 0.0) {
        t = 0.0;
        u = -S2 / D2;
        const double uf = clamp01(u);
        if (uf != u) { t = clamp01((uf * R + S1) / D1); u = uf; }
    } else {
        t = clamp01((S1 * D2 - S2 * R) / den);
        u = (t * R - S2) / D2;
        const double uf = clamp01(u);
        if (uf != u) { t = clamp01((uf * R + S1) / D1); u = uf; }
    }
    double e[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        e[r] = d1[r] * t - d2[r] * u - d12[r];
        pts[r] = p1s[r] + d1[r] * t;
        pts[3 + r] = p2s[r] + d2[r] * u;
    }
    *t_out = t;
    return sqrt(dot3(e, e));
}

// closest point of triangle ABC to P (Voronoi regions; Ericson, Real-Time Collision Detection 5.1.5)
__device__ void closest_pt_triangle(const double *P, const double *A, const double *B, const double *C, double *Q)
{
    double ab[3], ac[3], ap[3], bp[3], cp[3];
    sub3(B, A, ab); sub3(C, A, ac); sub3(P, A, ap);
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) { Q[0] = A[0]; Q[1] = A[1]; Q[2] = A[2]; return; }
    sub3(P, B, bp);
    const double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0.0 && d4 <= d3) { Q[0] = B[0]; Q[1] = B[1]; Q[2] = B[2]; return; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
        const double v = d1 / (d1 - d3);
#pragma unroll
        for (int r = 0; r < 3; ++r) Q[r] = A[r] + v * ab[r];
        return;
    }
    sub3(P, C, cp);
    const double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) { Q[0] = C[0]; Q[1] = C[1]; Q[2] = C[2]; return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
        const double w = d2 / (d2 - d6);
#pragma unroll
        for (int r = 0; r < 3; ++r) Q[r] = A[r] + w * ac[r];
        return;
    }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
#pragma unroll
        for (int r = 0; r < 3; ++r) Q[r] = B[r] + w * (C[r] - B[r]);
        return;
    }
    const double denom = 1.0 / (va + vb + vc);
    const double v = vb * denom, w = vc * denom;
#pragma unroll
    for (int r = 0; r < 3; ++r) Q[r] = A[r] + ab[r] * v + ac[r] * w;
}

struct Best {                  // incumbent of one query
    double d, t;
    double pts[6];
    int tri;
};
__device__ __forceinline__ void take(Best &b, double dis, double t, const double *pl, const double *pm, int tri)
{
    if (dis < b.d || (dis == b.d && t < b.t)) {
        b.d = dis; b.t = t; b.tri = tri;
#pragma unroll
        for (int r = 0; r < 3; ++r) { b.pts[r] = pl[r]; b.pts[3 + r] = pm[r]; }
    }
}

// segment P0P1 against triangle T (9 doubles)
__device__ void seg_tri_update(const double *P0, const double *P1, const double *T, int tri, Best &b)
{
    const double *A = T, *B = T + 3, *C = T + 6;
    double d[3], ab[3], ac[3], n[3], e0[3], e1[3];
    sub3(P1, P0, d);
    const double D = dot3(d, d);
    sub3(B, A, ab); sub3(C, A, ac);
    cross3(ab, ac, n);
    if (dot3(n, n) > 0.0) {                                  // proper triangle: does the segment pierce it?
        sub3(P0, A, e0); sub3(P1, A, e1);
        const double s0 = dot3(n, e0), s1 = dot3(n, e1);
        if (s0 * s1 <= 0.0 && s0 != s1) {
            const double t = s0 / (s0 - s1);
            double X[3], xa[3], xb[3], xc[3], bc[3], ca[3], c0[3], c1[3], c2[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) X[r] = P0[r] + t * d[r];
            sub3(X, A, xa); sub3(X, B, xb); sub3(X, C, xc);
            sub3(C, B, bc); sub3(A, C, ca);
            cross3(ab, xa, c0); cross3(bc, xb, c1); cross3(ca, xc, c2);
            if (dot3(n, c0) >= 0.0 && dot3(n, c1) >= 0.0 && dot3(n, c2) >= 0.0) { take(b, 0.0, t, X, X, tri); return; }
        }
    }
    {
        double Q[3], e[3];
        closest_pt_triangle(P0, A, B, C, Q);
        sub3(P0, Q, e);
        take(b, sqrt(dot3(e, e)), 0.0, P0, Q, tri);
        closest_pt_triangle(P1, A, B, C, Q);
        sub3(P1, Q, e);
        take(b, sqrt(dot3(e, e)), D == 0.0 ? 0.0 : 1.0, P1, Q, tri);
    }
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
        const double *ea = T + 3 * k, *eb = T + 3 * ((k + 1) % 3);
        double p6[6], tt;
        const double dis = seg_seg_full(P0, P1, ea, eb, &tt, p6);
        double tpar = 0.0;
        if (D != 0.0) { double e[3]; sub3(p6, P0, e); tpar = dot3(e, d) / D; }
        take(b, dis, tpar, p6, p6 + 3, tri);
    }
}

// Rigorous lower bound of dist(segment, box), normally the distance itself.  f(t) = dist^2(P0 + t d, box) is convex and
// piecewise quadratic: on the piece where the set of violated slabs is fixed it is sum_r (e_r + t d_r)^2.  Starting from
// the middle, minimise the current piece and move there; when the minimiser lies in its own piece it is the global one
// (convexity).  That takes 2-3 rounds; a point that sits exactly on a slab boundary can make the pattern alternate, and
// then the bound falls back to a cover of the segment by LB_BALLS balls (radius |d| / (2 LB_BALLS)) plus the box-box
// distance -- any lower bound of a cover is a lower bound of the segment.
#ifndef CFS_LB_BALLS
#define CFS_LB_BALLS 4
#endif
__device__ double node_lower_bound(const double *P0, const double *P1, const double *blo, const double *bhi)
{
    double d[3];
    sub3(P1, P0, d);
    const double len2 = dot3(d, d);
    if (len2 > 0.0) {
        double t = 0.5;
#pragma unroll 1
        for (int it = 0; it < 5; ++it) {
            double A = 0.0, B = 0.0, C = 0.0;
            int pat = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double p = P0[r] + t * d[r];
                if (p < blo[r]) { const double e = P0[r] - blo[r]; A += d[r] * d[r]; B += e * d[r]; C += e * e; pat |= 1 << (2 * r); }
                else if (p > bhi[r]) { const double e = P0[r] - bhi[r]; A += d[r] * d[r]; B += e * d[r]; C += e * e; pat |= 2 << (2 * r); }
            }
            if (pat == 0) return 0.0;                            // the point is inside the box
            const double tn = A > 0.0 ? fmin(1.0, fmax(0.0, -B / A)) : t;
            int pat2 = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double p = P0[r] + tn * d[r];
                if (p < blo[r]) pat2 |= 1 << (2 * r);
                else if (p > bhi[r]) pat2 |= 2 << (2 * r);
            }
            if (pat2 == pat) {                                   // the minimiser of this piece lies in this piece: global minimum
                double v = 0.0;                                  // sum of squares at tn, term by term: no cancellation when the
#pragma unroll                                                   // segment touches the box (A tn^2 + 2 B tn + C would lose it)
                for (int r = 0; r < 3; ++r) {
                    const double p = P0[r] + tn * d[r];
                    const double g = fmax(0.0, fmax(blo[r] - p, p - bhi[r]));
                    v += g * g;
                }
                return sqrt(v) * (1.0 - 1e-12) - 1e-13 * (1.0 + sqrt(C));   // shaved: stays a lower bound under rounding
            }
            t = tn;
        }
    }
    constexpr int NB = CFS_LB_BALLS;
    double bb = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double lo = fmin(P0[r], P1[r]), hi = fmax(P0[r], P1[r]);
        const double g = fmax(0.0, fmax(blo[r] - hi, lo - bhi[r]));
        bb += g * g;
    }
    bb = sqrt(bb);
    if (len2 == 0.0) return bb * (1.0 - 1e-14);              // a point: the box-box bound is the exact point-box distance
    const double rad = sqrt(len2) * (0.5 / NB);
    double sp2 = INFINITY;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const double f = (2 * i + 1) * (0.5 / NB);
        double g2 = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double c = P0[r] + f * d[r];
            const double g = fmax(0.0, fmax(blo[r] - c, c - bhi[r]));
            g2 += g * g;
        }
        sp2 = fmin(sp2, g2);
    }
    const double sp = sqrt(sp2);
    // the balls' bound loses a few ulp in rad and the square root: shave it so that it stays a lower bound
    const double spb = (sp - rad) - 1e-12 * (sp + rad);
    return fmax(bb * (1.0 - 1e-14), fmax(0.0, spb));
}

// Triangles whose distance is within `margin` of the minimum, gathered while a query runs (LDS, strided like the stack).
// A pose shifted by less than margin/2 has its closest triangle among them, so the shifted poses of num_jac need no traversal.
constexpr int NEAR_CAP = 12;
struct NearList {
    int *idx;            // [NEAR_CAP] strided
    float *dd;           // [NEAR_CAP] strided, distances rounded DOWN (an entry is never dropped wrongly, at worst kept needlessly)
    double margin;
    int n;
    bool over;           // more than NEAR_CAP triangles tie within the margin: the caller falls back to traversals
};
template <int STRIDE>
__device__ __forceinline__ void near_compact(NearList &nl, double best)
{
    int w = 0;
    for (int i = 0; i < nl.n; ++i)
        if ((double)nl.dd[i * STRIDE] <= best + nl.margin) { nl.idx[w * STRIDE] = nl.idx[i * STRIDE]; nl.dd[w * STRIDE] = nl.dd[i * STRIDE]; ++w; }
    nl.n = w;
}

// nearest-first traversal; the thread's private stack (node, lower bound) lives in LDS, strided by STRIDE
// `bound`: only triangles closer than this matter to the caller (b.tri stays -1 when there is none)
template <int STRIDE, bool COLLECT>
__device__ void mesh_query(const DevMesh &m, const double *P0, const double *P1, int seed_tri, int *stack, float *lbs, Best &b, NearList *nl,
                           double bound = INFINITY)
{
    b.d = bound; b.t = INFINITY; b.tri = -1;
#pragma unroll
    for (int r = 0; r < 6; ++r) b.pts[r] = 0.0;
    if (m.nt == 0) return;
    if (seed_tri >= 0) seg_tri_update(P0, P1, m.tri + 9 * (size_t)seed_tri, seed_tri, b);   // incumbent from a nearby query
    int sp = 0;
    int cur = 0;                                            // the root is always an inner node (upload_mesh)
    const double slack = COLLECT ? nl->margin : 0.0;        // with a collector, everything within the margin must be visited
    for (;;) {
        if (cur < 0) {                                      // leaf
            const int code = -(cur + 1), first = code >> 3, count = code & 7;
            for (int k = first; k < first + count; ++k) {
                if (k == seed_tri) continue;
                if (COLLECT) {
                    Best tb;
                    tb.d = INFINITY; tb.t = INFINITY; tb.tri = -1;
                    seg_tri_update(P0, P1, m.tri + 9 * (size_t)k, k, tb);
                    take(b, tb.d, tb.t, tb.pts, tb.pts + 3, k);
                    if (!nl->over && tb.d <= b.d + nl->margin) {
                        if (nl->n == NEAR_CAP) near_compact<STRIDE>(*nl, b.d);
                        if (nl->n == NEAR_CAP) nl->over = true;
                        else { nl->idx[nl->n * STRIDE] = k; nl->dd[nl->n * STRIDE] = __double2float_rd(tb.d); ++nl->n; }
                    }
                } else {
                    seg_tri_update(P0, P1, m.tri + 9 * (size_t)k, k, b);
                }
            }
            cur = 0x7fffffff;
        } else {
            const BvhNode nd = m.nodes[cur];                // one load: both children's boxes
            const double ll = node_lower_bound(P0, P1, nd.lo[0], nd.hi[0]);
            const double lr = node_lower_bound(P0, P1, nd.lo[1], nd.hi[1]);
            const int nearc = ll <= lr ? nd.child[0] : nd.child[1], farc = ll <= lr ? nd.child[1] : nd.child[0];
            const double ln = fmin(ll, lr), lf = fmax(ll, lr);
            cur = 0x7fffffff;
            if (ln <= b.d + slack) {
                cur = nearc;
                if (lf <= b.d + slack && sp < MESH_STACK) { stack[sp * STRIDE] = farc; lbs[sp * STRIDE] = __double2float_rd(lf); ++sp; }
            }
        }
        while (cur == 0x7fffffff) {
            if (sp == 0) { if (COLLECT) near_compact<STRIDE>(*nl, b.d); return; }
            --sp;
            if ((double)lbs[sp * STRIDE] <= b.d + slack) cur = stack[sp * STRIDE];   // the incumbent may have improved since the push
        }
    }
}

// ---- kernels ------------------------------------------------------------------------------------------
struct SegQueryParams { DevMesh m; int n; const double *segs; double *dis, *pts; int *tri; };

__global__ __launch_bounds__(MESH_THREADS) void cfs_mesh_seg_kernel(SegQueryParams P)
{
    __shared__ int s_stack[MESH_STACK * MESH_THREADS];
    __shared__ float s_lbs[MESH_STACK * MESH_THREADS];
    const int i = blockIdx.x * MESH_THREADS + threadIdx.x;
    if (i >= P.n) return;
    double seg[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) seg[r] = P.segs[(size_t)i * 6 + r];
    Best b;
    mesh_query<MESH_THREADS, false>(P.m, seg, seg + 3, -1, s_stack + threadIdx.x, s_lbs + threadIdx.x, b, nullptr);
    P.dis[i] = b.d;
    if (P.pts)
#pragma unroll
        for (int r = 0; r < 6; ++r) P.pts[(size_t)i * 6 + r] = b.pts[r];
    if (P.tri) P.tri[i] = b.tri >= 0 ? P.m.orig[b.tri] : -1;
}

// dist_arm over a mesh (M200i/dist_arm_surf_200i.m:1-29): one thread per (pose, link), min over links in LDS
struct ArmMeshParams { const DevRobot *rb; DevMesh m; int N, nj; const double *theta; double *d; int *linkid; double *pts; };

__global__ __launch_bounds__(MESH_THREADS) void cfs_dist_arm_mesh_kernel(ArmMeshParams P)
{
    __shared__ int s_stack[MESH_STACK * MESH_THREADS];
    __shared__ float s_lbs[MESH_STACK * MESH_THREADS];
    __shared__ double s_dis[MESH_THREADS];
    __shared__ double s_pts[MESH_THREADS * 6];
    const int nj = P.nj, per = MESH_THREADS / nj;           // poses per workgroup
    const int lp = threadIdx.x / nj, k = threadIdx.x - lp * nj;
    const int pose = blockIdx.x * per + lp;
    const bool live = lp < per && pose < P.N;
    double dis = INFINITY;
    if (live) {
        const DevRobot *rb = P.rb;
        double M[12], Mn[12], e6[6];
        for (int k1 = 0; k1 <= k; ++k1) {                    // CapPos.m:13-20 up to this thread's link
            double sn, cs;
            sincos(P.theta[(size_t)pose * nj + k1] - rb->th_off[k1], &sn, &cs);
            fk_step(rb, k1, sn, cs, k1 == 0 ? nullptr : M, Mn);
            for (int q = 0; q < 12; ++q) M[q] = Mn[q];
        }
        link_ends(rb, k, M, e6);
        Best b;
        mesh_query<MESH_THREADS, false>(P.m, e6, e6 + 3, -1, s_stack + threadIdx.x, s_lbs + threadIdx.x, b, nullptr);
        dis = b.d;
        if (fabs(dis) < 0.0001) {                            // dist_arm_surf_200i.m:22-24
            const double qx = b.pts[0] - e6[3], qy = b.pts[1] - e6[4], qz = b.pts[2] - e6[5];
            dis = -sqrt(qx * qx + qy * qy + qz * qz);
        }
        for (int r = 0; r < 6; ++r) s_pts[threadIdx.x * 6 + r] = b.pts[r];
    }
    s_dis[threadIdx.x] = dis;
    __syncthreads();
    if (live && k == 0) {
        double d = INFINITY;
        int id = 0;
        for (int i = 0; i < nj; ++i) { const double v = s_dis[lp * nj + i]; if (v < d) { d = v; id = i + 1; } }   // :25-28, first minimum wins
        P.d[pose] = d;
        if (P.linkid) P.linkid[pose] = id;
        if (P.pts && id > 0)
            for (int r = 0; r < 6; ++r) P.pts[(size_t)pose * 6 + r] = s_pts[(lp * nj + id - 1) * 6 + r];
    }
}

// Linearisation against mesh obstacles: the same scheme as cfs_fused.hip (link variants, base distances, pruned
// candidates, minima per evaluation point of num_jac), the distance being a hierarchy query.  Four kernels, every one
// with a thread per item, so that the expensive cold traversals of the base pose fill the machine:
//   fk     (b, waypoint, evaluation point) -> end points of every link variant
//   upper  (b, waypoint, link, mesh)       -> greedy upper bound of the base-pose distance (one descent)
//   base   (b, waypoint, link, mesh, piece)-> distance + winning triangle at the base pose per quarter of the link axis, for what can matter
//   reduce (b, waypoint, link, mesh)       -> the link's minimum over its pieces, surrogate, merged near-tie list
//   shift  (b, waypoint, mesh, variant)    -> distance at the shifted poses of the links that can be the minimum there,
//                                             started from the base pose's triangle (their traversals prune almost everything)
//   fd     (b, waypoint, mesh)             -> minima per evaluation point, distance and literal num_jac gradient
template <int NJ>
__global__ __launch_bounds__(MESH_THREADS) void mesh_fk_kernel(LinMeshParams P)
{
    constexpr int NS = 2 * NJ, NVT = nvt(NJ), NE = 2 * NJ + 1;
    const int e = blockIdx.x * MESH_THREADS + threadIdx.x;
    if (e >= P.B * P.H * NE) return;
    const int ev = e % NE, wp = (e / NE) % P.H, b = e / (NE * P.H);
    if (P.status_done && P.status_done[b] != 0) return;
    const DevRobot *rb = P.rb;
    double M[12], Mn[12], e6[6];
#pragma unroll
    for (int k1 = 1; k1 <= NJ; ++k1) {
        // evaluation point ev has joint m at +eps/2 if ev == 2m-1 and at -eps/2 if ev >= 2m (num_jac.m:8-14: xp is never restored)
        double x = P.x_[((size_t)b * P.H + wp) * NS + k1 - 1];
        if (ev == 2 * k1 - 1) x = x + FD_EPS / 2;            // num_jac.m:11
        else if (ev >= 2 * k1) x = x - FD_EPS / 2;           // num_jac.m:13
        x = x - rb->th_off[k1 - 1];                          // same joint offset as dist_arm_3D_200i_2.m:11
        double sn, cs;
        sincos(x, &sn, &cs);
        fk_step(rb, k1 - 1, sn, cs, k1 == 1 ? nullptr : M, Mn);
#pragma unroll
        for (int q = 0; q < 12; ++q) M[q] = Mn[q];
        if (ev <= 2 * k1) {                                  // link k1 at evaluation point ev is variant min(ev, 2 k1): stored by its owner
            link_ends(rb, k1 - 1, M, e6);
            double *dst = P.ends + (((size_t)b * P.H + wp) * NVT + kvoff(k1) + ev) * 6;
#pragma unroll
            for (int q = 0; q < 6; ++q) dst[q] = e6[q];
        }
    }
}

__device__ __forceinline__ double with_surrogate(const Best &bq, const double *a6)
{
    double dis = bq.d;
    if (fabs(dis) < 0.0001) {                                // dist_arm_surf_200i.m:22-24
        const double qx = bq.pts[0] - a6[3], qy = bq.pts[1] - a6[4], qz = bq.pts[2] - a6[5];
        dis = -sqrt(qx * qx + qy * qy + qz * qz);
    }
    return dis;
}

// one descent to the nearest leaf, no backtracking: an upper bound of the distance for ~2 log2(nt) box bounds
__device__ double mesh_greedy_upper(const DevMesh &m, const double *P0, const double *P1)
{
    if (m.nt == 0) return INFINITY;
    int cur = 0;
    for (;;) {
        if (cur < 0) {
            const int code = -(cur + 1), first = code >> 3, count = code & 7;
            Best b;
            b.d = INFINITY; b.t = INFINITY; b.tri = -1;
            for (int k = first; k < first + count; ++k) seg_tri_update(P0, P1, m.tri + 9 * (size_t)k, k, b);
            return b.d;
        }
        const BvhNode nd = m.nodes[cur];
        cur = node_lower_bound(P0, P1, nd.lo[0], nd.hi[0]) <= node_lower_bound(P0, P1, nd.lo[1], nd.hi[1]) ? nd.child[0] : nd.child[1];
    }
}

template <int NJ>
__global__ __launch_bounds__(MESH_THREADS, 2) void mesh_upper_kernel(LinMeshParams P)
{
    constexpr int NVT = nvt(NJ);
    const int e = blockIdx.x * MESH_THREADS + threadIdx.x;
    if (e >= P.B * P.H * NJ * P.nmesh) return;
    const int wp = e % P.H, b = (e / P.H) % P.B, jm = (e / (P.H * P.B)) % P.nmesh, k0 = e / (P.H * P.B * P.nmesh);
    if (P.status_done && P.status_done[b] != 0) return;
    double a6[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) a6[q] = P.ends[(((size_t)b * P.H + wp) * NVT + kvoff(k0 + 1)) * 6 + q];
    const size_t o = (((size_t)b * P.H + wp) * NJ + k0) * P.nmesh + jm;
    double up = mesh_greedy_upper(P.meshes[jm], a6, a6 + 3);
    if (P.seed_prev) {
        const int tp = P.base_t[o];
        if (tp >= 0 && tp < P.meshes[jm].nt) {
            Best bp;
            bp.d = INFINITY; bp.t = INFINITY; bp.tri = -1;
            seg_tri_update(a6, a6 + 3, P.meshes[jm].tri + 9 * (size_t)tp, tp, bp);
            up = fmin(up, bp.d);
        }
    }
    P.upper_d[o] = up;
}

// A link axis is queried in MESH_PIECES equal pieces: the pieces far from the surface die at the root against the common
// bound, and the others have short verification sets.  dist(link, T) = min over the pieces of dist(piece, T).
#ifndef CFS_MESH_PIECES
#define CFS_MESH_PIECES 4
#endif
constexpr int MESH_PIECES = CFS_MESH_PIECES;
constexpr int PIECE_D = 5;                                   // per piece: raw distance, parameter along the whole axis, closest point on the axis
constexpr int PIECE_I = 2 + NEAR_CAP;                        // winning triangle, near count (-1: overflow), near triangles

// ---- wave-cooperative query: one wavefront works on ONE query at a time, 64 hierarchy nodes or 64 triangles per round, so no
// lane waits for another lane's longer traversal.  Same semantics as mesh_query<.., true> (lexicographic minimum, near list).
#ifndef CFS_CO_QPW
#define CFS_CO_QPW 16
#endif
constexpr int CO_FR = 768, CO_TR = 320, CO_QPW = CFS_CO_QPW;   // frontier / triangle buffer of a wavefront; queries set up per wavefront
static_assert(64 + 128 * LEAF_TRIS <= CO_TR, "a round of 64 nodes can emit 128 leaves");
struct CoopLds {
    int fr[CO_FR];
    int tr[CO_TR];
    double cb[128 * 6];      // boxes (lo, hi) of the children a round could not rule out cheaply, compacted
    int cc[128];             // their child codes
    int ni[NEAR_CAP];
    float nd[NEAR_CAP];
};
__device__ __forceinline__ double wmin64(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ int prefix_of(unsigned long long mask, int lane) { return __popcll(mask & ((1ull << lane) - 1ull)); }

// false: a buffer would overflow (the caller repeats the query sequentially)
__device__ bool mesh_query_coop(const DevMesh &m, const double *P0, const double *P1, CoopLds &L, int lane, double bound, double margin,
                                Best &b, int &near_n, bool &near_over)
{
    b.d = bound; b.t = INFINITY; b.tri = -1;
#pragma unroll
    for (int r = 0; r < 6; ++r) b.pts[r] = 0.0;
    near_n = 0; near_over = false;
    if (m.nt == 0) return true;
    int nf = m.ncut, nt = m.ncut_tri;                           // start at the cut (<= 64 inner nodes, <= 254 triangles), not at the root
    if (lane < nf) L.fr[lane] = m.cut[lane];
    for (int k = lane; k < nt; k += 64) L.tr[k] = m.cut_tri[k];
    __syncthreads();
    while (nf > 0 || nt > 0) {
        if (nf > 0 && nt < 64) {
            // expand up to 64 inner nodes from the end of the frontier (depth first in blocks of 64)
            if (nf + 64 > CO_FR) return false;
            const int take = min(nf, 64);
            nf -= take;
            const bool act = lane < take;
            const int ref = act ? L.fr[nf + lane] : 0;
            __syncthreads();
            // Two passes.  Most children of a round are far away: the box-box distance between the segment's bounding box and
            // the child's (a lower bound of the true distance, ~20 flops) rules them out.  An inner child that passes is
            // opened; a leaf that passes is compacted through LDS, so the exact segment-box bound (2-5 Newton rounds,
            // divergent) runs once over dense lanes and only where it saves triangle tests.
            const double lim = b.d + margin, lim2 = lim * lim * (1.0 + 1e-13);
            int nc = 0;
            {
                BvhNode nd;
                if (act) nd = m.nodes[ref];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bool cand = false;
                    if (act) {
                        double g2 = 0.0;
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const double lo = fmin(P0[r], P1[r]), hi = fmax(P0[r], P1[r]);
                            const double g = fmax(0.0, fmax(nd.lo[c][r] - hi, lo - nd.hi[c][r]));
                            g2 += g * g;
                        }
                        cand = !(g2 * (1.0 - 1e-13) > lim2);      // an empty child has lo = +inf: g2 = inf, never a candidate
                    }
                    const bool inner = cand && nd.child[c] >= 0;
                    const unsigned long long mi = __ballot(inner);
                    if (inner) L.fr[nf + prefix_of(mi, lane)] = nd.child[c];
                    nf += __popcll(mi);
                    const bool leafc = cand && nd.child[c] < 0 && ((-(nd.child[c] + 1)) & 7) > 0;
                    const unsigned long long mk = __ballot(leafc);
                    if (leafc) {
                        const int sl = nc + prefix_of(mk, lane);
#pragma unroll
                        for (int r = 0; r < 3; ++r) { L.cb[sl * 6 + r] = nd.lo[c][r]; L.cb[sl * 6 + 3 + r] = nd.hi[c][r]; }
                        L.cc[sl] = nd.child[c];
                    }
                    nc += __popcll(mk);
                }
            }
            __syncthreads();
            for (int base = 0; base < nc; base += 64) {
                const int sl = base + lane;
                const bool has_c = sl < nc;
                double lcx = INFINITY;
                int chx = 0;
                if (has_c) {
                    double blo[3], bhi[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) { blo[r] = L.cb[sl * 6 + r]; bhi[r] = L.cb[sl * 6 + 3 + r]; }
                    chx = L.cc[sl];
                    // measured per solve, exact bound everywhere / nowhere / leaves only: config 5 9.6 / 8.3 / 8.7 ms, one
                    // 5 120-triangle sphere 116 / 122 / 110 ms, posts + beam 17.3 / 15.9 / 16.3 ms
                    lcx = node_lower_bound(P0, P1, blo, bhi);
                }
                const bool q = has_c && lcx <= lim;
                const int code = -(chx + 1), first = code >> 3, cnt = q ? (code & 7) : 0;
#pragma unroll
                for (int k = 0; k < LEAF_TRIS; ++k) {
                    const bool has = k < cnt;
                    const unsigned long long mt = __ballot(has);
                    if (has) L.tr[nt + prefix_of(mt, lane)] = first + k;
                    nt += __popcll(mt);
                }
            }
            __syncthreads();
        } else {
            // test up to 64 triangles, one per lane
            const int take = min(nt, 64);
            nt -= take;
            const int k = lane < take ? L.tr[nt + lane] : -1;
            __syncthreads();
            Best tb;
            tb.d = INFINITY; tb.t = INFINITY; tb.tri = -1;
#pragma unroll
            for (int r = 0; r < 6; ++r) tb.pts[r] = 0.0;
            if (k >= 0) seg_tri_update(P0, P1, m.tri + 9 * (size_t)k, k, tb);
            const double dmin = wmin64(tb.d);
            const double tmin = wmin64(tb.d == dmin ? tb.t : INFINITY);
            if (dmin < b.d || (dmin == b.d && tmin < b.t)) {
                const unsigned long long win = __ballot(tb.d == dmin && tb.t == tmin);
                const int wl = (int)__builtin_ctzll(win);
                b.d = dmin; b.t = tmin; b.tri = __shfl(tb.tri, wl, 64);
#pragma unroll
                for (int r = 0; r < 6; ++r) b.pts[r] = __shfl(tb.pts[r], wl, 64);
            }
            if (!near_over) {
                const bool nr = k >= 0 && tb.d <= b.d + margin;
                const unsigned long long mn = __ballot(nr);
                const int add = __popcll(mn);
                if (near_n + add > NEAR_CAP) {                // compact what is there against the current minimum first
                    const bool keep = lane < near_n && (double)L.nd[lane] <= b.d + margin;
                    const int ki = lane < near_n ? L.ni[lane] : 0;
                    const float kd = lane < near_n ? L.nd[lane] : 0.0f;
                    const unsigned long long mkp = __ballot(keep);
                    __syncthreads();
                    if (keep) { L.ni[prefix_of(mkp, lane)] = ki; L.nd[prefix_of(mkp, lane)] = kd; }
                    near_n = __popcll(mkp);
                    __syncthreads();
                }
                if (near_n + add > NEAR_CAP) near_over = true;
                else {
                    if (nr) { L.ni[near_n + prefix_of(mn, lane)] = k; L.nd[near_n + prefix_of(mn, lane)] = __double2float_rd(tb.d); }
                    near_n += add;
                }
                __syncthreads();
            }
        }
    }
    if (!near_over) {                                          // final filter against the final minimum
        const bool keep = lane < near_n && (double)L.nd[lane] <= b.d + margin;
        const int ki = lane < near_n ? L.ni[lane] : 0;
        const float kd = lane < near_n ? L.nd[lane] : 0.0f;
        const unsigned long long mkp = __ballot(keep);
        __syncthreads();
        if (keep) { L.ni[prefix_of(mkp, lane)] = ki; L.nd[prefix_of(mkp, lane)] = kd; }
        near_n = __popcll(mkp);
        __syncthreads();
    }
    return true;
}

template <int NJ>
__global__ __launch_bounds__(64) void mesh_base_kernel(LinMeshParams P)
{
    constexpr int NVT = nvt(NJ);
    __shared__ CoopLds L;
    const int lane = threadIdx.x;
    const int total = P.B * P.H * NJ * P.nmesh * MESH_PIECES;
    // Stage 1, a query per lane: set it up and test it against the root's two boxes.  Most pieces of most links have nothing
    // within their bound and end here.  Piece and link index fastest: every wavefront gets its share of the few that survive
    // (stage 2 takes them one after the other).
    const int e = blockIdx.x * CO_QPW + lane;
    bool live = lane < CO_QPW && e < total;
    int jm = 0, pc = 0;
    size_t o = 0;
    bool point = false;
    double s6[6] = {0, 0, 0, 0, 0, 0}, bound = 0.0;
    const double margin = 4.0 * P.rb->shift_bound;              // twice what the argument needs (see NearList)
    if (live) {
        pc = e % MESH_PIECES;
        const int k0 = (e / MESH_PIECES) % NJ;
        jm = (e / (MESH_PIECES * NJ)) % P.nmesh;
        const int wp = (e / (MESH_PIECES * NJ * P.nmesh)) % P.H, b = e / (MESH_PIECES * NJ * P.nmesh * P.H);
        live = !(P.status_done && P.status_done[b] != 0);
        o = (((size_t)b * P.H + wp) * NJ + k0) * P.nmesh + jm;
        if (live) {
            double a6[6], d[3];
#pragma unroll
            for (int q = 0; q < 6; ++q) a6[q] = P.ends[(((size_t)b * P.H + wp) * NVT + kvoff(k0 + 1)) * 6 + q];
            sub3(a6 + 3, a6, d);
            point = dot3(d, d) == 0.0;                           // M200i links 1 and 3 are points: one "piece"
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                s6[r] = point || pc == 0 ? a6[r] : a6[r] + ((double)pc / MESH_PIECES) * d[r];
                s6[3 + r] = point || pc == MESH_PIECES - 1 ? a6[3 + r] : a6[r] + ((double)(pc + 1) / MESH_PIECES) * d[r];
            }
            // Only the minimum over the links and the links within prune_tol of it matter (see mesh_shift_kernel), and the
            // minimum is at most the smallest greedy upper bound: a piece with nothing closer than `bound` is out.
            double umin = INFINITY;
#pragma unroll
            for (int kk = 0; kk < NJ; ++kk) umin = fmin(umin, P.upper_d[(((size_t)b * P.H + wp) * NJ + kk) * P.nmesh + jm]);
            bound = (fmax(umin, 0.0001) + P.rb->prune_tol) * (1.0 + 1e-12) + margin;
            double *rec = P.piece_d + (o * MESH_PIECES + pc) * PIECE_D;
            int *reci = P.piece_i + (o * MESH_PIECES + pc) * PIECE_I;
            rec[0] = INFINITY; rec[1] = INFINITY; rec[2] = rec[3] = rec[4] = 0.0; reci[0] = -1; reci[1] = 0;   // "nothing within the bound"
            bool go = !(point && pc > 0) && P.meshes[jm].nt > 0;
            if (go) {                                            // box of the piece against the boxes of the root's children (cheap; the
                const BvhNode nd = P.meshes[jm].nodes[0];        // cut round of stage 2 is cheap too for what slips through)
                const double lim = bound + margin, lim2 = lim * lim * (1.0 + 1e-13);
                double g2min = INFINITY;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    double g2 = 0.0;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const double lo = fmin(s6[r], s6[3 + r]), hi = fmax(s6[r], s6[3 + r]);
                        const double g = fmax(0.0, fmax(nd.lo[c][r] - hi, lo - nd.hi[c][r]));
                        g2 += g * g;
                    }
                    g2min = fmin(g2min, g2);
                }
                go = !(g2min * (1.0 - 1e-13) > lim2);
            }
            live = go;
        }
    }
    // Stage 2, the survivors one at a time, the whole wavefront on each.
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int src = (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        double q6[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) q6[r] = __shfl(s6[r], src, 64);
        const double qbound = __shfl(bound, src, 64);
        const int qjm = __shfl(jm, src, 64), qpc = __shfl(pc, src, 64);
        const bool qpoint = __shfl((int)point, src, 64) != 0;
        const unsigned long long qo = __shfl((unsigned long long)o, src, 64);
        double *rec = P.piece_d + (qo * MESH_PIECES + qpc) * PIECE_D;
        int *reci = P.piece_i + (qo * MESH_PIECES + qpc) * PIECE_I;
        double *recn = P.piece_nd + (qo * MESH_PIECES + qpc) * NEAR_CAP;
        Best bq;
        int near_n;
        bool near_over;
        __syncthreads();
        if (!mesh_query_coop(P.meshes[qjm], q6, q6 + 3, L, lane, qbound, margin, bq, near_n, near_over)) {
            // a buffer was about to overflow: lane 0 repeats the query with the sequential traversal (its stack in the same LDS)
            __syncthreads();
            NearList nl;
            nl.idx = L.ni; nl.dd = L.nd; nl.n = 0; nl.over = false; nl.margin = margin;
            if (lane == 0) mesh_query<1, true>(P.meshes[qjm], q6, q6 + 3, -1, L.fr, reinterpret_cast<float *>(L.tr), bq, &nl, qbound);
            __syncthreads();
            bq.d = __shfl(bq.d, 0, 64); bq.t = __shfl(bq.t, 0, 64); bq.tri = __shfl(bq.tri, 0, 64);
#pragma unroll
            for (int r = 0; r < 6; ++r) bq.pts[r] = __shfl(bq.pts[r], 0, 64);
            near_n = __shfl(nl.n, 0, 64);
            near_over = __shfl((int)nl.over, 0, 64) != 0;
        }
        if (bq.tri >= 0) {
            if (lane == 0) {
                rec[0] = bq.d;
                rec[1] = qpoint ? 0.0 : ((double)qpc + bq.t) / MESH_PIECES;
                rec[2] = bq.pts[0]; rec[3] = bq.pts[1]; rec[4] = bq.pts[2];
                reci[0] = bq.tri;
                reci[1] = near_over ? -1 : near_n;
            }
            if (!near_over && lane < near_n) { reci[2 + lane] = L.ni[lane]; recn[lane] = (double)L.nd[lane]; }
        }
        __syncthreads();
    }
}

// pieces -> link: lexicographic minimum (distance, parameter), surrogate, and the union of the pieces' near lists
template <int NJ>
__global__ __launch_bounds__(MESH_THREADS) void mesh_reduce_kernel(LinMeshParams P)
{
    constexpr int NVT = nvt(NJ);
    const int e = blockIdx.x * MESH_THREADS + threadIdx.x;
    if (e >= P.B * P.H * NJ * P.nmesh) return;
    const int jm = e % P.nmesh, k0 = (e / P.nmesh) % NJ, wp = (e / (P.nmesh * NJ)) % P.H, b = e / (P.nmesh * NJ * P.H);
    if (P.status_done && P.status_done[b] != 0) return;
    const size_t o = (((size_t)b * P.H + wp) * NJ + k0) * P.nmesh + jm;
    Best bq;
    bq.d = INFINITY; bq.t = INFINITY; bq.tri = -1;
    for (int pc = 0; pc < MESH_PIECES; ++pc) {
        const double *rec = P.piece_d + (o * MESH_PIECES + pc) * PIECE_D;
        const int tri = P.piece_i[(o * MESH_PIECES + pc) * PIECE_I];
        if (tri >= 0) take(bq, rec[0], rec[1], rec + 2, rec + 2, tri);
    }
    int *near = P.near + o * (NEAR_CAP + 1);
    if (bq.tri < 0) { P.base_d[o] = INFINITY; P.base_t[o] = -1; near[0] = 0; return; }   // farther than the bound: never the minimum, never a candidate
    const double *a6 = P.ends + (((size_t)b * P.H + wp) * NVT + kvoff(k0 + 1)) * 6;
    P.base_d[o] = with_surrogate(bq, a6);
    P.base_t[o] = bq.tri;
    const double lim = bq.d + 4.0 * P.rb->shift_bound;
    int n = 0;
    bool over = false;
    for (int pc = 0; pc < MESH_PIECES && !over; ++pc) {
        const int *reci = P.piece_i + (o * MESH_PIECES + pc) * PIECE_I;
        const double *recn = P.piece_nd + (o * MESH_PIECES + pc) * NEAR_CAP;
        if (reci[0] < 0) continue;
        // a piece whose own minimum is beyond the limit contributes nothing, whatever its list says
        if (P.piece_d[(o * MESH_PIECES + pc) * PIECE_D] > lim) continue;
        if (reci[1] < 0) { over = true; break; }
        for (int i = 0; i < reci[1]; ++i) {
            if (recn[i] > lim) continue;
            bool dup = false;
            for (int w = 0; w < n; ++w) dup = dup || near[1 + w] == reci[2 + i];
            if (dup) continue;
            if (n == NEAR_CAP) { over = true; break; }
            near[1 + n++] = reci[2 + i];
        }
    }
    near[0] = over ? -1 : n;                                 // < 0: too many ties, the shifted poses traverse instead
}

template <int NJ>
__global__ __launch_bounds__(MESH_THREADS, 2) void mesh_shift_kernel(LinMeshParams P)
{
    constexpr int NVT = nvt(NJ), NSV = NVT - NJ;
    __shared__ int s_stack[MESH_STACK * MESH_THREADS];
    __shared__ float s_lbs[MESH_STACK * MESH_THREADS];
    const int e = blockIdx.x * MESH_THREADS + threadIdx.x;
    if (e >= P.B * P.H * P.nmesh * NSV) return;
    const int wp = e % P.H, b = (e / P.H) % P.B, jm = (e / (P.H * P.B)) % P.nmesh, sv = e / (P.H * P.B * P.nmesh);   // variant slowest (see base)
    if (P.status_done && P.status_done[b] != 0) return;
    int k1 = 1, v = 1;                                       // sv enumerates (link k1, variant v = 1..2 k1)
    {
        int acc = 0;
#pragma unroll
        for (int kk = 1; kk <= NJ; ++kk) { if (sv >= acc && sv < acc + 2 * kk) { k1 = kk; v = sv - acc + 1; } acc += 2 * kk; }
    }
    const size_t bo = (((size_t)b * P.H + wp) * NJ) * P.nmesh + jm;
    double m0 = INFINITY;
#pragma unroll
    for (int k0 = 0; k0 < NJ; ++k0) m0 = fmin(m0, P.base_d[bo + (size_t)k0 * P.nmesh]);
    // a link farther than prune_tol above max(min, 1e-4) cannot be the minimum nor reach the surrogate at a shifted pose
    const double thr = fmax(m0, 0.0001) + P.rb->prune_tol;
    double out = INFINITY;
    if (P.base_d[bo + (size_t)(k1 - 1) * P.nmesh] < thr) {
        double a6[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) a6[q] = P.ends[(((size_t)b * P.H + wp) * NVT + kvoff(k1) + v) * 6 + q];
        Best bq;
        const size_t ob = bo + (size_t)(k1 - 1) * P.nmesh;
        const int *near = P.near + ob * (NEAR_CAP + 1);
        const int nn_ = near[0];
        if (nn_ >= 0) {                                      // the pose moved by < margin/2: its closest triangle is on the list
            const DevMesh &m = P.meshes[jm];
            bq.d = INFINITY; bq.t = INFINITY; bq.tri = -1;
            for (int i = 0; i < nn_; ++i) { const int k = near[1 + i]; seg_tri_update(a6, a6 + 3, m.tri + 9 * (size_t)k, k, bq); }
        } else {
            mesh_query<MESH_THREADS, false>(P.meshes[jm], a6, a6 + 3, P.base_t[ob], s_stack + threadIdx.x, s_lbs + threadIdx.x, bq, nullptr);
        }
        out = with_surrogate(bq, a6);
    }
    P.shift_d[(((size_t)b * P.H + wp) * P.nmesh + jm) * NSV + sv] = out;   // +inf: not a candidate
}

template <int NJ>
__global__ __launch_bounds__(MESH_THREADS) void mesh_fd_kernel(LinMeshParams P)
{
    constexpr int NVT = nvt(NJ), NSV = NVT - NJ, NE = 2 * NJ + 1;
    const int e = blockIdx.x * MESH_THREADS + threadIdx.x;
    if (e >= P.B * P.H * P.nmesh) return;
    const int jm = e % P.nmesh, wp = (e / P.nmesh) % P.H, b = e / (P.nmesh * P.H);
    if (P.status_done && P.status_done[b] != 0) return;
    const size_t bo = (((size_t)b * P.H + wp) * NJ) * P.nmesh + jm;
    const double *sh = P.shift_d + (((size_t)b * P.H + wp) * P.nmesh + jm) * NSV;
    double dev[NE];
    dev[0] = INFINITY;
#pragma unroll
    for (int k0 = 0; k0 < NJ; ++k0) dev[0] = fmin(dev[0], P.base_d[bo + (size_t)k0 * P.nmesh]);
#pragma unroll
    for (int ev = 1; ev < NE; ++ev) {
        double d = INFINITY;
#pragma unroll
        for (int k1 = 1; k1 <= NJ; ++k1) {
            const double dis = sh[k1 * (k1 - 1) + min(ev, 2 * k1) - 1];   // variants of link k1 start at sum_{k<k1} 2k = k1 (k1-1)
            if (dis < d) d = dis;
        }
        dev[ev] = d;
    }
    const size_t o = ((size_t)b * P.nmesh + jm) * P.H + wp;
    P.dist[o] = dev[0];
#pragma unroll
    for (int m = 0; m < NJ; ++m) P.grad[o * NJ + m] = (dev[2 * m + 1] - dev[2 * m + 2]) / FD_EPS;   // num_jac.m:14
}

// ---- host: hierarchy ------------------------------------------------------------------------------------
struct Builder {
    const double *tri;             // nt x 9, caller order
    std::vector<int> order;        // permutation being built
    std::vector<double> cen;       // nt x 3 centroids
    std::vector<BvhNode> nodes;
    int depth = 0;

    // builds the subtree of triangles [first, first + count); returns its reference (inner node index, or leaf code < 0) and box
    int build(int first, int count, int level, double *lo, double *hi)
    {
        depth = std::max(depth, level);
        double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int r = 0; r < 3; ++r) { lo[r] = INFINITY; hi[r] = -INFINITY; }
        for (int i = first; i < first + count; ++i) {
            const double *T = tri + 9 * (size_t)order[i];
            for (int v = 0; v < 3; ++v)
                for (int r = 0; r < 3; ++r) { lo[r] = std::min(lo[r], T[3 * v + r]); hi[r] = std::max(hi[r], T[3 * v + r]); }
            for (int r = 0; r < 3; ++r) { clo[r] = std::min(clo[r], cen[3 * (size_t)order[i] + r]); chi[r] = std::max(chi[r], cen[3 * (size_t)order[i] + r]); }
        }
        if (count <= LEAF_TRIS) return -(first * 8 + count) - 1;
        const int id = (int)nodes.size();
        nodes.emplace_back();
        int ax = 0;
        for (int r = 1; r < 3; ++r) if (chi[r] - clo[r] > chi[ax] - clo[ax]) ax = r;
        const int mid = first + count / 2;                    // median split: balanced, depth <= ceil(log2(nt)) + 1
        std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                         [&](int a, int c) { const double ca = cen[3 * (size_t)a + ax], cc = cen[3 * (size_t)c + ax]; return ca < cc || (ca == cc && a < c); });
        BvhNode nd;
        memset(&nd, 0, sizeof nd);
        nd.child[0] = build(first, mid - first, level + 1, nd.lo[0], nd.hi[0]);
        nd.child[1] = build(mid, first + count - mid, level + 1, nd.lo[1], nd.hi[1]);
        nodes[id] = nd;
        return id;
    }
};

int upload_mesh(const std::vector<double> &tri9, cfs_mesh **out)
{
    const int nt = (int)(tri9.size() / 9);
    for (double v : tri9) if (!std::isfinite(v)) return cfs_fail(CFS_ERR_INVALID_ARG, "mesh has a non-finite coordinate");
    Builder bd;
    bd.tri = tri9.data();
    bd.order.resize(nt);
    std::iota(bd.order.begin(), bd.order.end(), 0);
    bd.cen.resize(3 * (size_t)nt);
    for (int i = 0; i < nt; ++i)
        for (int r = 0; r < 3; ++r) bd.cen[3 * (size_t)i + r] = (tri9[9 * (size_t)i + r] + tri9[9 * (size_t)i + 3 + r] + tri9[9 * (size_t)i + 6 + r]) / 3.0;
    bd.nodes.reserve(2 * (size_t)nt / LEAF_TRIS + 8);
    double rlo[3], rhi[3];
    if (nt > (1 << 27)) return cfs_fail(CFS_ERR_INVALID_ARG, "mesh too large (%d triangles)", nt);
    const int root = bd.build(0, nt, 1, rlo, rhi);
    if (root < 0) {                                          // a mesh that is one leaf: give it an inner root with an empty second child
        BvhNode nd;
        memset(&nd, 0, sizeof nd);
        for (int r = 0; r < 3; ++r) { nd.lo[0][r] = rlo[r]; nd.hi[0][r] = rhi[r]; nd.lo[1][r] = INFINITY; nd.hi[1][r] = -INFINITY; }
        nd.child[0] = root; nd.child[1] = -1;                // -1 = leaf code of zero triangles
        bd.nodes.push_back(nd);
    }
    if (bd.depth + 2 > MESH_STACK) return cfs_fail(CFS_ERR_INVALID_ARG, "mesh hierarchy too deep (%d levels)", bd.depth);
    std::vector<double> tri_o(9 * (size_t)nt);
    for (int i = 0; i < nt; ++i) memcpy(&tri_o[9 * (size_t)i], &tri9[9 * (size_t)bd.order[i]], 72);
    cfs_mesh *m = new (std::nothrow) cfs_mesh();
    if (!m) return cfs_fail(CFS_ERR_ALLOC, "out of host memory");
    m->device = cfs_current_device();
    m->nt = nt; m->nnodes = (int)bd.nodes.size(); m->depth = bd.depth;
    for (int r = 0; r < 3; ++r) { m->bbox[r] = rlo[r]; m->bbox[3 + r] = rhi[r]; }
    hipError_t e = hipSetDevice(m->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->nodes_d), bd.nodes.size() * sizeof(BvhNode));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->tri_d), tri_o.size() * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->orig_d), (size_t)nt * 4);
    if (e == hipSuccess) e = hipMemcpy(m->nodes_d, bd.nodes.data(), bd.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->tri_d, tri_o.data(), tri_o.size() * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(m->orig_d, bd.order.data(), (size_t)nt * 4, hipMemcpyHostToDevice);
    // the cut a cooperative traversal starts from: open the hierarchy level by level while the inner nodes fit a wavefront
    std::vector<int> cut{0}, cut_tri;
    while (!cut.empty() && cut.size() <= 32) {
        std::vector<int> next;
        for (int id : cut)
            for (int c = 0; c < 2; ++c) {
                const int ch = bd.nodes[id].child[c];
                if (ch >= 0) next.push_back(ch);
                else { const int code = -(ch + 1); for (int k = 0; k < (code & 7); ++k) cut_tri.push_back((code >> 3) + k); }
            }
        cut.swap(next);
    }
    m->ncut = (int)cut.size(); m->ncut_tri = (int)cut_tri.size();
    cut.insert(cut.end(), cut_tri.begin(), cut_tri.end());
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->cut_d), (cut.size() + 1) * 4);
    if (e == hipSuccess && !cut.empty()) e = hipMemcpy(m->cut_d, cut.data(), cut.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        const int rc = cfs_fail(cfs_device_count() > 0 ? CFS_ERR_HIP : CFS_ERR_NO_DEVICE, "mesh upload failed: %s", hipGetErrorString(e));
        cfs_mesh_destroy(m);
        return rc;
    }
    *out = m;
    return CFS_SUCCESS;
}

}  // namespace

template <int NJ>
static hipError_t launch_linearize_mesh_nj(const LinMeshParams &p, hipStream_t s)
{
    constexpr int NVT = nvt(NJ), NE = 2 * NJ + 1;
    const dim3 block(MESH_THREADS);
    auto blocks = [](size_t n) { return dim3((unsigned)((n + MESH_THREADS - 1) / MESH_THREADS)); };
    const size_t bh = (size_t)p.B * p.H;
    hipLaunchKernelGGL(mesh_fk_kernel<NJ>, blocks(bh * NE), block, 0, s, p);
    hipLaunchKernelGGL(mesh_upper_kernel<NJ>, blocks(bh * NJ * p.nmesh), block, 0, s, p);
    hipLaunchKernelGGL(mesh_base_kernel<NJ>, dim3((unsigned)((bh * NJ * p.nmesh * MESH_PIECES + CO_QPW - 1) / CO_QPW)), dim3(64), 0, s, p);
    hipLaunchKernelGGL(mesh_reduce_kernel<NJ>, blocks(bh * NJ * p.nmesh), block, 0, s, p);
    hipLaunchKernelGGL(mesh_shift_kernel<NJ>, blocks(bh * p.nmesh * (NVT - NJ)), block, 0, s, p);
    hipLaunchKernelGGL(mesh_fd_kernel<NJ>, blocks(bh * p.nmesh), block, 0, s, p);
    return hipGetLastError();
}

hipError_t launch_linearize_mesh(int nj, const LinMeshParams &p, hipStream_t s)
{
    switch (nj) {
    case 2: return launch_linearize_mesh_nj<2>(p, s);
    case 3: return launch_linearize_mesh_nj<3>(p, s);
    case 4: return launch_linearize_mesh_nj<4>(p, s);
    case 5: return launch_linearize_mesh_nj<5>(p, s);
    case 6: return launch_linearize_mesh_nj<6>(p, s);
    default: return hipErrorInvalidValue;
    }
}

// doubles / ints of workspace per (problem, waypoint) the pipeline needs
void linearize_mesh_workspace(int nj, int nmesh, size_t *ends, size_t *base, size_t *shift, size_t *near, size_t *piece_d, size_t *piece_i,
                              size_t *piece_nd)
{
    *ends = (size_t)nvt(nj) * 6;
    *base = (size_t)nj * nmesh;
    *shift = (size_t)nmesh * (nvt(nj) - nj);
    *near = (size_t)nj * nmesh * (NEAR_CAP + 1);
    *piece_d = (size_t)nj * nmesh * MESH_PIECES * PIECE_D;
    *piece_i = (size_t)nj * nmesh * MESH_PIECES * PIECE_I;
    *piece_nd = (size_t)nj * nmesh * MESH_PIECES * NEAR_CAP;
}

extern "C" {

int cfs_mesh_create(const double *vertices, int nv, const int *triangles, int nt, cfs_mesh **out)
{
    if (!out) return cfs_fail(CFS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!vertices || !triangles || nv < 3 || nt < 1) return cfs_fail(CFS_ERR_INVALID_ARG, "a mesh needs vertices and at least one triangle");
    if (cfs_device_count() <= 0) return cfs_fail(CFS_ERR_NO_DEVICE, "no HIP device");
    std::vector<double> tri9(9 * (size_t)nt);
    for (int i = 0; i < nt; ++i)
        for (int v = 0; v < 3; ++v) {
            const int id = triangles[3 * (size_t)i + v];
            if (id < 0 || id >= nv) return cfs_fail(CFS_ERR_INVALID_ARG, "triangle %d refers to vertex %d of %d", i, id, nv);
            for (int r = 0; r < 3; ++r) tri9[9 * (size_t)i + 3 * v + r] = vertices[3 * (size_t)id + r];
        }
    return upload_mesh(tri9, out);
}

// binary STL (80-byte header, uint32 count, 50-byte records); map_from_stl applies Lib/functions/MapFromSTL.m:6-10
// (shift every axis to start at 0, y -= 100, then (x, y, z) <- (z, x, y)) before the uniform `scale`
int cfs_mesh_load_stl(const char *path, double scale, int map_from_stl, cfs_mesh **out)
{
    if (!out) return cfs_fail(CFS_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!path) return cfs_fail(CFS_ERR_INVALID_ARG, "path is NULL");
    if (!(scale > 0)) return cfs_fail(CFS_ERR_INVALID_ARG, "scale must be positive");
    FILE *f = fopen(path, "rb");
    if (!f) return cfs_fail(CFS_ERR_INVALID_ARG, "cannot open %s", path);
    unsigned char head[84];
    if (fread(head, 1, 84, f) != 84) { fclose(f); return cfs_fail(CFS_ERR_INVALID_ARG, "%s: shorter than an STL header", path); }
    unsigned int nt;
    memcpy(&nt, head + 80, 4);
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    if (nt < 1 || sz != 84 + 50 * (long)nt) { fclose(f); return cfs_fail(CFS_ERR_INVALID_ARG, "%s: not a binary STL (%u triangles, %ld bytes)", path, nt, sz); }
    fseek(f, 84, SEEK_SET);
    std::vector<unsigned char> rec(50 * (size_t)nt);
    const size_t got = fread(rec.data(), 50, nt, f);
    fclose(f);
    if (got != nt) return cfs_fail(CFS_ERR_INVALID_ARG, "%s: truncated", path);
    std::vector<double> tri9(9 * (size_t)nt);
    for (size_t i = 0; i < nt; ++i) {
        float v[9];
        memcpy(v, rec.data() + 50 * i + 12, 36);
        for (int r = 0; r < 9; ++r) tri9[9 * i + r] = (double)v[r];
    }
    if (map_from_stl) {
        double mn[3] = {INFINITY, INFINITY, INFINITY};
        for (size_t i = 0; i < 3 * (size_t)nt; ++i)
            for (int r = 0; r < 3; ++r) mn[r] = std::min(mn[r], tri9[3 * i + r]);
        for (size_t i = 0; i < 3 * (size_t)nt; ++i) {
            const double x = tri9[3 * i] - mn[0], y = (tri9[3 * i + 1] - mn[1]) - 100.0, z = tri9[3 * i + 2] - mn[2];
            tri9[3 * i] = z; tri9[3 * i + 1] = x; tri9[3 * i + 2] = y;
        }
    }
    for (double &v : tri9) v *= scale;
    if (cfs_device_count() <= 0) return cfs_fail(CFS_ERR_NO_DEVICE, "no HIP device");
    return upload_mesh(tri9, out);
}

int cfs_mesh_info(const cfs_mesh *m, int *ntri, int *nnodes, int *depth, double *bbox6)
{
    if (!m) return cfs_fail(CFS_ERR_INVALID_ARG, "mesh is NULL");
    if (ntri) *ntri = m->nt;
    if (nnodes) *nnodes = m->nnodes;
    if (depth) *depth = m->depth;
    if (bbox6) memcpy(bbox6, m->bbox, sizeof m->bbox);
    return CFS_SUCCESS;
}

void cfs_mesh_destroy(cfs_mesh *m)
{
    if (!m) return;
    if (m->nodes_d) (void)hipFree(m->nodes_d);
    if (m->tri_d) (void)hipFree(m->tri_d);
    if (m->orig_d) (void)hipFree(m->orig_d);
    if (m->cut_d) (void)hipFree(m->cut_d);
    delete m;
}

// point2surface_dis for n segments (host arrays): dis[n], points[n x 6] (may be NULL), tri[n] (may be NULL)
int cfs_mesh_segment_distance(const cfs_mesh *m, int n, const double *segs, double *dis, double *points, int *tri)
{
    if (!m || !segs || !dis || n < 1) return cfs_fail(CFS_ERR_INVALID_ARG, "mesh/segs/dis must be given, n >= 1");
    CFS_HIPCHK(hipSetDevice(m->device));
    double *d_seg = nullptr, *d_dis = nullptr, *d_pts = nullptr;
    int *d_tri = nullptr;
    int rc = CFS_SUCCESS;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_seg), (size_t)n * 48);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_dis), (size_t)n * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_pts), (size_t)n * 48);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_tri), (size_t)n * 4);
    if (e == hipSuccess) e = hipMemcpy(d_seg, segs, (size_t)n * 48, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        SegQueryParams q{m->view(), n, d_seg, d_dis, d_pts, d_tri};
        hipLaunchKernelGGL(cfs_mesh_seg_kernel, dim3((n + MESH_THREADS - 1) / MESH_THREADS), dim3(MESH_THREADS), 0, nullptr, q);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(dis, d_dis, (size_t)n * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && points) e = hipMemcpy(points, d_pts, (size_t)n * 48, hipMemcpyDeviceToHost);
    if (e == hipSuccess && tri) e = hipMemcpy(tri, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = cfs_fail(CFS_ERR_HIP, "cfs_mesh_segment_distance: %s", hipGetErrorString(e));
    (void)hipFree(d_seg); (void)hipFree(d_dis); (void)hipFree(d_pts); (void)hipFree(d_tri);
    return rc;
}

// dist_arm_surf_200i for N poses (host arrays): d[N], linkid[N] (1-based, may be NULL), points[N x 6] (may be NULL)
int cfs_dist_arm_mesh(const cfs_robot *robot, int njoint, int N, const double *theta, const cfs_mesh *m,
                      double *d, int *linkid, double *points)
{
    if (!robot || !theta || !m || !d || N < 1) return cfs_fail(CFS_ERR_INVALID_ARG, "robot/theta/mesh/d must be given, N >= 1");
    int rc = cfs_check_robot(robot, njoint);
    if (rc) return rc;
    CFS_HIPCHK(hipSetDevice(m->device));
    DevRobot hr;
    cfs_build_dev_robot(*robot, hr);
    DevRobot *d_rb = nullptr;
    double *d_th = nullptr, *d_d = nullptr, *d_pts = nullptr;
    int *d_id = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_rb), sizeof(DevRobot));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_th), (size_t)N * njoint * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_d), (size_t)N * 8);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_pts), (size_t)N * 48);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_id), (size_t)N * 4);
    if (e == hipSuccess) e = hipMemcpy(d_rb, &hr, sizeof hr, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_th, theta, (size_t)N * njoint * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        ArmMeshParams q{d_rb, m->view(), N, njoint, d_th, d_d, d_id, d_pts};
        const int per = MESH_THREADS / njoint;
        hipLaunchKernelGGL(cfs_dist_arm_mesh_kernel, dim3((N + per - 1) / per), dim3(MESH_THREADS), 0, nullptr, q);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(d, d_d, (size_t)N * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && linkid) e = hipMemcpy(linkid, d_id, (size_t)N * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && points) e = hipMemcpy(points, d_pts, (size_t)N * 48, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = cfs_fail(CFS_ERR_HIP, "cfs_dist_arm_mesh: %s", hipGetErrorString(e));
    (void)hipFree(d_rb); (void)hipFree(d_th); (void)hipFree(d_d); (void)hipFree(d_pts); (void)hipFree(d_id);
    return rc;
}

}  // extern "C"
