/* mesh_oracle.c -- CPU oracle for mesh obstacles.  TEST INFRASTRUCTURE ONLY:
 * nothing in the product path may call into oracle/.
 *
 * The reference calls `[dis, points] = point2surface_dis(pos{i}.p, obs)` (M200i/dist_arm_surf_200i.m:21,
 * Lib/functions/dist_arm_surface.m:43) but does not contain that function: there is nothing to restate, so
 * the contract is the build's own (DESIGN.md "Mesh obstacles") -- PARITY UNPINNED:
 *   dis    = min over the triangles of the Euclidean distance between the link axis (a segment) and the
 *            triangle, 0 when they intersect;
 *   points = [closest point on the link axis ; closest point on the mesh];
 *   ties (equal dis) are broken towards the smaller parameter of the point along the link axis, so the
 *   result does not depend on the order in which triangles are visited.
 * This file is the brute-force statement of that contract (every triangle, no hierarchy); the HIP path
 * (BVH traversal) is tested against it.  The segment/segment part reuses orc_dist_lin_seg, the restatement
 * of Lib/functions/distLinSeg.m:23-91.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

double orc_dist_lin_seg(const double *p1s, const double *p1e, const double *p2s, const double *p2e, double *points);

#define ORC_MAX_MESH 16
typedef struct { int nt; double *tri; } orc_mesh;            /* tri: nt x 9 = A, B, C */
static orc_mesh g_mesh[ORC_MAX_MESH];

/* register triangle soup `tri` (nt x 9 doubles) under id 0..15; nt = 0 frees the slot */
int orc_mesh_register(int id, int nt, const double *tri)
{
    if (id < 0 || id >= ORC_MAX_MESH) return -1;
    free(g_mesh[id].tri);
    g_mesh[id].tri = 0;
    g_mesh[id].nt = 0;
    if (nt > 0) {
        g_mesh[id].tri = (double *)malloc(sizeof(double) * 9 * (size_t)nt);
        if (!g_mesh[id].tri) return -1;
        memcpy(g_mesh[id].tri, tri, sizeof(double) * 9 * (size_t)nt);
        g_mesh[id].nt = nt;
    }
    return 0;
}

static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void sub3(const double *a, const double *b, double *c) { c[0] = a[0] - b[0]; c[1] = a[1] - b[1]; c[2] = a[2] - b[2]; }
static void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* closest point of triangle ABC to P (Voronoi-region walk, Ericson, "Real-Time Collision Detection" 5.1.5) */
static void closest_pt_triangle(const double *P, const double *A, const double *B, const double *C, double *Q)
{
    double ab[3], ac[3], ap[3], bp[3], cp[3];
    sub3(B, A, ab); sub3(C, A, ac); sub3(P, A, ap);
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0 && d2 <= 0) { memcpy(Q, A, 24); return; }
    sub3(P, B, bp);
    const double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0 && d4 <= d3) { memcpy(Q, B, 24); return; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) {
        const double v = d1 / (d1 - d3);
        for (int r = 0; r < 3; ++r) Q[r] = A[r] + v * ab[r];
        return;
    }
    sub3(P, C, cp);
    const double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0 && d5 <= d6) { memcpy(Q, C, 24); return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) {
        const double w = d2 / (d2 - d6);
        for (int r = 0; r < 3; ++r) Q[r] = A[r] + w * ac[r];
        return;
    }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
        const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int r = 0; r < 3; ++r) Q[r] = B[r] + w * (C[r] - B[r]);
        return;
    }
    const double denom = 1.0 / (va + vb + vc);
    const double v = vb * denom, w = vc * denom;
    for (int r = 0; r < 3; ++r) Q[r] = A[r] + ab[r] * v + ac[r] * w;
}

/* candidate (dis, t, points) replaces the incumbent if dis is smaller, or equal with a smaller parameter t */
static void take(double dis, double t, const double *pl, const double *pm, double *bd, double *bt, double *pts)
{
    if (dis < *bd || (dis == *bd && t < *bt)) {
        *bd = dis; *bt = t;
        memcpy(pts, pl, 24); memcpy(pts + 3, pm, 24);
    }
}

/* parameter of point X along P0->P1 (0 for a zero-length segment) */
static double param_of(const double *P0, const double *d, double D, const double *X)
{
    if (D == 0) return 0;
    double e[3];
    sub3(X, P0, e);
    return dot3(e, d) / D;
}

/* distance between segment P0P1 and triangle ABC; updates the incumbent (bd, bt, pts) */
void orc_seg_tri_update(const double *P0, const double *P1, const double *A, const double *B, const double *C,
                        double *bd, double *bt, double *pts)
{
    double d[3], ab[3], ac[3], n[3], e0[3], e1[3];
    sub3(P1, P0, d);
    const double D = dot3(d, d);
    sub3(B, A, ab); sub3(C, A, ac);
    cross3(ab, ac, n);
    const double nn = dot3(n, n);
    if (nn > 0) {                                            /* proper triangle: does the segment pierce it? */
        sub3(P0, A, e0); sub3(P1, A, e1);
        const double s0 = dot3(n, e0), s1 = dot3(n, e1);
        if (s0 * s1 <= 0 && s0 != s1) {
            const double t = s0 / (s0 - s1);
            double X[3], xa[3], xb[3], xc[3], bc[3], ca[3], c0[3], c1[3], c2[3];
            for (int r = 0; r < 3; ++r) X[r] = P0[r] + t * d[r];
            sub3(X, A, xa); sub3(X, B, xb); sub3(X, C, xc);
            sub3(C, B, bc); sub3(A, C, ca);
            cross3(ab, xa, c0); cross3(bc, xb, c1); cross3(ca, xc, c2);
            if (dot3(n, c0) >= 0 && dot3(n, c1) >= 0 && dot3(n, c2) >= 0) {
                take(0.0, t, X, X, bd, bt, pts);
                return;
            }
        }
    }
    /* the two end points against the triangle */
    const double *ends[2] = {P0, P1};
    for (int k = 0; k < 2; ++k) {
        double Q[3], e[3];
        closest_pt_triangle(ends[k], A, B, C, Q);
        sub3(ends[k], Q, e);
        take(sqrt(dot3(e, e)), (k == 0 || D == 0) ? 0.0 : 1.0, ends[k], Q, bd, bt, pts);
    }
    /* the segment against the three edges (distLinSeg) */
    const double *ea[3] = {A, B, C}, *eb[3] = {B, C, A};
    for (int k = 0; k < 3; ++k) {
        double p6[6];
        const double dis = orc_dist_lin_seg(P0, P1, ea[k], eb[k], p6);
        take(dis, param_of(P0, d, D, p6), p6, p6 + 3, bd, bt, pts);
    }
}

/* point2surface_dis: seg = [p(:,1); p(:,2)] (6), returns dis; points[6]; *tri_id = index of the winning triangle */
double orc_mesh_seg_distance(int mesh_id, const double *seg, double *points, int *tri_id)
{
    const orc_mesh *m = &g_mesh[mesh_id];
    double bd = INFINITY, bt = INFINITY, pts[6] = {0, 0, 0, 0, 0, 0};
    int best = -1;
    for (int k = 0; k < m->nt; ++k) {
        const double *T = m->tri + 9 * (size_t)k;
        const double od = bd, ot = bt;
        orc_seg_tri_update(seg, seg + 3, T, T + 3, T + 6, &bd, &bt, pts);
        if (bd != od || bt != ot) best = k;
    }
    if (points) memcpy(points, pts, sizeof pts);
    if (tri_id) *tri_id = best;
    return bd;
}

void orc_mesh_seg_distance_batch(int mesh_id, int n, const double *segs, double *dis, double *points, int *tri_id)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n; ++i)
        dis[i] = orc_mesh_seg_distance(mesh_id, segs + 6 * (size_t)i, points ? points + 6 * (size_t)i : 0, tri_id ? tri_id + i : 0);
}
