// cfs_host.h -- host-side helpers shared by the translation units of libcfs_hip.so (not part of the C ABI).
#pragma once
#include "cfs_device.h"
#include <vector>

int cfs_fail(int code, const char *fmt, ...);            // records the message of cfs_last_error(), returns code
int cfs_current_device();                                // device chosen with cfs_set_device
int cfs_check_robot(const cfs_robot *r, int nj);         // CFS_SUCCESS or an error code (message recorded)
void cfs_build_dev_robot(const cfs_robot &r, DevRobot &d);

#define CFS_HIPCHK(call)                                                                                     \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return cfs_fail(CFS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));   \
    } while (0)

// ---- mesh obstacles (cfs_mesh.hip) --------------------------------------------------------------
struct BvhNode {                 // 128 B = one L2 line: an inner node carries the boxes of BOTH children, so a level costs one
    double lo[2][3], hi[2][3];   // dependent load; an empty child has lo = +inf, hi = -inf (its lower bound is +inf)
    int child[2];                // >= 0: inner node index; < 0: leaf, -(first * 8 + count) - 1, triangles [first, first + count) in BVH order
    int pad[2];
};
struct DevMesh {                 // device view of one mesh
    const BvhNode *nodes;
    const double *tri;           // nt x 9 (A, B, C), in BVH order
    const int *orig;             // BVH order -> index in the caller's triangle list
    int nnodes, nt;
    // where a wave-cooperative traversal starts: the hierarchy cut open to <= 64 inner nodes (a wavefront's width), plus the
    // triangles of the leaves met on the way -- the 1, 2, 4, ... lane rounds near the root are one dependent load each
    const int *cut, *cut_tri;
    int ncut, ncut_tri;
};
struct cfs_mesh {
    int device = 0, nt = 0, nnodes = 0, depth = 0;
    double bbox[6] = {0, 0, 0, 0, 0, 0};
    BvhNode *nodes_d = nullptr;
    double *tri_d = nullptr;
    int *orig_d = nullptr;
    int *cut_d = nullptr;        // ncut node indices, then ncut_tri triangle indices
    int ncut = 0, ncut_tri = 0;
    DevMesh view() const { return DevMesh{nodes_d, tri_d, orig_d, nnodes, nt, cut_d, cut_d + ncut, ncut, ncut_tri}; }
};

struct LinMeshParams {           // distance + literal finite-difference Jacobian against mesh obstacles
    const DevRobot *rb;
    int B, H, nmesh;
    const DevMesh *meshes;       // device array [nmesh]
    const double *x_;            // B x (H*2*NJ)
    const int *status_done;      // B, may be null: problems whose entry is non-zero have finished and are skipped
    int seed_prev;               // base_t holds the winning triangles of the previous outer iteration of the same problems: their
                                 // distance to the moved link is one more upper bound (the trajectory moves little between iterations)
    double *dist;                // B x nmesh x H
    double *grad;                // B x nmesh x H x NJ
    // workspace per (problem, waypoint), sizes from linearize_mesh_workspace
    double *ends;                // [NVT][6]          end points of every link variant
    double *upper_d;             // [NJ][nmesh]       greedy upper bound of the base-pose distance
    double *base_d;              // [NJ][nmesh]       base-pose distance (surrogate applied; +inf: farther than any link that matters)
    int *base_t;                 // [NJ][nmesh]       base-pose winning triangle (hierarchy order)
    double *shift_d;             // [nmesh][NVT-NJ]   shifted-pose distances of the candidate links, +inf otherwise
    int *near;                   // [NJ][nmesh][1+cap] count (-1: overflow) + triangles within the shift margin of the base minimum
    double *piece_d, *piece_nd;  // per piece of a link axis: record / near-list distances (sizes from linearize_mesh_workspace)
    int *piece_i;
};
hipError_t launch_linearize_mesh(int nj, const LinMeshParams &p, hipStream_t s);
void linearize_mesh_workspace(int nj, int nmesh, size_t *ends, size_t *base, size_t *shift, size_t *near, size_t *piece_d, size_t *piece_i,
                              size_t *piece_nd);

// ---- CHOMP_FANUC (cfs_chomp.hip) -----------------------------------------------------------------------
struct ChompParams {
    const DevRobot *rb;
    int B, H, nobs, max_o_iter;
    double dt, alpha, epsilon_O;
    const double *QQ;                                    // nn x nn column-major
    const double *x_init, *xR1, *ff, *caug, *obs, *u0;   // per problem: H*ns, ns, nn, 1, nobs*6, nn
    const double *D, *eps;                               // nobs: obs{j}.D, obs{j}.epsilon
    double *u, *x_, *cost_all, *e_cost_all, *e_u_all;
    int *iter_O, *total_iter, *status;
    double delta[26], fdarule[2], rmat[12], pinv[12], cov_scale;   // derivest's constant tables (chomp_derivest_tables)
};
hipError_t launch_chomp(int nj, const ChompParams &p, hipStream_t s);
void chomp_derivest_tables(ChompParams &p);
bool chomp_fits(int nj, int H, int nobs);

// ---- RRT / RRT* tree growth (cfs_rrt.hip) ---------------------------------------------------------------------
struct RrtParams {
    DevRobot rb;                                         // by value: no device allocation, no synchronisation in the _device entry
    int S, nobs, solver, max_iter, per_tree;             // trees; obstacles; 0 RRT | 1 RRT*; MAX_ITER; 1: x0 / goal / goal_th are S x nstate
    double bi, rewire;                                   // goal bias threshold (0.5, RRT_FANUC.m:38), re-parenting radius (0.2, :135)
    const double *x0, *goal, *goal_th;                   // nstate (or S x nstate)
    const double *region_g, *region_s, *sample_off, *ratial;   // nstate
    const double *obs, *D;                               // nobs x 6, nobs
    const double *uniforms;                              // S x ndraw or null (then the counter-based generator with `seed`)
    int ndraw;
    unsigned long long seed;
    long long max_draws;                                 // generator mode: uniforms a tree may consume before it gives up (fail = 2)
    int *node_num, *fail, *parent, *route_len;           // S, S, S x (max_iter+1), S
    double *nodes, *total_dis, *all_ee, *route;          // S x (max_iter+1) x nstate, S x (max_iter+1), S x max_iter x 3 (may be null), S x (max_iter+1) x nstate
    long long *draws_used, *proposals;                   // S (may be null)
};
hipError_t launch_rrt(int nj, const RrtParams &p, hipStream_t s);
size_t rrt_lds_bytes(int nj, int max_iter);
