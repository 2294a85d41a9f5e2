// cfs_device.h -- shared declarations of libcfs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cfs_hip.h"

#define CFS_WAVE 64

// Robot constants in the form the kernels consume (built on the host from cfs_robot).
// cos/sin of the constant DH alpha are evaluated once on the host (CapPos.m:13-15 evaluates
// them on every call; they are call-invariant).
struct DevRobot {
    int kind;                    // cfs_robot_kind
    int nlink;
    double dh_d[CFS_MAX_LINKS];  // DH(:,2)
    double dh_a[CFS_MAX_LINKS];  // DH(:,3)
    double ca[CFS_MAX_LINKS];    // cos(DH(:,4))
    double sa[CFS_MAX_LINKS];    // sin(DH(:,4))
    double th_off[CFS_MAX_LINKS];// subtracted from theta: pi/2 on joint 2 for M200i (dist_arm_3D_200i_2.m:11)
    double base[3];
    double cap[CFS_MAX_LINKS * 6];  // cap[i*6 + k*3 + r]
    double t2l[CFS_MAX_LINKS * 3];  // 2L: translation of link i = robot.T(:,i+1)  (CapPos2.m:25)
    double shift_bound;             // no point of the arm moves farther than this between the base pose and an evaluation point of
                                    // num_jac: nlink * eps/2 * reach
    double prune_tol;               // a link farther than this from the base-pose minimum cannot become the minimum at any
                                    // evaluation point of num_jac (bound on the motion of any arm point: nlink*eps/2*reach)
};
static_assert(sizeof(DevRobot) % 8 == 0, "DevRobot is copied to LDS as doubles");

// Structure of QQ = Baug'*Qaug*Baug + cR*(R+R') when the handle was created from the cost weights (cfs_problem_create_from_weights):
// QQ*u = Bpos'(w_i (Qp p_i + qc v_i)) + Bvel'(w_i (qc p_i + Qv v_i)) + Rs u_i with (p, v) = (Bpos u, Bvel u), w_i = w_stage | w_terminal
struct DevCost {
    double Qp[36], Qv[36], Rs[36];   // nj x nj column-major (leading dimension nj): Qp, Qv, cR*(Rblk + Rblk')
    double qc, ws, wt, pad;
};
static_assert(sizeof(DevCost) % 8 == 0, "DevCost is read as doubles");

// ---- plain dist_arm and the dense constraint writer (cfs_geom.hip) ----------------------------
struct DistArmParams {
    const DevRobot *rb;
    int N, nobs, nj;
    const double *theta;         // N x nj
    const double *obs;           // nobs x 6
    double *d;                   // N x nobs
    int *linkid;                 // N x nobs
    double *pos;                 // N x nj x 6 or null
};
void launch_dist_arm(const DistArmParams &p, hipStream_t s);

// launch order of the fused solver (cfs_geom.hip): problems whose initial trajectory violates the most (waypoint, obstacle)
// clearances first -- those run the longest active sets (rank correlation 0.7 with the QP steps on config 3)
struct OrderParams {
    const DevRobot *rb;
    int B, H, nj, nobs;          // nobs: line obstacles only
    int obs_stride;              // obstacles per problem in `obs` (line + mesh)
    const double *x_init;        // B x H x 2nj
    const double *obs;           // B x obs_stride x 6
    const double *margin;        // obs_stride
    int *key;                    // B
    int *order;                  // B
};
void launch_order(const OrderParams &p, hipStream_t s);

struct DenseConParams {
    int B, H, nj, nobs;
    double dt;
    const double *dist, *grad;   // B x nobs x H, B x nobs x H x nj (the fused kernel's linearisation dump)
    const double *u;             // B x nn
    const double *xR1;           // B x 2nj
    const double *lim, *margin;
    double *Ainq;                // B x rows x nn (column-major per problem)
    double *binq;                // B x rows
};
void launch_dense_con(const DenseConParams &p, hipStream_t s);

enum { QP_OK = 0, QP_INFEASIBLE = 2, QP_NUMERIC = 3 };   // outcome of one QP inside the fused kernel

// ---- K3: batched dense products on the matrix cores (fp64 MFMA) --------------------------------
struct GemvParams {
    int B, nn;
    const double *M;             // nn x nn column-major
    const double *X;             // B x nn
    double *Y;                   // B x nn : Y[b] = scale * M * X[b]
    double scale;
};
void launch_batched_gemv(const GemvParams &p, hipStream_t s);

// ---- fused persistent solver: one workgroup owns one problem for its whole outer loop -----------
struct FusedParams {
    const DevRobot *rb;
    int B, H, nobs, mode, has_bounds, max_o_iter, noise_rows;
    int qy, lin_w;               // filled by launch_fused: Y rows held in LDS, waypoints per linearisation tile
    double dt, alpha, epsilon_O;
    double lmax_vel;             // rigorous upper bound of lambda_max(D'HD/dt^2): curvature of the QP objective in velocity space
    double polish_tol;           // relative drift |slack| / (1 + |bound|) of an active row at the optimum that triggers the projection
    double lmax_H;               // rigorous upper bound of lambda_max(H), H = QQ symmetrised (CFS) | I (PSGCFS projection)
    const double *M1, *M2, *M3;  // nn x nn column-major: H^{-1}Bpos', H^{-1}Bvel', H^{-1} (natural row order)
    const double *M1v, *M1p, *M2v, *M2p, *M3v, *M3p;   // Bvel* and Bpos* of every column of M1, M2, M3
    const double *QQ;            // raw sys_info.QQ
    const DevCost *cost;         // structure of QQ (null: dense QQ only)
    const double *lim, *maxin, *margin;
    const double *x_init, *xR1, *ff, *caug, *obs, *noise;
    const double *x0;            // CFS: -H^{-1} ff  (B x nn)
    double *u, *x_, *cost_all, *e_cost_all, *e_u_all;
    int *iter_O, *total_iter, *status;
    double *Yg;                  // pool_n x nn x nn: Y rows beyond the LDS capacity (one slot per workgroup that spills)
    double *Pt;                  // pool_n x pt_stride: columns of the inverse Gram matrix beyond the register-resident ones
    size_t pt_stride;
    int *pool_flag;              // pool_n: 0 free / 1 taken (atomicCAS by the workgroup that needs a slot)
    int pool_n;
    double *dbg;                 // optional trace: 8 doubles per active-set step of problem dbg_b (developer aid)
    int dbg_b, dbg_cap;
    const int *order;            // optional launch order: workgroup w solves problem order[w] (a permutation of 0..B-1); NULL = identity
    unsigned long long *stamps;  // optional: 16 cycle accumulators per problem (developer aid)
    int opt;                     // developer A/B switches (bit 0: gather w only and roll it on the fly)
    double *u_hist;              // CFS: B x max_o_iter x nn log of u per outer iteration (cost history computed afterwards)
    double *u_log;               // test aid (cfs_debug_log_u): the same log for either solver, no effect on the solve; may be null
    // mesh obstacles: the last nmesh of the nobs obstacles; their rows come from cfs_linearize_mesh_kernel, which needs the
    // current iterate, so the host drives such solves one outer iteration per launch and the state travels through HBM
    int nmesh;
    const double *ext_dist;      // B x nmesh x H
    const double *ext_grad;      // B x nmesh x H x NJ
    int resume;                  // 0: constructor state (CFS_FANUC.m:55-58), 1: continue from u / x_ / iter_O / total_iter / status and st_*
    int max_launch_iters;        // outer iterations per launch (0: run to the end)
    double *st_qu;               // B x nn      QQ*u
    double *st_cost;             // B x 2       cost_new, cost_old
    int *st_noise, *st_done;     // B           noise rows consumed; 1 when the problem has finished
    // pieces of the path through the SAME kernel (cfs_linearize / cfs_get_con / cfs_qp, kernel-level parity tests):
    int warm_max;                // developer knob: largest previous active set (slots) the warm start takes (0: the register-resident columns)
    int piece;                   // 0: whole solve; 1: linearise x_init and return (dump_*); 2: one QP on the given linearisation
                                 //    (ext_dist / ext_grad with nmesh = nobs, u = linearisation point in, solution out, x0 = start)
    int no_prune;                // test switch: evaluate every link at every evaluation point of num_jac (no candidate pruning)
    double *dump_dist;           // B x nobs x H       distances of the first linearisation of this launch (may be null)
    double *dump_grad;           // B x nobs x H x NJ  Diff of the same
    int *dump_linkid;            // B x nobs x H       closest link (1-based, line obstacles only)
    double *dump_lambda;         // B x (nobs*H + 4nn) multipliers of the last QP [collision (j,i) | vel+ | vel- | bound+ | bound-] (may be null)
};
// cfs_fused.hip is compiled into three tiers (workgroups per CU / register-resident columns of the inverse Gram matrix):
//   w1  1 / 64  whole CU per problem: longest on-chip active sets
//   w2m 2 / 32  two problems per CU hide each other's latencies; medium active sets stay in registers
//   w2s 2 / 16  same, smallest register footprint (no spills): the projection QPs of PSGCFS have 2-3 active rows
hipError_t launch_fused_w1(int nj, FusedParams p, hipStream_t s);
hipError_t launch_fused_w2m(int nj, FusedParams p, hipStream_t s);
hipError_t launch_fused_w2s(int nj, FusedParams p, hipStream_t s);
bool fused_fits_w1(int nj, int H, int nobs);
bool fused_fits_w2m(int nj, int H, int nobs);
bool fused_fits_w2s(int nj, int H, int nobs);
hipError_t launch_fused(int nj, FusedParams p, hipStream_t s, bool force_w1 = false);   // cfs_api.hip: tier by mode and capacity
bool fused_fits(int nj, int H, int nobs);

struct CostHistParams {          // EVAL.get_cost / store_result for a logged u history (CFS mode)
    int B, nn, max_o_iter;
    const double *u_hist, *qu_hist;   // B x max_o_iter x nn: u and QQ*u
    const double *ff, *caug;
    const int *iter_O;
    double *cost_all, *e_cost_all;
};
void launch_cost_history(const CostHistParams &p, hipStream_t s);

struct TermsParams {             // per-problem line reference and cost terms from (x0, xg)
    int B, H, nj;
    const double *F1, *F2;       // nn x nj column-major: Baug'*Qaug*Aaug(:,1:nj), Baug'*Qaug*G
    const double *Cq;            // 2nj x 2nj column-major: E'*Qaug*E, E = [Aaug(:,1:nj), -G]
    const double *x0, *xg;       // B x nj
    const double *route;         // optional, B x nwp x nj (waypoint-major): x0 / xg are its ends, x_init its cubic resampling
    int nwp;
    const int *nwp_b;            // optional, B: rows of route b actually used (ragged routes, stride nwp_stride rows per problem)
    int nwp_stride;
    double dt;
    double *x_init, *xR1, *ff, *caug;
};
void launch_build_terms(const TermsParams &p, hipStream_t s);
