/*
 * cfs_hip.h -- C ABI of libcfs_hip.so: the MI355X (gfx950) Convex-Feasible-Set inner loop.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (JessicaLeu-code/MotionPlanning_5D_m) is MATLAB with no FFI of its own; the entry points
 * below are what a MEX gateway for that path binds (see INTEGRATION.md and matlab/cfs_mex.cpp).
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference root).
 *
 * Conventions
 *   - all floating point is IEEE fp64; all matrices are COLUMN-MAJOR exactly as MATLAB hands
 *     them (mxGetPr); a leading batch dimension B, where present, is the slowest one;
 *   - plain pointers and sizes only; the caller owns every buffer; the library owns only the
 *     opaque cfs_problem handle (device copies of the problem-family constants and workspace);
 *   - functions return CFS_SUCCESS (0) or a negative cfs_error; they never abort; the text of
 *     the last error of the calling thread is available from cfs_last_error();
 *   - entry points with the suffix _device take DEVICE pointers and a hipStream_t (passed as
 *     void*); they enqueue work and return without synchronising; the others take HOST
 *     pointers, copy in/out and synchronise;
 *   - there is no CPU fallback anywhere: without a HIP device every compute entry point
 *     returns CFS_ERR_NO_DEVICE.
 */
#ifndef CFS_HIP_H
#define CFS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CFS_ABI_VERSION 1
#define CFS_MAX_LINKS 8   /* rows of robot.DH / entries of robot.cap the library accepts */
#define CFS_MAX_OBS 32    /* obstacles per problem                                       */
#define CFS_MAX_H 64      /* horizon (waypoints), one wavefront lane per waypoint         */

/* error codes (return values) */
typedef enum cfs_error {
    CFS_SUCCESS = 0,
    CFS_ERR_INVALID_ARG = -1,
    CFS_ERR_NO_DEVICE = -2,
    CFS_ERR_HIP = -3,          /* a HIP runtime call failed, see cfs_last_error()           */
    CFS_ERR_NOT_SPD = -4,      /* QQ (symmetrised) is not positive definite                  */
    CFS_ERR_DYNAMICS = -5,     /* sys_info.Aaug/Baug are not the double integrator of robot.A/B */
    CFS_ERR_ALLOC = -6
} cfs_error;

/* per-problem status written by the solvers; the reference has no error convention
 * (quadprog's exitflag is ignored, Lib/CFS_FANUC.m:85) -- an infeasible QP there crashes at
 * Lib/CFS_FANUC.m:92; here it is reported. */
typedef enum cfs_status {
    CFS_OK_CONVERGED = 0,  /* "Converged at stepN"  (Lib/EVAL.m:65-67) */
    CFS_OK_MAXITER = 1,    /* "MAX_ITER"            (Lib/EVAL.m:69-72) */
    CFS_QP_INFEASIBLE = 2, /* the linearised constraints of some outer iteration are infeasible */
    CFS_NUMERIC = 3        /* the active-set solver gave up (iteration cap / breakdown)      */
} cfs_status;

/* which dist_arm_* the class constructor selects (Lib/CFS_FANUC.m:49-54) */
typedef enum cfs_robot_kind {
    CFS_ROBOT_M16IB = 0, /* Lib/M16iB/dist_arm_3D_Heu_2.m : DH chain (Lib/functions/CapPos.m)            */
    CFS_ROBOT_M200I = 1, /* Lib/200i/dist_arm_3D_200i_2.m : DH chain with theta(2) - pi/2 (:11)           */
    CFS_ROBOT_2L = 2     /* Lib/2L/dist_arm_2L.m + Lib/2L/CapPos2.m : planar Rz chain with robot.T        */
} cfs_robot_kind;

typedef enum cfs_mode {
    CFS_MODE_CFS = 0,    /* Lib/CFS_FANUC.m    : QP with QQ, ff, bounds +-MAX_input, margin obs{j}.epsilon */
    CFS_MODE_PSGCFS = 1  /* Lib/PSGCFS_FANUC.m : noisy gradient step + projection QP, margin obs{j}.D      */
} cfs_mode;

/* robot = the fields of robotproperty2(id) (Lib/functions/robotproperty2.m:1-153) the path reads */
typedef struct cfs_robot {
    int kind;                       /* cfs_robot_kind                                           */
    int nlink;                      /* size(robot.DH,1)                                          */
    double DH[CFS_MAX_LINKS * 4];   /* robot.DH, nlink x 4 COLUMN-MAJOR: DH[i + c*nlink]         */
    double base[3];                 /* robot.base                                                */
    double cap[CFS_MAX_LINKS * 6];  /* robot.cap{i+1}.p, 3x2 column-major each: cap[i*6 + k*3+r] */
    double T[9];                    /* 2L only: robot.T, 3x3 column-major                        */
    double delta_t;                 /* robot.delta_t                                             */
} cfs_robot;

/* problem family = everything in sys_info that does not change across the batch
 * (main_FANUC.m:106-127): consumed by cfs_problem_create. */
typedef struct cfs_problem_desc {
    cfs_robot robot;        /* sys_info.robot                                                  */
    int mode;               /* cfs_mode                                                        */
    int H;                  /* sys_info.H       (<= CFS_MAX_H)                                 */
    int njoint;             /* sys_info.njoint  (= sys_info.nu; nstate = 2*njoint)             */
    int nobs;               /* size(obs,2)      (<= CFS_MAX_OBS)                               */
    const double *QQ;       /* sys_info.QQ, nn x nn, nn = H*njoint                             */
    const double *Aaug;     /* sys_info.Aaug, (H*nstate) x nstate; may be NULL (then implied)  */
    const double *Baug;     /* sys_info.Baug, (H*nstate) x nn;     may be NULL (then implied)  */
    const double *lim;      /* sys_info.lim, njoint                                            */
    const double *MAX_input;/* sys_info.MAX_input, nn (CFS mode; ignored for PSGCFS)           */
    const double *margin;   /* nobs: obs{j}.epsilon (CFS, CFS_FANUC.m:117) / obs{j}.D (PSGCFS_FANUC.m:158) */
    double epsilon_O;       /* sys_info.epsilon_O                                              */
    int MAX_O_ITER;         /* sys_info.MAX_O_ITER                                             */
    double alpha;           /* sys_info.alpha (PSGCFS step, main_FANUC.m:120); CFS mode does not read it */
    int max_batch;          /* capacity B_max of the handle's device workspace                 */
} cfs_problem_desc;

typedef struct cfs_problem cfs_problem; /* opaque */

/* per-batch inputs: what differs between the B problems (start/goal/obstacles/seeds) */
typedef struct cfs_batch_in {
    int B;
    const double *x_init; /* B x (H*nstate): sys_info.x_  (stacked [theta;omega] of waypoints 1..H) */
    const double *xR1;    /* B x nstate    : sys_info.xR(:,1)                                       */
    const double *ff;     /* B x nn        : sys_info.ff                                            */
    const double *caug;   /* B             : sys_info.caug                                          */
    const double *obs;    /* B x nobs x 6  : [obs{j}.l(:,1); obs{j}.l(:,2)]                         */
    const double *noise;  /* PSGCFS: B x noise_rows x nn draws of normrnd(0,0.1) (PSGCFS_FANUC.m:109),
                             one row consumed per PSG step; NULL = zeros                            */
    int noise_rows;
} cfs_batch_in;

/* per-batch outputs: what the callers read back (main_FANUC.m:144-162, RRTstar_CFS.m:197-203) */
typedef struct cfs_batch_out {
    double *u;           /* B x nn          : self.u                                   */
    double *x_;          /* B x (H*nstate)  : self.x_                                  */
    double *cost_all;    /* B x MAX_O_ITER  : self.eval.cost_all   (first iter_O-1 entries valid) */
    double *e_cost_all;  /* B x MAX_O_ITER  : self.eval.e_cost_all                     */
    double *e_u_all;     /* B x MAX_O_ITER  : self.eval.e_u_all                        */
    int *iter_O;         /* B : self.iter_O (reference convention: iterations run = iter_O-1) */
    int *total_iter;     /* B : self.total_iter (sum of active-set steps; stands in for quadprog's output.iterations) */
    int *status;         /* B : cfs_status                                             */
} cfs_batch_out;

/* ---- library ------------------------------------------------------------------------------ */
int cfs_abi_version(void);
const char *cfs_last_error(void);
int cfs_device_count(void);                 /* number of HIP devices (0 without a GPU)  */
int cfs_set_device(int device);             /* device used by subsequently created handles */

/* ---- problem family handle ------------------------------------------------------------------
 * replaces: the constructors CFS_FANUC(obs,sys_info,ROBOT) (Lib/CFS_FANUC.m:40-59) and
 * PSGCFS_FANUC(obs,sys_info,ROBOT) (Lib/PSGCFS_FANUC.m:43-62) plus the once-per-solve setup
 * quadprog does internally (factorising QQ).  Host pointers.  Validates Aaug/Baug against
 * the double integrator of robot.A/robot.B (robotproperty2.m:136-139) when they are given. */
int cfs_problem_create(const cfs_problem_desc *desc, cfs_problem **out);
void cfs_problem_destroy(cfs_problem *p);

/* The same from the WEIGHTS of the drivers' cost instead of the assembled matrices (row f2 of the scope table): the library
 * builds Aaug/Baug (double integrator of robot.A/robot.B), Q = [Qp q_cross*I; q_cross*I Qv], Qaug = blkdiag(w_stage*Q, ...,
 * w_terminal*Q), R = kron(I_H, Rblk), R = R + R', QQ = Baug'*Qaug*Baug + cR*R exactly as main_FANUC.m:64-97 (RRTstar_CFS.m:
 * 124-157, main_2L.m:69-93) do, the state-cost terms cfs_set_state_cost would otherwise be given (so cfs_build_terms_device
 * works at once), and, when desc->alpha == 0, alpha = 1/max(svd(QQ)) (main_FANUC.m:120).  desc->QQ / Aaug / Baug are ignored.
 * Neither QQ nor Qaug crosses the boundary; cfs_problem_family reads back what was built.  Because the structure of QQ is
 * known, QQ*u inside the solver (get_cost, Lib/EVAL.m:51-53; dcostArm_f, Lib/PSGCFS_FANUC.m:131-133) is evaluated through
 * Baug and the 2nj x 2nj blocks instead of the dense nn x nn matrix. */
typedef struct cfs_cost_weights {
    const double *Qp;   /* njoint x njoint column-major: Q(1:nj,1:nj)           (main_FANUC.m:66-70)   */
    const double *Qv;   /* njoint x njoint:              Q(nj+1:2nj,nj+1:2nj)   (main_FANUC.m:73-77)   */
    double q_cross;     /* Q(1:nj,nj+1:2nj) = Q(nj+1:2nj,1:nj) = q_cross*eye    (0.1, main_FANUC.m:71-72) */
    double w_stage;     /* Qaug block of waypoints 1..H-1 = Q*w_stage           (0.1, main_FANUC.m:81) */
    double w_terminal;  /* Qaug block of waypoint H      = Q*w_terminal         (10000, main_FANUC.m:83) */
    const double *Rblk; /* njoint x njoint column-major                         (main_FANUC.m:90-94)   */
    double cR;          /* QQ = Baug'*Qaug*Baug + R.*cR                         (50 | 10 | 0.1, :97)   */
} cfs_cost_weights;
int cfs_problem_create_from_weights(const cfs_problem_desc *desc, const cfs_cost_weights *w, cfs_problem **out);
/* what the handle was built with: QQ (nn x nn column-major, HOST pointer, may be NULL) and alpha (may be NULL) */
int cfs_problem_family(const cfs_problem *p, double *QQ, double *alpha);

/* ---- whole solve -----------------------------------------------------------------------------
 * replaces: self.optimizer() (Lib/CFS_FANUC.m:62-79, Lib/PSGCFS_FANUC.m:65-82) for B problems.
 * Host pointers; copies in, runs all outer iterations on the device, copies out, synchronises. */
int cfs_solve_batch(cfs_problem *p, const cfs_batch_in *in, const cfs_batch_out *out);

/* Same with DEVICE pointers on `stream` (hipStream_t as void*), no synchronisation.
 * in->B <= max_batch.  This is the entry bench.py times (inputs resident in HBM). */
int cfs_solve_batch_device(cfs_problem *p, const cfs_batch_in *in, const cfs_batch_out *out, void *stream);

/* Launch order of the fused solver: workgroup w solves problem order[w].  The added batch dimension has no counterpart in
 * the reference; a launch lasts as long as its longest problem plus the time that problem waited for a free compute unit.
 *   order = NULL, n = 0  automatic (default): problems whose initial trajectory violates the most (waypoint, obstacle)
 *                        clearances first, counted by a pre-pass on the solve's stream when B exceeds the device's compute units
 *   order = NULL, n < 0  identity (blockIdx order)
 *   order != NULL        HOST pointer, a permutation of 0..n-1, used by the next solves with B = n (a replanning loop may pass
 *                        the previous solve's total_iter, sorted); synchronises the device
 * Results do not depend on the order. */
int cfs_set_launch_order(cfs_problem *p, const int *order, int n);

/* ---- per-problem setup on the device (row f2 of the scope table) ---------------------------------------
 * replaces, for B (start, goal) pairs at once: the straight-line reference of main_FANUC.m:38-49
 * (x_ = joint-space line, zero velocities, waypoint 0 dropped), xR(:,1) = [x0; 0] and the cost terms
 * ff = ((Aaug*xR(:,1)-gaug)'*Qaug*Baug)' and caug = (Aaug*xR(:,1)-gaug)'*Qaug*(Aaug*xR(:,1)-gaug) with
 * gaug = kron(ones(H,1),[xg;0])  (main_FANUC.m:98-103).
 * cfs_set_state_cost: Qaug = the drivers' state-cost matrix (H*nstate x H*nstate, column-major, HOST pointer;
 * main_FANUC.m:79-84), given once per handle.  cfs_build_terms_device: x0, xg: B x njoint (DEVICE);
 * outputs (DEVICE): x_init B x H*nstate, xR1 B x nstate, ff B x nn, caug B; enqueued on `stream`. */
int cfs_set_state_cost(cfs_problem *p, const double *Qaug);
int cfs_build_terms_device(cfs_problem *p, int B, const double *x0, const double *xg,
                           double *x_init, double *xR1, double *ff, double *caug, void *stream);
/* The same from B RRT routes (RRTstar_CFS.m:94-110): routes is B x nwp x njoint (DEVICE; per problem the 5 x nwp route_wp as
 * MATLAB stores it); x_init = cubicpolytraj(route, (0:nwp-1)*delta_t, linspace(0,(nwp-1)*delta_t,H+1)) with zero waypoint
 * velocities (its default), waypoint 0 dropped, zero velocities; x0 / xg = the route's ends. */
int cfs_build_terms_from_routes_device(cfs_problem *p, int B, const double *routes, int nwp,
                                       double *x_init, double *xR1, double *ff, double *caug, void *stream);
/* The same for routes of DIFFERENT lengths, as cfs_rrt_grow_device leaves them: routes is B x nwp_stride x njoint, route b
 * has nwp[b] >= 2 rows (nwp: DEVICE int array; a route of a single row -- a start inside the goal region -- is treated as
 * start = goal).  RRTstar_CFS.m:96-100 resamples size(self.route,2) waypoints to horizon+1 samples whatever the length. */
int cfs_build_terms_from_ragged_routes_device(cfs_problem *p, int B, const double *routes, int nwp_stride, const int *nwp,
                                              double *x_init, double *xR1, double *ff, double *caug, void *stream);

/* ---- baseline cost ---------------------------------------------------------------------------
 * replaces: Cost_b = EVAL(sys_info).get_Cost_b() (Lib/EVAL.m:75-78, called at main_FANUC.m:131-132) for B problems of the
 * handle's family: u_b = quadprog(Qaug, paug) without constraints = -H^{-1} ff (H = QQ symmetrised, as quadprog does),
 * cost_b = get_cost(u_b) = 0.5*u_b'*QQ*u_b + ff'*u_b + caug (Lib/EVAL.m:51-53).  HOST pointers.  ff: B x nn; caug: B;
 * cost_b: B; u_b: B x nn (may be NULL).  Works for handles of either mode (the family's QQ is the same). */
int cfs_cost_b(cfs_problem *p, int B, const double *ff, const double *caug, double *cost_b, double *u_b);
/* replaces: cost = self.eval.get_cost(u) (Lib/EVAL.m:51-53) for B given u (B x nn): 0.5*u'*QQ*u + ff'*u + caug.  HOST pointers. */
int cfs_get_cost(cfs_problem *p, int B, const double *u, const double *ff, const double *caug, double *cost);

/* ---- measurement ------------------------------------------------------------------------------
 * When enabled, cfs_solve_batch_device brackets each kernel launch with hipEvents recorded on the
 * caller's stream (the reference has only tic/toc around the solver calls, main_FANUC.m:140-152).
 * cfs_profile_read synchronises on those events and returns, accumulated since the last read: the
 * milliseconds spent in the fused solve kernel (the event pair sits directly around its launch, after
 * the launch-order pre-pass; for handles with mesh obstacles it spans the loop of per-iteration
 * launches) and in the MFMA batched product, and the number of solves. */
int cfs_profile_enable(cfs_problem *p, int on);
int cfs_profile_read(cfs_problem *p, double *solve_kernel_ms, double *gemm_kernel_ms, int *solves);

/* ---- pieces of the path (host pointers; for callers that drive the outer loop themselves and
 *      for kernel-level parity tests) ------------------------------------------------------- */

/* [d,linkid] = dist_arm_all(theta,base,obs{j}.l,robot) (Lib/CFS_FANUC.m:115 ->
 * Lib/200i/dist_arm_3D_200i_2.m:1-30 | Lib/M16iB/dist_arm_3D_Heu_2.m | Lib/2L/dist_arm_2L.m)
 * for N configurations x nobs obstacles.  theta: N x njoint; obs: nobs x 6;
 * d: N x nobs; linkid: N x nobs (1-based); pos (optional): N x njoint x 6 capsule end points
 * [pos{i}.p(:,1); pos{i}.p(:,2)] (Lib/functions/CapPos.m:18-20). */
int cfs_dist_arm(const cfs_robot *robot, int njoint, int N, const double *theta, int nobs, const double *obs,
                 double *d, int *linkid, double *pos);

/* the distance/Jacobian half of get_con (Lib/CFS_FANUC.m:110-121): for every (problem, obstacle,
 * waypoint) the distance, closest link and Diff = num_jac(f,theta)' (Lib/functions/num_jac.m:1-17,
 * literal scheme).  x_: B x (H*nstate); obs: B x nobs x 6;
 * dist: B x nobs x H; linkid: B x nobs x H; grad: B x nobs x H x njoint. */
int cfs_linearize(cfs_problem *p, int B, const double *x_, const double *obs, double *dist, int *linkid, double *grad);

/* self.get_con() with the reference's public dense outputs self.Ainq / self.binq
 * (Lib/CFS_FANUC.m:101-135): rows = nobs*H*(1+2*njoint) in the reference's row order.
 * x_: B x (H*nstate); u: B x nn; xR1: B x nstate; obs: B x nobs x 6;
 * Ainq: B x (rows x nn column-major); binq: B x rows. */
int cfs_get_con(cfs_problem *p, int B, const double *x_, const double *u, const double *xR1, const double *obs,
                double *Ainq, double *binq);

/* one QP of the path for B problems on given linearisation data (dist/grad as returned by
 * cfs_linearize at u_lin): CFS mode  = quadprog(QQ,ff,Ainq,binq,[],[],-MAX_input,MAX_input)
 * (Lib/CFS_FANUC.m:85); PSGCFS mode = quadprog(I,-u_,Ainq,binq) (Lib/PSGCFS_FANUC.m:117-120) with
 * u_ passed in `lin`.  lin: B x nn (CFS: ff; PSGCFS: u_); u_lin: B x nn; xR1: B x nstate.
 * Outputs u: B x nn; lambda (optional): B x (nobs*H + 4*nn) multipliers ordered
 * [collision (j,i) | vel+ (i,c) | vel- (i,c) | bound+ | bound-]; qp_iter, status: B. */
int cfs_qp(cfs_problem *p, int B, const double *lin, const double *u_lin, const double *xR1,
           const double *dist, const double *grad, double *u, double *lambda, int *qp_iter, int *status);

/* ---- mesh obstacles -----------------------------------------------------
 * The reference measures the arm against a surface with `[dis, points] = point2surface_dis(pos{i}.p, obs)`
 * (M200i/dist_arm_surf_200i.m:21, Lib/functions/dist_arm_surface.m:43) and loads its maps with stlread
 * (Lib/functions/MapFromSTL.m:1-11), but contains neither function.  The contract here is the build's own:
 * dis = min over the triangles of the Euclidean distance between the link axis and the triangle (0 when they
 * intersect); points = [closest point on the link axis ; closest point on the mesh]; equal distances resolve
 * to the smaller parameter along the axis.  A cfs_mesh owns the device copy of the triangles and of a bounding
 * volume hierarchy; it lives on the device selected with cfs_set_device at creation. */
typedef struct cfs_mesh cfs_mesh; /* opaque */

/* vertices: nv x 3 (x, y, z per row); triangles: nt x 3 vertex indices (0-based); HOST pointers */
int cfs_mesh_create(const double *vertices, int nv, const int *triangles, int nt, cfs_mesh **out);
/* binary STL file.  map_from_stl != 0 applies Lib/functions/MapFromSTL.m:6-10 (every axis shifted to start at 0,
 * y -= 100, then (x, y, z) <- (z, x, y)); every coordinate is finally multiplied by `scale` (the maps are in mm). */
int cfs_mesh_load_stl(const char *path, double scale, int map_from_stl, cfs_mesh **out);
/* any output may be NULL; bbox6 = [min xyz, max xyz] */
int cfs_mesh_info(const cfs_mesh *m, int *ntri, int *nnodes, int *depth, double *bbox6);
void cfs_mesh_destroy(cfs_mesh *m);

/* point2surface_dis for n segments (HOST pointers): segs n x 6 = [p(:,1); p(:,2)]; dis n;
 * points n x 6 (may be NULL); tri n = index of the closest triangle in the caller's list (may be NULL) */
int cfs_mesh_segment_distance(const cfs_mesh *m, int n, const double *segs, double *dis, double *points, int *tri);

/* dist_arm_surf_200i (M200i/dist_arm_surf_200i.m:1-29) for N poses (HOST pointers): theta N x njoint;
 * d N; linkid N (1-based, may be NULL); points N x 6 of the closest link (may be NULL).  Same near-zero
 * surrogate (:22-24) and first-minimum rule (:25-28) as cfs_dist_arm. */
int cfs_dist_arm_mesh(const cfs_robot *robot, int njoint, int N, const double *theta, const cfs_mesh *m,
                      double *d, int *linkid, double *points);

/* From now on the LAST nmesh of the handle's nobs obstacles are these meshes (their entries of cfs_batch_in.obs are
 * ignored; margin[] still applies per obstacle): every later solve measures the arm against them with the contract
 * above, rows ordered as get_con orders obstacles (Lib/CFS_FANUC.m:111).  nmesh = 0 restores line obstacles only.
 * The meshes must outlive the solves. */
int cfs_problem_set_meshes(cfs_problem *p, int nmesh, const cfs_mesh *const *meshes);

/* ---- CHOMP_FANUC ----------------------------------------------------------
 * self = CHOMP_FANUC(obs_, sys_info, uref, ROBOT); self = self.optimizer()  (Lib/CHOMP_FANUC.m:34-69) for B problems of
 * the handle's family (robot, H, QQ, alpha, epsilon_O, MAX_O_ITER; create it in CFS mode).  HOST pointers.
 * in: x_init, xR1, ff, caug, obs as for cfs_solve_batch (noise unused); u0: B x nn = uref; D, epsilon: nobs =
 * obs_{j+1}.D / .epsilon.  out: u, x_, cost_all / e_cost_all / e_u_all (B x MAX_O_ITER), iter_O; total_iter (0) and
 * status may be NULL.  The update is the reference's, literally (step 3*alpha, gradient rows Baug((i-1)*njoint+1:
 * i*njoint,:), dm_f without the M200i joint offset, derivest derivatives): see csrc/cfs_chomp.hip for the list. */
int cfs_chomp_batch(cfs_problem *p, const cfs_batch_in *in, const double *u0, const double *D, const double *epsilon,
                    const cfs_batch_out *out);

/* ---- RRT / RRT* tree growth ---------------------------------------------------------
 * replaces: RRT_FANUC(obs, sys_info, goal, region_g, region_s, sample_off, ROBOT, SOLVER).find_route() (Lib/RRT_FANUC.m:48-91)
 * for S independent trees at once -- the seeds Lib/functions/s_Parallel_rrt.m:14-28 spreads over a parfor pool.  One wavefront
 * grows one tree entirely on the device (nearest neighbour under the `ratial`-weighted norm, 0.1-rad extension, capsule
 * feasibility against every obstacle with the same FK + distLinSeg + near-zero surrogate as the CFS path, RRT* re-parenting
 * within `rewire` of the sample, goal test, failure at node_num > MAX_ITER, route back-tracking); the reference's quirks are
 * kept (csrc/cfs_rrt.hip lists them).  MATLAB's rand stream cannot be reproduced: pass the uniforms (S x ndraw, consumed as
 * the reference consumes rand: one per proposal, nstate more when the sample is random) or NULL + a seed for the library's
 * counter-based generator (u = splitmix64 finaliser of seed + tree*0x9E3779B97F4A7C15 + (counter+1)*0xBF58476D1CE4E5B9,
 * top 53 bits * 2^-53). */
typedef enum cfs_rrt_solver { CFS_RRT = 0, CFS_RRT_STAR = 1 } cfs_rrt_solver;   /* SOLVER 'RRT' | 'RRT*' (Lib/RRT_FANUC.m:70-84) */
typedef struct cfs_rrt_desc {
    cfs_robot robot;          /* sys_info.robot (+ ROBOT: robot.kind decides the theta(2) - pi/2 offset, :158-160)   */
    int nstate;               /* sys_info.nstate = joints of the tree (2..6)                                         */
    int solver;               /* cfs_rrt_solver                                                                      */
    int max_iter;             /* MAX_ITER (400, :37); <= 1000                                                        */
    double bi;                /* goal bias threshold (0.5, :38): pp < bi -> random sample, else goal_th              */
    double rewire;            /* RRT* re-parenting radius around the sample (0.2, :135)                              */
    int per_tree;             /* 0: x0 / goal / goal_th are nstate vectors shared by all trees; 1: S x nstate         */
    const double *x0;         /* sys_info.x0                                                                         */
    const double *goal;       /* goal (centre of the goal region, :195-197)                                          */
    const double *goal_th;    /* sys_info.goal_th (the biased sample, :113)                                          */
    const double *region_g, *region_s, *sample_off, *ratial;   /* nstate each (:108-111, :117, :195-197)             */
    int nobs;
    const double *obs;        /* nobs x 6: [obs{j}.l(:,1); obs{j}.l(:,2)]                                            */
    const double *D;          /* nobs: obs{j}.D (:174)                                                               */
    const double *uniforms;   /* S x ndraw draws of rand, or NULL                                                    */
    int ndraw;
    unsigned long long seed;  /* generator mode (uniforms == NULL)                                                   */
    long long max_draws;      /* generator mode: uniforms a tree may consume before it gives up (fail = 2)           */
} cfs_rrt_desc;
typedef struct cfs_rrt_out {
    int *node_num;            /* S : self.node_num                                                                    */
    int *fail;                /* S : 0 route found | 1 node_num > MAX_ITER ("Failed to find path.", :201-205) | 2 the uniforms
                                     ran out while sampling | 3 RRT* re-parenting closed a cycle (the reference would never return) */
    int *parent;              /* S x (max_iter+1) : self.all_nodes(1,:) (1-based, -1 for the root)                    */
    double *nodes;            /* S x (max_iter+1) x nstate : self.all_nodes(2:end,:)' (one node per row)              */
    double *total_dis;        /* S x (max_iter+1) : self.total_dis                                                    */
    double *all_ee;           /* S x max_iter x 3 : self.all_ee' (may be NULL)                                        */
    int *route_len;           /* S : size(self.route,2)                                                               */
    double *route;            /* S x (max_iter+1) x nstate : self.route' (first route_len rows), start to goal       */
    long long *draws_used;    /* S : uniforms consumed (may be NULL)                                                  */
    long long *proposals;     /* S : getRandNode calls (may be NULL)                                                  */
} cfs_rrt_out;
/* HOST pointers in the descriptor and in `out`; copies in, grows, copies out, synchronises */
int cfs_rrt_grow(const cfs_rrt_desc *d, int S, const cfs_rrt_out *out);
/* DEVICE pointers (every array of the descriptor and of `out`; the descriptor struct itself is host memory), enqueued on
 * `stream`; routes can go straight into cfs_build_terms_from_ragged_routes_device */
int cfs_rrt_grow_device(const cfs_rrt_desc *d, int S, const cfs_rrt_out *out, void *stream);

/* ---- developer / test entry points -------------------------------------------------------------------
 * No caller of the path needs these; they exist so that every shortcut the solver takes can be switched off and compared
 * under pytest (tests/test_gpu_shortcuts.py), and for the cycle-stamp / step-trace probes under tools/.  All state is per
 * handle: nothing is read from the environment, nothing is process-wide.  Results with and without each switch are the
 * same optimum of the same strictly convex QPs; what the tests assert bit for bit, and what to a tolerance, is stated there. */
#define CFS_DBG_GATHER_ROLLOUTS 1   /* H = QQ: load the precomputed rollouts of the family-matrix columns instead of prefix sums in LDS */
#define CFS_DBG_NO_REFINE 2         /* no iterative refinement of the step directions                                                  */
#define CFS_DBG_NO_WARM_START 8     /* every QP starts from the empty active set                                                       */
#define CFS_DBG_NO_CERTIFICATE 16   /* CFS_FANUC: no step-free infeasibility certificate (infeasible QPs are proven by the dual steps)  */
#define CFS_DBG_NO_PRUNE 32         /* num_jac evaluates every link at every evaluation point (no candidate pruning)                    */
#define CFS_DBG_NO_AUTO_ORDER 64    /* no automatic launch order                                                                       */
#define CFS_DBG_TIER_W1 128         /* one workgroup per compute unit (64 register-resident columns of the inverse Gram matrix)         */
/* mask: OR of CFS_DBG_*; warm_max: largest previous active set a warm start takes (0 = default: 24 rows for CFS_FANUC, the
 * register-resident columns for PSGCFS_FANUC; <= 64); polish_tol: relative drift of an active row at the optimum that
 * triggers the projection (<= 0 = default 1e-11).  Applies to the following solves / pieces of this handle. */
int cfs_debug_set_options(cfs_problem *p, int mask, int warm_max, double polish_tol);
/* cycle stamps: B > 0, out == NULL: enable for the next solves of <= B problems; out != NULL: read 12 accumulators per
 * problem (HOST pointer, synchronises); B <= 0, out == NULL: off */
int cfs_debug_stamps(cfs_problem *p, int B, unsigned long long *out);
/* trace of the active-set steps of problem b: 8 doubles per step, at most cap steps; cap <= 0: off.
 * cfs_debug_trace_read: out = (cap+1)*8 doubles (HOST), out[0] = number of records */
int cfs_debug_trace_begin(cfs_problem *p, int b, int cap);
int cfs_debug_trace_read(cfs_problem *p, double *out);
/* log of u after every outer iteration (either solver; the solve itself is unchanged): on != 0 allocates
 * max_batch x MAX_O_ITER x nn doubles; cfs_debug_read_u_log copies the first B problems to `out` (HOST) */
int cfs_debug_log_u(cfs_problem *p, int on);
int cfs_debug_read_u_log(cfs_problem *p, int B, double *out);

#ifdef __cplusplus
}
#endif
#endif /* CFS_HIP_H */
