// cfs_mex.cpp -- MEX gateway over include/cfs_hip.h.
// Build on a machine with MATLAB + ROCm:  mex -I../include cfs_mex.cpp -L../motionplanning_5d_m_amd -lcfs_hip
// (not compiled in the build image: no MATLAB, no mex.h; the C ABI underneath is what the test suite exercises).
//
//   [u, x_, cost_all, e_cost_all, e_u_all, iter_O, total_iter, status] = cfs_mex('solve', mode, obs, sys_info, ROBOT, noise)
//        mode 0 = CFS_FANUC.optimizer (Lib/CFS_FANUC.m:62-79), 1 = PSGCFS_FANUC.optimizer (Lib/PSGCFS_FANUC.m:65-82);
//        obs = the reference's obs cell (obs{j}.l 3x2, .epsilon, .D; obs{j}.mesh = handle for a mesh obstacle, last in the cell);
//        noise = nn x rows matrix of normrnd(0,0.1) draws consumed one column per PSG step (PSGCFS_FANUC.m:109), or []
//   [Ainq, binq] = cfs_mex('get_con', mode, obs, sys_info, ROBOT, x_, u)   % self.get_con() (Lib/CFS_FANUC.m:101-135): dense, reference row order
//   [u, x_, cost_all, e_cost_all, e_u_all, iter_O] = cfs_mex('chomp', obs_, sys_info, ROBOT, uref)   % CHOMP_FANUC.optimizer (Lib/CHOMP_FANUC.m:54-69);
//        obs_ = the reference's cell: obs_{1}.num_obs followed by the obstacles (M16iB/CHOMP.m:26-29)
//   [d, linkid] = cfs_mex('dist_arm', theta, obs_l, robot, ROBOT)  % dist_arm_3D_200i_2 / dist_arm_3D_Heu_2 / dist_arm_2L(theta, base, obs_l, robot)
//        theta = njoint x N (one pose per column), obs_l = 3x2 obstacle axis (or 6 x nobs, one [l(:,1); l(:,2)] per column): the
//        geometry kernel RRT_FANUC.feasible (Lib/RRT_FANUC.m:146-181) and get_con (Lib/CFS_FANUC.m:115) call; d, linkid = nobs x N
//   [route, all_nodes, total_dis, all_ee, fail, node_num] = cfs_mex('rrt', obs, sys_info, goal, region_g, region_s, sample_off, ROBOT, SOLVER, U)
//        RRT_FANUC(obs, sys_info, goal, region_g, region_s, sample_off, ROBOT, SOLVER).find_route() (Lib/RRT_FANUC.m:48-91) grown on the GPU;
//        U = rand(ndraw, S): MATLAB's own rand, consumed per tree exactly as find_route consumes it (one per proposal + nstate for a random
//        sample); S > 1 grows S seeds at once (s_Parallel_rrt.m:16's parfor) and the outputs become cells
//   Cost_b = cfs_mex('cost_b', sys_info, ROBOT)                    % EVAL.get_Cost_b (Lib/EVAL.m:75-78, main_FANUC.m:131-132)
//   h = cfs_mex('mesh_load_stl', path, scale, map_from_stl)        % Lib/functions/MapFromSTL.m
//   [dis, points] = cfs_mex('mesh_segment_distance', h, seg6)      % point2surface_dis (M200i/dist_arm_surf_200i.m:21)
//   cfs_mex('mesh_destroy', h)
#include "mex.h"
#include "cfs_hip.h"
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

static double field_scalar(const mxArray *s, const char *name)
{
    const mxArray *f = mxGetField(s, 0, name);
    if (!f) mexErrMsgIdAndTxt("cfs:field", "sys_info.%s is missing", name);
    return mxGetScalar(f);
}
static double *field_ptr(const mxArray *s, const char *name, bool required = true)
{
    const mxArray *f = mxGetField(s, 0, name);
    if (!f) { if (required) mexErrMsgIdAndTxt("cfs:field", "sys_info.%s is missing", name); return nullptr; }
    return mxGetPr(f);
}
static void check(int rc) { if (rc != CFS_SUCCESS) mexErrMsgIdAndTxt("cfs:abi", "%s", cfs_last_error()); }
static cfs_mesh *mesh_of(const mxArray *h) { return reinterpret_cast<cfs_mesh *>(static_cast<uintptr_t>(*static_cast<uint64_t *>(mxGetData(h)))); }

static void fill_robot(const mxArray *robot, const char *ROBOT, int nj, cfs_robot &r)
{
    memset(&r, 0, sizeof r);
    r.kind = !strcmp(ROBOT, "M200i") ? CFS_ROBOT_M200I : (!strcmp(ROBOT, "2L") ? CFS_ROBOT_2L : CFS_ROBOT_M16IB);   // CFS_FANUC.m:49-54
    const mxArray *DH = mxGetField(robot, 0, "DH"), *cap = mxGetField(robot, 0, "cap"), *T = mxGetField(robot, 0, "T");
    r.nlink = r.kind == CFS_ROBOT_2L ? nj : (int)mxGetM(DH);
    if (DH) memcpy(r.DH, mxGetPr(DH), sizeof(double) * 4 * r.nlink);                 // nlink x 4, column-major as MATLAB holds it
    memcpy(r.base, mxGetPr(mxGetField(robot, 0, "base")), sizeof(double) * 3);
    for (int i = 0; i < r.nlink && i < (int)mxGetNumberOfElements(cap); ++i)          // robot.cap{i}.p is 3x2 = [p1 p2]
        memcpy(r.cap + 6 * i, mxGetPr(mxGetField(mxGetCell(cap, i), 0, "p")), sizeof(double) * 6);
    if (T) memcpy(r.T, mxGetPr(T), sizeof(double) * 9);                              // 2L: robot.T (robotproperty2.m:117-119)
    r.delta_t = mxGetScalar(mxGetField(robot, 0, "delta_t"));
}

// the problem-family handle of one MATLAB object: obs cell (first = index of the first obstacle in it) + sys_info + ROBOT
struct Family {
    cfs_problem *p = nullptr;
    cfs_problem_desc d;
    std::vector<double> margin, obs6, Dv, epsv;
    ~Family() { if (p) cfs_problem_destroy(p); }
};
static void make_family(Family &f, int mode, const mxArray *obs, int first, int nobs, const mxArray *S, const char *ROBOT, bool need_both = false)
{
    cfs_problem_desc &d = f.d;
    memset(&d, 0, sizeof d);
    d.mode = mode;
    d.H = (int)field_scalar(S, "H");
    d.njoint = (int)field_scalar(S, "njoint");
    d.nobs = nobs;
    fill_robot(mxGetField(S, 0, "robot"), ROBOT, d.njoint, d.robot);
    d.QQ = field_ptr(S, "QQ"); d.Aaug = field_ptr(S, "Aaug"); d.Baug = field_ptr(S, "Baug"); d.lim = field_ptr(S, "lim");
    d.MAX_input = field_ptr(S, "MAX_input", mode == CFS_MODE_CFS);
    d.epsilon_O = field_scalar(S, "epsilon_O");
    d.MAX_O_ITER = (int)field_scalar(S, "MAX_O_ITER");
    d.alpha = mxGetField(S, 0, "alpha") ? field_scalar(S, "alpha") : 0.0;
    d.max_batch = 1;
    f.margin.assign(nobs, 0.0); f.obs6.assign(6 * (size_t)nobs, 0.0); f.Dv.assign(nobs, 0.0); f.epsv.assign(nobs, 0.0);
    std::vector<const cfs_mesh *> meshes;
    for (int j = 0; j < nobs; ++j) {
        const mxArray *o = mxGetCell(obs, first + j);
        if (!o) mexErrMsgIdAndTxt("cfs:obs", "obs{%d} is empty", first + j + 1);
        // the margin field the mode reads must exist (CFS_FANUC.m:117 reads .epsilon, PSGCFS_FANUC.m:158 reads .D); the other one is
        // optional (0) -- a cell written for one solver need not carry the other solver's field
        const mxArray *fD = mxGetField(o, 0, "D"), *fE = mxGetField(o, 0, "epsilon");
        if (!(mode == CFS_MODE_CFS ? fE : fD))
            mexErrMsgIdAndTxt("cfs:obs", "obs{%d}.%s is missing", first + j + 1, mode == CFS_MODE_CFS ? "epsilon" : "D");
        if (need_both && !(fD && fE)) mexErrMsgIdAndTxt("cfs:obs", "obs_{%d} needs both .D and .epsilon (Lib/CHOMP_FANUC.m:95-96)", first + j + 1);
        f.Dv[j] = fD ? mxGetScalar(fD) : 0.0;
        f.epsv[j] = fE ? mxGetScalar(fE) : 0.0;
        f.margin[j] = mode == CFS_MODE_CFS ? f.epsv[j] : f.Dv[j];
        const mxArray *mh = mxGetField(o, 0, "mesh");
        if (mh) meshes.push_back(mesh_of(mh));
        else if (!meshes.empty()) mexErrMsgTxt("mesh obstacles must come last in the obs cell");
        else {
            const mxArray *fl = mxGetField(o, 0, "l");
            if (!fl || mxGetNumberOfElements(fl) != 6) mexErrMsgIdAndTxt("cfs:obs", "obs{%d}.l must be 3x2", first + j + 1);
            memcpy(&f.obs6[6 * (size_t)j], mxGetPr(fl), sizeof(double) * 6);   // [l(:,1); l(:,2)]
        }
    }
    d.margin = f.margin.data();
    check(cfs_problem_create(&d, &f.p));
    if (!meshes.empty()) check(cfs_problem_set_meshes(f.p, (int)meshes.size(), meshes.data()));
}

static void solve(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs < 5) mexErrMsgTxt("cfs_mex('solve', mode, obs, sys_info, ROBOT [, noise])");
    const int mode = (int)mxGetScalar(prhs[1]);
    const mxArray *S = prhs[3];
    const std::string ROBOT = mxArrayToString(prhs[4]);
    Family f;
    make_family(f, mode, prhs[2], 0, (int)mxGetNumberOfElements(prhs[2]), S, ROBOT.c_str());
    const cfs_problem_desc &d = f.d;
    const int nn = d.H * d.njoint, nx = d.H * 2 * d.njoint, K = d.MAX_O_ITER;
    double caug = field_scalar(S, "caug");
    cfs_batch_in in;
    memset(&in, 0, sizeof in);
    in.B = 1;
    in.x_init = field_ptr(S, "x_"); in.xR1 = field_ptr(S, "xR"); in.ff = field_ptr(S, "ff"); in.caug = &caug; in.obs = f.obs6.data();
    if (nrhs > 5 && !mxIsEmpty(prhs[5])) { in.noise = mxGetPr(prhs[5]); in.noise_rows = (int)mxGetN(prhs[5]); }   // nn x rows, one column per draw
    mxArray *o_u = mxCreateDoubleMatrix(nn, 1, mxREAL), *o_x = mxCreateDoubleMatrix(nx, 1, mxREAL);
    mxArray *o_c = mxCreateDoubleMatrix(K, 1, mxREAL), *o_ec = mxCreateDoubleMatrix(K, 1, mxREAL), *o_eu = mxCreateDoubleMatrix(K, 1, mxREAL);
    int iter_O = 1, total_iter = 0, status = 0;
    cfs_batch_out out;
    out.u = mxGetPr(o_u); out.x_ = mxGetPr(o_x); out.cost_all = mxGetPr(o_c); out.e_cost_all = mxGetPr(o_ec); out.e_u_all = mxGetPr(o_eu);
    out.iter_O = &iter_O; out.total_iter = &total_iter; out.status = &status;
    check(cfs_solve_batch(f.p, &in, &out));
    mxArray *outs[8] = {o_u, o_x, o_c, o_ec, o_eu, mxCreateDoubleScalar(iter_O), mxCreateDoubleScalar(total_iter), mxCreateDoubleScalar(status)};
    for (int k = 0; k < 8; ++k) { if (k < nlhs || k == 0) plhs[k] = outs[k]; else mxDestroyArray(outs[k]); }
}

// self.get_con(): dense self.Ainq / self.binq at the object's current (x_, u)  (Lib/CFS_FANUC.m:101-135, PSGCFS_FANUC.m:145-184)
static void get_con(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs < 7) mexErrMsgTxt("[Ainq, binq] = cfs_mex('get_con', mode, obs, sys_info, ROBOT, x_, u)");
    const int mode = (int)mxGetScalar(prhs[1]);
    const std::string ROBOT = mxArrayToString(prhs[4]);
    Family f;
    make_family(f, mode, prhs[2], 0, (int)mxGetNumberOfElements(prhs[2]), prhs[3], ROBOT.c_str());
    const int nn = f.d.H * f.d.njoint, rows = f.d.nobs * f.d.H * (1 + 2 * f.d.njoint);
    mxArray *A = mxCreateDoubleMatrix(rows, nn, mxREAL), *b = mxCreateDoubleMatrix(rows, 1, mxREAL);
    check(cfs_get_con(f.p, 1, mxGetPr(prhs[5]), mxGetPr(prhs[6]), field_ptr(prhs[3], "xR"), f.obs6.data(), mxGetPr(A), mxGetPr(b)));
    plhs[0] = A;
    if (nlhs > 1) plhs[1] = b; else mxDestroyArray(b);
}

// CHOMP_FANUC(obs_, sys_info, uref, ROBOT).optimizer()  (Lib/CHOMP_FANUC.m:34-69; Lib/functions/s_Solver.m:12-21)
static void chomp(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs < 5) mexErrMsgTxt("cfs_mex('chomp', obs_, sys_info, ROBOT, uref)");
    const mxArray *S = prhs[2];
    const std::string ROBOT = mxArrayToString(prhs[3]);
    const int nobs = (int)mxGetScalar(mxGetField(mxGetCell(prhs[1], 0), 0, "num_obs"));    // obs_{1}.num_obs (M16iB/CHOMP.m:26)
    Family f;
    make_family(f, CFS_MODE_CFS, prhs[1], 1, nobs, S, ROBOT.c_str(), true);
    const int nn = f.d.H * f.d.njoint, nx = f.d.H * 2 * f.d.njoint, K = f.d.MAX_O_ITER;
    double caug = field_scalar(S, "caug");
    cfs_batch_in in;
    memset(&in, 0, sizeof in);
    in.B = 1;
    in.x_init = field_ptr(S, "x_"); in.xR1 = field_ptr(S, "xR"); in.ff = field_ptr(S, "ff"); in.caug = &caug; in.obs = f.obs6.data();
    mxArray *o_u = mxCreateDoubleMatrix(nn, 1, mxREAL), *o_x = mxCreateDoubleMatrix(nx, 1, mxREAL);
    mxArray *o_c = mxCreateDoubleMatrix(K, 1, mxREAL), *o_ec = mxCreateDoubleMatrix(K, 1, mxREAL), *o_eu = mxCreateDoubleMatrix(K, 1, mxREAL);
    int iter_O = 1;
    cfs_batch_out out;
    memset(&out, 0, sizeof out);
    out.u = mxGetPr(o_u); out.x_ = mxGetPr(o_x); out.cost_all = mxGetPr(o_c); out.e_cost_all = mxGetPr(o_ec); out.e_u_all = mxGetPr(o_eu);
    out.iter_O = &iter_O;
    check(cfs_chomp_batch(f.p, &in, mxGetPr(prhs[4]), f.Dv.data(), f.epsv.data(), &out));
    mxArray *outs[6] = {o_u, o_x, o_c, o_ec, o_eu, mxCreateDoubleScalar(iter_O)};
    for (int k = 0; k < 6; ++k) { if (k < nlhs || k == 0) plhs[k] = outs[k]; else mxDestroyArray(outs[k]); }
}

// [d, linkid] = dist_arm_*(theta, base, obs_l, robot) for N poses x nobs obstacle axes: the primitive under RRT_FANUC.feasible
// (Lib/RRT_FANUC.m:146-181) and get_con (Lib/CFS_FANUC.m:115); matlab/dist_arm_3D_200i_2.m is the one-line shim over it
static void dist_arm(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs < 5) mexErrMsgTxt("[d, linkid] = cfs_mex('dist_arm', theta, obs_l, robot, ROBOT)");
    const std::string ROBOT = mxArrayToString(prhs[4]);
    const int nj = (int)mxGetM(prhs[1]), N = (int)mxGetN(prhs[1]);
    if (mxGetNumberOfElements(prhs[2]) % 6) mexErrMsgTxt("obs_l must be 3x2 (or 6 x nobs)");
    const int nobs = (int)(mxGetNumberOfElements(prhs[2]) / 6);
    cfs_robot r;
    fill_robot(prhs[3], ROBOT.c_str(), nj, r);
    std::vector<double> d((size_t)N * nobs);
    std::vector<int> lid((size_t)N * nobs);
    check(cfs_dist_arm(&r, nj, N, mxGetPr(prhs[1]), nobs, mxGetPr(prhs[2]), d.data(), lid.data(), nullptr));   // theta njoint x N column-major = N x njoint row-major
    mxArray *od = mxCreateDoubleMatrix(nobs, N, mxREAL), *ol = mxCreateDoubleMatrix(nobs, N, mxREAL);
    for (size_t k = 0; k < d.size(); ++k) { mxGetPr(od)[k] = d[k]; mxGetPr(ol)[k] = lid[k]; }               // N x nobs row-major = nobs x N column-major
    plhs[0] = od;
    if (nlhs > 1) plhs[1] = ol; else mxDestroyArray(ol);
}

// RRT_FANUC.find_route for S = size(U,2) seeds (Lib/RRT_FANUC.m:63-91; Lib/functions/s_Parallel_rrt.m:16-25)
static void rrt(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs < 10) mexErrMsgTxt("cfs_mex('rrt', obs, sys_info, goal, region_g, region_s, sample_off, ROBOT, SOLVER, U)");
    const mxArray *obs = prhs[1], *S_ = prhs[2];
    const std::string ROBOT = mxArrayToString(prhs[7]), SOLVER = mxArrayToString(prhs[8]);
    cfs_rrt_desc d;
    memset(&d, 0, sizeof d);
    d.nstate = (int)field_scalar(S_, "nstate");
    fill_robot(mxGetField(S_, 0, "robot"), ROBOT.c_str(), d.nstate, d.robot);
    d.solver = SOLVER == "RRT*" ? CFS_RRT_STAR : CFS_RRT;
    d.max_iter = 400; d.bi = 0.5; d.rewire = 0.2;                                   // class property defaults (Lib/RRT_FANUC.m:37-38, :135)
    d.x0 = field_ptr(S_, "x0"); d.goal_th = field_ptr(S_, "goal_th"); d.ratial = field_ptr(S_, "ratial");
    d.goal = mxGetPr(prhs[3]); d.region_g = mxGetPr(prhs[4]); d.region_s = mxGetPr(prhs[5]); d.sample_off = mxGetPr(prhs[6]);
    const int nobs = (int)mxGetNumberOfElements(obs);
    std::vector<double> obs6(6 * (size_t)nobs), D(nobs);
    for (int j = 0; j < nobs; ++j) {
        const mxArray *o = mxGetCell(obs, j), *fl = o ? mxGetField(o, 0, "l") : nullptr, *fD = o ? mxGetField(o, 0, "D") : nullptr;
        if (!fl || !fD || mxGetNumberOfElements(fl) != 6) mexErrMsgIdAndTxt("cfs:obs", "obs{%d} needs .l (3x2) and .D", j + 1);
        memcpy(&obs6[6 * (size_t)j], mxGetPr(fl), sizeof(double) * 6);
        D[j] = mxGetScalar(fD);
    }
    d.nobs = nobs; d.obs = obs6.data(); d.D = D.data();
    const int ndraw = (int)mxGetM(prhs[9]), S = (int)mxGetN(prhs[9]);
    d.uniforms = mxGetPr(prhs[9]); d.ndraw = ndraw;                                 // ndraw x S column-major = S x ndraw row-major
    const size_t N = (size_t)d.max_iter + 1, nj = d.nstate;
    std::vector<int> node_num(S), fail(S), route_len(S), parent(S * N);
    std::vector<double> nodes(S * N * nj), total_dis(S * N), all_ee((size_t)S * d.max_iter * 3), route(S * N * nj);
    cfs_rrt_out o;
    memset(&o, 0, sizeof o);
    o.node_num = node_num.data(); o.fail = fail.data(); o.route_len = route_len.data(); o.parent = parent.data();
    o.nodes = nodes.data(); o.total_dis = total_dis.data(); o.all_ee = all_ee.data(); o.route = route.data();
    check(cfs_rrt_grow(&d, S, &o));
    mxArray *outs[6];
    for (int k = 0; k < 4; ++k) outs[k] = S > 1 ? mxCreateCellMatrix(1, S) : nullptr;
    outs[4] = mxCreateDoubleMatrix(1, S, mxREAL); outs[5] = mxCreateDoubleMatrix(1, S, mxREAL);
    for (int t = 0; t < S; ++t) {
        const int n = node_num[t], L = route_len[t];
        mxArray *r = mxCreateDoubleMatrix(nj, L, mxREAL), *an = mxCreateDoubleMatrix(nj + 1, n, mxREAL);
        mxArray *td = mxCreateDoubleMatrix(1, n, mxREAL), *ee = mxCreateDoubleMatrix(3, n > 0 ? n - 1 : 0, mxREAL);
        memcpy(mxGetPr(r), &route[(size_t)t * N * nj], sizeof(double) * nj * L);      // rows of `route` are MATLAB's columns
        for (int i = 0; i < n; ++i) {
            mxGetPr(an)[(size_t)i * (nj + 1)] = parent[(size_t)t * N + i];            // all_nodes = [parent; node] (Lib/RRT_FANUC.m:66, :185)
            memcpy(mxGetPr(an) + (size_t)i * (nj + 1) + 1, &nodes[((size_t)t * N + i) * nj], sizeof(double) * nj);
            mxGetPr(td)[i] = total_dis[(size_t)t * N + i];
        }
        if (n > 1) memcpy(mxGetPr(ee), &all_ee[(size_t)t * d.max_iter * 3], sizeof(double) * 3 * (n - 1));
        mxArray *one[4] = {r, an, td, ee};
        for (int k = 0; k < 4; ++k) { if (S > 1) mxSetCell(outs[k], t, one[k]); else outs[k] = one[k]; }
        mxGetPr(outs[4])[t] = fail[t]; mxGetPr(outs[5])[t] = n;
    }
    for (int k = 0; k < 6; ++k) { if (k < nlhs || k == 0) plhs[k] = outs[k]; else mxDestroyArray(outs[k]); }
}

// Cost_b = EVAL(sys_info).get_Cost_b()  (Lib/EVAL.m:75-78)
static void cost_b(mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs < 3) mexErrMsgTxt("Cost_b = cfs_mex('cost_b', sys_info, ROBOT)");
    const mxArray *S = prhs[1];
    const std::string ROBOT = mxArrayToString(prhs[2]);
    Family f;
    cfs_problem_desc &d = f.d;
    memset(&d, 0, sizeof d);
    d.mode = CFS_MODE_CFS;
    d.H = (int)field_scalar(S, "H"); d.njoint = (int)field_scalar(S, "njoint"); d.nobs = 1;
    fill_robot(mxGetField(S, 0, "robot"), ROBOT.c_str(), d.njoint, d.robot);
    d.QQ = field_ptr(S, "QQ"); d.lim = field_ptr(S, "lim"); d.MAX_input = field_ptr(S, "MAX_input");
    d.epsilon_O = field_scalar(S, "epsilon_O"); d.MAX_O_ITER = (int)field_scalar(S, "MAX_O_ITER"); d.max_batch = 1;
    const double zero = 0.0;
    d.margin = &zero;
    check(cfs_problem_create(&d, &f.p));
    double caug = field_scalar(S, "caug"), cost = 0.0;
    check(cfs_cost_b(f.p, 1, field_ptr(S, "ff"), &caug, &cost, nullptr));
    plhs[0] = mxCreateDoubleScalar(cost);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[])
{
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgTxt("cfs_mex(command, ...)");
    const std::string cmd = mxArrayToString(prhs[0]);
    if (cmd == "solve") {
        solve(nlhs, plhs, nrhs, prhs);
    } else if (cmd == "get_con") {
        get_con(nlhs, plhs, nrhs, prhs);
    } else if (cmd == "chomp") {
        chomp(nlhs, plhs, nrhs, prhs);
    } else if (cmd == "dist_arm") {
        dist_arm(nlhs, plhs, nrhs, prhs);
    } else if (cmd == "rrt") {
        rrt(nlhs, plhs, nrhs, prhs);
    } else if (cmd == "cost_b") {
        cost_b(plhs, nrhs, prhs);
    } else if (cmd == "mesh_load_stl") {
        cfs_mesh *m = nullptr;
        check(cfs_mesh_load_stl(mxArrayToString(prhs[1]), nrhs > 2 ? mxGetScalar(prhs[2]) : 1.0, nrhs > 3 && mxGetScalar(prhs[3]) != 0, &m));
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *static_cast<uint64_t *>(mxGetData(plhs[0])) = static_cast<uint64_t>(reinterpret_cast<uintptr_t>(m));
    } else if (cmd == "mesh_segment_distance") {
        const int n = (int)(mxGetNumberOfElements(prhs[2]) / 6);
        plhs[0] = mxCreateDoubleMatrix(n, 1, mxREAL);
        mxArray *pts = mxCreateDoubleMatrix(6, n, mxREAL);
        check(cfs_mesh_segment_distance(mesh_of(prhs[1]), n, mxGetPr(prhs[2]), mxGetPr(plhs[0]), mxGetPr(pts), nullptr));
        if (nlhs > 1) plhs[1] = pts; else mxDestroyArray(pts);
    } else if (cmd == "mesh_destroy") {
        cfs_mesh_destroy(mesh_of(prhs[1]));
    } else {
        mexErrMsgIdAndTxt("cfs:cmd", "unknown command %s", cmd.c_str());
    }
}
