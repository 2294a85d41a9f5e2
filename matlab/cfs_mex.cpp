// MEX gateway over include/cfs_hip.h (abridged).
#include "mex.h"
#include "cfs_hip.h"
// [u, x_, cost_all, e_cost_all, e_u_all, iter_O, total_iter, status] = cfs_mex(mode, obs, sys_info, ROBOT, noise)
void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    cfs_problem_desc d = {};                       // filled from sys_info (column-major mxGetPr pointers as-is)
    const mxArray *S = prhs[2], *robot = mxGetField(S, 0, "robot");
    d.mode   = (int)mxGetScalar(prhs[0]);
    d.H      = (int)mxGetScalar(mxGetField(S, 0, "H"));
    d.njoint = (int)mxGetScalar(mxGetField(S, 0, "njoint"));
    d.nobs   = (int)mxGetNumberOfElements(prhs[1]);
    d.QQ = mxGetPr(mxGetField(S, 0, "QQ"));   d.Aaug = mxGetPr(mxGetField(S, 0, "Aaug"));
    d.Baug = mxGetPr(mxGetField(S, 0, "Baug")); d.lim = mxGetPr(mxGetField(S, 0, "lim"));
    d.MAX_input = mxGetPr(mxGetField(S, 0, "MAX_input"));
    d.epsilon_O = mxGetScalar(mxGetField(S, 0, "epsilon_O"));
    d.MAX_O_ITER = (int)mxGetScalar(mxGetField(S, 0, "MAX_O_ITER"));
    d.alpha = mxGetScalar(mxGetField(S, 0, "alpha"));   d.max_batch = 1;
    /* robot.DH (nlink x 4, column-major), robot.base, robot.cap{i}.p, robot.T, robot.delta_t -> d.robot;
       margins: obs{j}.epsilon (CFS) or obs{j}.D (PSGCFS); obs{j}.l (3x2) -> 6 doubles per obstacle */
    cfs_problem *p;  if (cfs_problem_create(&d, &p)) mexErrMsgTxt(cfs_last_error());
    cfs_batch_in in = {1, mxGetPr(mxGetField(S,0,"x_")), mxGetPr(mxGetField(S,0,"xR")), mxGetPr(mxGetField(S,0,"ff")),
                       &caug, obs6, noise, noise_rows};
    cfs_batch_out out = { /* mxCreateDoubleMatrix outputs */ };
    if (cfs_solve_batch(p, &in, &out)) mexErrMsgTxt(cfs_last_error());
    cfs_problem_destroy(p);
}
