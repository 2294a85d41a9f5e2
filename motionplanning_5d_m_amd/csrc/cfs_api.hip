// cfs_api.hip -- host side of libcfs_hip.so: the C ABI declared in include/cfs_hip.h.
// Owns the problem-family handle (device constants + workspace) and enqueues the kernels of one
// solve on a caller-supplied stream without any host round trip inside the outer loop.
#include "cfs_device.h"
#include "cfs_host.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

namespace {

thread_local char g_err[512] = "";
int g_device = 0;

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail(CFS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

int have_device()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void build_dev_robot(const cfs_robot &r, DevRobot &d)
{
    memset(&d, 0, sizeof d);
    d.kind = r.kind;
    d.nlink = r.nlink;
    for (int i = 0; i < r.nlink && i < CFS_MAX_LINKS; ++i) {
        d.dh_d[i] = r.DH[i + 1 * r.nlink];
        d.dh_a[i] = r.DH[i + 2 * r.nlink];
        const double al = r.DH[i + 3 * r.nlink];
        d.ca[i] = cos(al);
        d.sa[i] = sin(al);
        d.th_off[i] = 0.0;
        for (int e = 0; e < 6; ++e) d.cap[i * 6 + e] = r.cap[i * 6 + e];
    }
    if (r.kind == CFS_ROBOT_M200I) d.th_off[1] = M_PI / 2;   // dist_arm_3D_200i_2.m:11
    for (int e = 0; e < 3; ++e) d.base[e] = r.base[e];
    if (r.kind == CFS_ROBOT_2L)
        for (int i = 0; i + 1 < 3 && i < CFS_MAX_LINKS; ++i)   // link i uses robot.T(:,i+2) (1-based), CapPos2.m:25
            for (int e = 0; e < 3; ++e) d.t2l[i * 3 + e] = r.T[(i + 1) * 3 + e];
    // every point of the arm stays within `reach` of every joint axis; an evaluation point of num_jac moves each joint
    // by at most eps/2, so no link-obstacle distance changes by more than nlink*eps/2*reach (the segment distance is
    // 1-Lipschitz in the end points).  The pruning margin of the linearisation is a generous multiple of that.
    double reach = 0.0, capmax = 0.0;
    for (int i = 0; i < r.nlink && i < CFS_MAX_LINKS; ++i) {
        reach += (r.kind == CFS_ROBOT_2L) ? sqrt(d.t2l[i * 3] * d.t2l[i * 3] + d.t2l[i * 3 + 1] * d.t2l[i * 3 + 1] + d.t2l[i * 3 + 2] * d.t2l[i * 3 + 2])
                                          : fabs(d.dh_a[i]) + fabs(d.dh_d[i]);
        for (int k = 0; k < 2; ++k) {
            const double *c = d.cap + i * 6 + k * 3;
            capmax = std::max(capmax, sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]));
        }
    }
    d.shift_bound = r.nlink * (1e-5 / 2) * (reach + capmax);
    d.prune_tol = 1e-3 + 8.0 * d.shift_bound;
}

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        n = count;
        return hipMalloc(reinterpret_cast<void **>(&p), (count ? count : 1) * sizeof(T));
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
};

// device staging of host arrays for the host-pointer entry points
struct Stage {
    std::vector<void *> ptrs;
    hipError_t err = hipSuccess;
    template <class T> T *up(const T *h, size_t n)
    {
        if (err != hipSuccess) return nullptr;
        void *d = nullptr;
        err = hipMalloc(&d, (n ? n : 1) * sizeof(T));
        if (err != hipSuccess) return nullptr;
        ptrs.push_back(d);
        if (h) err = hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice);
        return static_cast<T *>(d);
    }
    template <class T> void down(T *h, const T *d, size_t n)
    {
        if (err == hipSuccess && h) err = hipMemcpy(h, d, n * sizeof(T), hipMemcpyDeviceToHost);
    }
    ~Stage() { for (void *q : ptrs) (void)hipFree(q); }
};

// symmetric positive definite inverse in extended precision (once per problem family)
// lambda_max(G) <= ||G^(2^k)||_inf^(1/2^k) for symmetric G: five squarings are within n^(1/32) of it (+ margin for fp64 rounding)
double lambda_max_upper(int nn, std::vector<double> G)
{
    std::vector<double> T2((size_t)nn * nn);
    double logscale = 0.0;
    for (int it = 0; it < 5; ++it) {
        double nrm = 0.0;
        for (int i = 0; i < nn; ++i) { double s = 0.0; for (int j = 0; j < nn; ++j) s += fabs(G[i + (size_t)j * nn]); if (s > nrm) nrm = s; }
        if (!(nrm > 0.0)) return 0.0;
        for (auto &v : G) v /= nrm;
        logscale = 2.0 * (logscale + log(nrm));
        for (int i = 0; i < nn; ++i)
            for (int j = 0; j < nn; ++j) { double s = 0.0; for (int k = 0; k < nn; ++k) s += G[i + (size_t)k * nn] * G[k + (size_t)j * nn]; T2[i + (size_t)j * nn] = s; }
        G.swap(T2);
    }
    double nrm = 0.0;
    for (int i = 0; i < nn; ++i) { double s = 0.0; for (int j = 0; j < nn; ++j) s += fabs(G[i + (size_t)j * nn]); if (s > nrm) nrm = s; }
    return exp((logscale + log(nrm)) / 32.0) * 1.001;
}

bool spd_inverse(int n, const double *Asym, std::vector<double> &inv, std::vector<long double> &Li)
{
    std::vector<long double> L((size_t)n * n, 0.0L);
    Li.assign((size_t)n * n, 0.0L);
    for (int j = 0; j < n; ++j) {
        long double s = Asym[j + (size_t)j * n];
        for (int k = 0; k < j; ++k) s -= L[j + (size_t)k * n] * L[j + (size_t)k * n];
        if (!(s > 0)) return false;
        const long double ljj = sqrtl(s);
        L[j + (size_t)j * n] = ljj;
        for (int i = j + 1; i < n; ++i) {
            long double t = Asym[i + (size_t)j * n];
            for (int k = 0; k < j; ++k) t -= L[i + (size_t)k * n] * L[j + (size_t)k * n];
            L[i + (size_t)j * n] = t / ljj;
        }
    }
    for (int j = 0; j < n; ++j) {          // Li = L^{-1}, lower
        Li[j + (size_t)j * n] = 1.0L / L[j + (size_t)j * n];
        for (int i = j + 1; i < n; ++i) {
            long double s = 0.0L;
            for (int k = j; k < i; ++k) s -= L[i + (size_t)k * n] * Li[k + (size_t)j * n];
            Li[i + (size_t)j * n] = s / L[i + (size_t)i * n];
        }
    }
    inv.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {     // A^{-1} = Li' Li
            long double s = 0.0L;
            for (int k = i; k < n; ++k) s += Li[k + (size_t)i * n] * Li[k + (size_t)j * n];
            inv[i + (size_t)j * n] = inv[j + (size_t)i * n] = (double)s;
        }
    return true;
}

}  // namespace

struct cfs_problem {
    cfs_problem_desc d;
    int device;
    int nn, ns, nx;
    double lmax_vel, lmax_H;
    DevRobot hrobot;
    DevBuf<DevRobot> rb;
    DevBuf<double> QQ, Hinv, Hq, M1n, M2n, lim, maxin, margin;
    DevBuf<double> F1, F2, Cq;   // per-problem cost terms from (x0, xg), set by cfs_set_state_cost
    DevBuf<DevCost> cost;        // structure of QQ (handles created from the cost weights)
    std::vector<double> QQ_host; // what cfs_problem_family hands back
    DevBuf<double> Mr[6];   // rollouts (Bvel*, Bpos*) of the columns of M1n, M2n, Hq
    // workspace (max_batch problems)
    DevBuf<double> x0, qu, dist, grad, Yg, Pt, u_hist, qu_hist;
    DevBuf<int> noise_row, linkid, pool_flag;
    bool pool_dirty = false;   // a solve of this handle failed to enqueue: clear the spill-pool flags before the next one
    int pool_n = 1;            // slots of the spill pool (Yg / Pt): one per workgroup that can be resident at once, never more than max_batch
    DevBuf<int> order, okey;   // launch order of the fused solver, automatic: violation count of the initial trajectory -> rank
    DevBuf<int> order_user;    // the caller's permutation (cfs_set_launch_order); a solve of another batch size falls back to the automatic order
    int n_cu = 256;            // compute units of the handle's device: a batch of at most n_cu problems starts all at once
    int order_mode = 0, order_n = 0;   // 0 automatic, 1 given (order_n entries), 2 identity
    // mesh obstacles (cfs_problem_set_meshes): the last nmesh of the nobs obstacles
    int nmesh = 0;
    DevBuf<DevMesh> meshes_d;
    DevBuf<double> st_cost, m_ends, m_base, m_shift, m_upper, m_pd, m_pnd;
    DevBuf<int> st_done, m_tri, m_near, m_pi;
    // developer / test switches (cfs_debug_*, include/cfs_hip.h): per handle, no process-wide state
    int dbg_mask = 0, dbg_warm_max = 0;
    double dbg_polish_tol = 1e-11;        // = the constraint scan's own feasibility tolerance
    DevBuf<unsigned long long> stamps;    // 12 cycle accumulators per problem
    int stamps_B = 0;
    DevBuf<double> trace;                 // 8 doubles per active-set step of problem trace_b
    int trace_b = -1, trace_cap = 0;
    DevBuf<double> u_log;                 // max_batch x MAX_O_ITER x nn: u after every outer iteration (both solvers)
    bool prof = false;
    std::vector<hipEvent_t> ev;   // 4 per profiled solve: gemm start/stop, fused start/stop
    std::vector<hipEvent_t> ev_free;   // recycled events: none is created inside a timed region once the pool is warm
    void release_all()
    {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_free) (void)hipEventDestroy(e);
        ev.clear(); ev_free.clear();
        rb.release(); QQ.release(); Hinv.release(); Hq.release();
        M1n.release(); M2n.release(); Pt.release(); u_hist.release(); qu_hist.release();
        for (auto &m : Mr) m.release();
        F1.release(); F2.release(); Cq.release(); cost.release();
        lim.release(); maxin.release(); margin.release(); x0.release(); qu.release(); dist.release();
        grad.release(); Yg.release(); noise_row.release(); order.release(); okey.release(); order_user.release();
        linkid.release(); pool_flag.release(); meshes_d.release(); st_cost.release(); st_done.release();
        m_ends.release(); m_base.release(); m_shift.release(); m_tri.release(); m_near.release(); m_upper.release();
        m_pd.release(); m_pnd.release(); m_pi.release();
        stamps.release(); trace.release(); u_log.release();
    }
};

int cfs_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
int cfs_current_device() { return g_device; }
void cfs_build_dev_robot(const cfs_robot &r, DevRobot &d) { build_dev_robot(r, d); }

// Tier of the fused kernel (cfs_device.h).  Two problems per CU win whenever they fit: measured on config 3, PSGCFS
// 3.6 -> 2.3 ms per solve with w2s, CFS 5.8 -> 5.6 ms with w2m (its infeasibility proofs run active sets of ~100 rows).
// w2s is compiled for the identity Hessian only (PSGCFS), w2m for QQ only (CFS), w1 for both; force_w1: CFS_DBG_TIER_W1.
bool fused_fits(int nj, int H, int nobs) { return fused_fits_w1(nj, H, nobs); }
hipError_t launch_fused(int nj, FusedParams p, hipStream_t s, bool force_w1)
{
    const bool ident = p.mode == CFS_MODE_PSGCFS;
    if (!force_w1 && ident && fused_fits_w2s(nj, p.H, p.nobs)) return launch_fused_w2s(nj, p, s);
    if (!force_w1 && !ident && fused_fits_w2m(nj, p.H, p.nobs)) return launch_fused_w2m(nj, p, s);
    return launch_fused_w1(nj, p, s);
}

extern "C" {

int cfs_abi_version(void) { return CFS_ABI_VERSION; }
const char *cfs_last_error(void) { return g_err; }
int cfs_device_count(void) { return have_device(); }

int cfs_set_device(int device)
{
    const int n = have_device();
    if (n == 0) return fail(CFS_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n) return fail(CFS_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
    g_device = device;
    return CFS_SUCCESS;
}

static size_t pt_stride(int nn) { return nn > 160 ? (size_t)256 * 256 : (nn > 96 ? (size_t)160 * 160 : (size_t)96 * 96); }   // QB*QB >= (QB-PR)*QB in every tier


static int check_robot(const cfs_robot *r, int nj)
{
    if (!r) return fail(CFS_ERR_INVALID_ARG, "robot is NULL");
    if (r->kind < CFS_ROBOT_M16IB || r->kind > CFS_ROBOT_2L) return fail(CFS_ERR_INVALID_ARG, "unknown robot kind %d", r->kind);
    if (r->nlink < 1 || r->nlink > CFS_MAX_LINKS) return fail(CFS_ERR_INVALID_ARG, "robot.nlink %d outside 1..%d", r->nlink, CFS_MAX_LINKS);
    if (nj < 1 || nj > 6 || nj > r->nlink) return fail(CFS_ERR_INVALID_ARG, "njoint %d unsupported (1..6, <= nlink)", nj);
    if (r->kind == CFS_ROBOT_2L && nj > 2) return fail(CFS_ERR_INVALID_ARG, "the 2L model has 2 joints");
    return CFS_SUCCESS;
}

}  // extern "C"
int cfs_check_robot(const cfs_robot *r, int nj) { return check_robot(r, nj); }
extern "C" {

int cfs_problem_create(const cfs_problem_desc *desc, cfs_problem **out)
{
    if (!desc || !out) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    const int nj = desc->njoint, H = desc->H;
    int rc = check_robot(&desc->robot, nj);
    if (rc) return rc;
    if (nj < 2) return fail(CFS_ERR_INVALID_ARG, "njoint %d unsupported (2..6)", nj);
    if (H < 1 || H > CFS_MAX_H) return fail(CFS_ERR_INVALID_ARG, "H %d outside 1..%d", H, CFS_MAX_H);
    if (desc->nobs < 1 || desc->nobs > CFS_MAX_OBS) return fail(CFS_ERR_INVALID_ARG, "nobs %d outside 1..%d", desc->nobs, CFS_MAX_OBS);
    if (desc->mode != CFS_MODE_CFS && desc->mode != CFS_MODE_PSGCFS) return fail(CFS_ERR_INVALID_ARG, "unknown mode %d", desc->mode);
    if (!desc->QQ || !desc->lim || !desc->margin) return fail(CFS_ERR_INVALID_ARG, "QQ/lim/margin must be given");
    if (desc->mode == CFS_MODE_CFS && !desc->MAX_input) return fail(CFS_ERR_INVALID_ARG, "MAX_input must be given in CFS mode");
    if (desc->max_batch < 1) return fail(CFS_ERR_INVALID_ARG, "max_batch must be >= 1");
    if (!fused_fits(nj, H, desc->nobs))
        return fail(CFS_ERR_INVALID_ARG, "H=%d x nobs=%d x njoint=%d exceeds the 160 KB on-chip budget of one problem (nobs*H*njoint*8 B of gradients must fit next to the solver state)", H, desc->nobs, nj);
    if (desc->MAX_O_ITER < 0) return fail(CFS_ERR_INVALID_ARG, "MAX_O_ITER must be >= 0");
    const double dt = desc->robot.delta_t;
    if (!(dt > 0)) return fail(CFS_ERR_INVALID_ARG, "robot.delta_t must be positive");
    const int nn = H * nj, ns = 2 * nj, nx = H * ns;

    // sys_info.Aaug / Baug must be the double integrator of robot.A / robot.B
    // (robotproperty2.m:136-139, main_FANUC.m:79-86): the kernels roll out with prefix sums.
    if (desc->Baug) {
        for (int i = 0; i < H; ++i)
            for (int k = 0; k < H; ++k)
                for (int r = 0; r < ns; ++r)
                    for (int c = 0; c < nj; ++c) {
                        double want = 0.0;
                        if (k <= i && (r % nj) == c) want = r < nj ? ((double)(i - k) + 0.5) * dt * dt : dt;
                        const double got = desc->Baug[(i * ns + r) + (size_t)(k * nj + c) * nx];
                        if (fabs(got - want) > 1e-12 * (1.0 + fabs(want)))
                            return fail(CFS_ERR_DYNAMICS, "Baug(%d,%d)=%.17g, double integrator expects %.17g", i * ns + r + 1, k * nj + c + 1, got, want);
                    }
    }
    if (desc->Aaug) {
        for (int i = 0; i < H; ++i)
            for (int r = 0; r < ns; ++r)
                for (int c = 0; c < ns; ++c) {
                    double want = (r == c) ? 1.0 : 0.0;
                    if (r < nj && c == r + nj) want = (double)(i + 1) * dt;
                    const double got = desc->Aaug[(i * ns + r) + (size_t)c * nx];
                    if (fabs(got - want) > 1e-12 * (1.0 + fabs(want)))
                        return fail(CFS_ERR_DYNAMICS, "Aaug(%d,%d)=%.17g, double integrator expects %.17g", i * ns + r + 1, c + 1, got, want);
                }
    }
    if (have_device() == 0) return fail(CFS_ERR_NO_DEVICE, "no HIP device visible");

    // H^{-1} of the QP Hessian: QQ symmetrised (quadprog does so silently) for CFS, identity for the
    // PSGCFS projection (PSGCFS_FANUC.m:117)
    std::vector<double> Hinv;
    std::vector<long double> Li;            // L^{-1} (lower), H = L L'
    {
        std::vector<double> sym((size_t)nn * nn);
        for (int j = 0; j < nn; ++j)
            for (int i = 0; i < nn; ++i) sym[i + (size_t)j * nn] = 0.5 * (desc->QQ[i + (size_t)j * nn] + desc->QQ[j + (size_t)i * nn]);
        if (!spd_inverse(nn, sym.data(), Hinv, Li)) return fail(CFS_ERR_NOT_SPD, "QQ is not positive definite");
    }
    std::vector<double> Hq;   // Hessian inverse used by the QP
    if (desc->mode == CFS_MODE_CFS) Hq = Hinv;
    else {
        Hq.assign((size_t)nn * nn, 0.0);
        for (int i = 0; i < nn; ++i) Hq[i + (size_t)i * nn] = 1.0;
    }
    // Family matrices H^{-1}Bpos', H^{-1}Bvel' (columns = constraint position (i, c), natural row order): extended-precision
    // sums over the entries of the ONE rounded matrix Hq, on purpose.  Every product n_a'H^{-1}n_p the kernel forms is then
    // exactly symmetric in (a, p) -- the QP is solved for a Hessian inverse that differs from the true one by 1 ulp per entry,
    // which is harmless -- whereas columns taken from a more accurate solve L^{-T}(L^{-1}Bpos') disagree with the rounded Hq
    // columns of the bound rows by eps*cond(H) ~ 1e-10: the step refinement then never reaches its residual target and runs
    // all its passes (measured: CFS mode 30 % slower, same answers).
    std::vector<double> M1n((size_t)nn * nn), M2n((size_t)nn * nn);
    {
        std::vector<long double> a1(nn), a2(nn);
        for (int c = 0; c < nj; ++c)
            for (int i = 0; i < H; ++i) {
                for (int r = 0; r < nn; ++r) {
                    long double s1 = 0.0L, s2 = 0.0L;
                    for (int k = 0; k <= i; ++k) {
                        const long double h = Hq[r + (size_t)(k * nj + c) * nn];
                        s1 += ((long double)(i - k) + 0.5L) * (long double)dt * (long double)dt * h;
                        s2 += (long double)dt * h;
                    }
                    a1[r] = s1; a2[r] = s2;
                }
                const int col = i * nj + c;
                for (int r = 0; r < nn; ++r) { M1n[r + (size_t)col * nn] = (double)a1[r]; M2n[r + (size_t)col * nn] = (double)a2[r]; }
            }
    }

    // rigorous upper bounds of lambda_max(H) and of lambda_max(G), G = D'HD/dt^2 (D = first difference along the waypoints, so
    // that u = D s/dt for s = Bvel u), H = the QP Hessian (QQ symmetrised | I): the early infeasibility test of the fused kernel
    double lmax_vel = 0.0, lmax_H = 1.0;
    {
        std::vector<double> Hs((size_t)nn * nn), G((size_t)nn * nn);
        for (int j = 0; j < nn; ++j)
            for (int i = 0; i < nn; ++i)
                Hs[i + (size_t)j * nn] = desc->mode == CFS_MODE_CFS ? 0.5 * ((double)desc->QQ[i + (size_t)j * nn] + desc->QQ[j + (size_t)i * nn]) : (i == j ? 1.0 : 0.0);
        auto Dt = [&](std::vector<double> &M, bool left) {     // M <- D'M (left) or M D (right): row/col k minus row/col k+nj
            for (int o = 0; o < nn; ++o)
                for (int k = 0; k + nj < nn; ++k) {
                    if (left) M[k + (size_t)o * nn] -= M[(k + nj) + (size_t)o * nn];
                    else M[o + (size_t)k * nn] -= M[o + (size_t)(k + nj) * nn];
                }
        };
        G = Hs; Dt(G, true); Dt(G, false);
        for (auto &v : G) v /= (double)(dt * dt);
        lmax_vel = lambda_max_upper(nn, G);
        if (desc->mode == CFS_MODE_CFS) lmax_H = lambda_max_upper(nn, Hs);   // not 1/alpha: alpha is the caller's PSGCFS step
    }
    // rollouts of every family column (double integrator: Bvel w = dt*cumsum(w), Bpos w = sum (i-k+1/2) dt^2 w_k)
    std::vector<double> Mroll[6];
    {
        const std::vector<double> *src[3] = {&M1n, &M2n, &Hq};
        for (int m = 0; m < 3; ++m) {
            Mroll[2 * m].assign((size_t)nn * nn, 0.0);
            Mroll[2 * m + 1].assign((size_t)nn * nn, 0.0);
            for (int col = 0; col < nn; ++col)
                for (int c = 0; c < nj; ++c) {
                    long double sv = 0.0L;
                    for (int i = 0; i < H; ++i) {
                        long double sp = 0.0L;
                        sv += (*src[m])[(i * nj + c) + (size_t)col * nn];
                        for (int k = 0; k <= i; ++k) sp += ((long double)(i - k) + 0.5L) * (*src[m])[(k * nj + c) + (size_t)col * nn];
                        Mroll[2 * m][(i * nj + c) + (size_t)col * nn] = (double)((long double)dt * sv);
                        Mroll[2 * m + 1][(i * nj + c) + (size_t)col * nn] = (double)((long double)dt * (long double)dt * sp);
                    }
                }
        }
    }
    cfs_problem *p = new (std::nothrow) cfs_problem();
    if (!p) return fail(CFS_ERR_ALLOC, "out of host memory");
    p->d = *desc;
    p->QQ_host.assign(desc->QQ, desc->QQ + (size_t)nn * nn);
    p->d.QQ = p->d.Aaug = p->d.Baug = p->d.lim = p->d.MAX_input = p->d.margin = nullptr;
    p->device = g_device;
    if (hipDeviceGetAttribute(&p->n_cu, hipDeviceAttributeMultiprocessorCount, g_device) != hipSuccess || p->n_cu < 1) p->n_cu = 256;
    p->nn = nn; p->ns = ns; p->nx = nx; p->lmax_vel = lmax_vel; p->lmax_H = lmax_H;
    build_dev_robot(desc->robot, p->hrobot);
    const size_t Bm = (size_t)desc->max_batch;
    // spill pool (rows of Y beyond LDS, columns of P beyond the registers): one slot per workgroup that can be resident at once (two per
    // compute unit), not one per problem of the batch -- config 4: 512 slots instead of 4 096 (0.4 GB instead of 3.4 GB per handle)
    p->pool_n = (int)std::min<size_t>(Bm, (size_t)2 * p->n_cu);
    const size_t Pn = (size_t)p->pool_n;
    hipError_t e = hipSetDevice(p->device);
#define A_(buf, count) if (e == hipSuccess) e = p->buf.alloc(count)
    A_(rb, 1); A_(QQ, (size_t)nn * nn); A_(Hinv, (size_t)nn * nn);
    for (int m = 0; m < 6; ++m) { A_(Mr[m], (size_t)nn * nn); }
    A_(M1n, (size_t)nn * nn); A_(M2n, (size_t)nn * nn); A_(Hq, (size_t)nn * nn); A_(Pt, Pn * pt_stride(nn)); A_(lim, nj); A_(maxin, nn); A_(margin, desc->nobs);
    A_(x0, Bm * nn); A_(qu, Bm * nn); A_(dist, Bm * desc->nobs * H); A_(grad, Bm * desc->nobs * H * nj);
    A_(Yg, Pn * nn * nn); A_(pool_flag, Pn * 16);
    if (desc->mode == CFS_MODE_CFS) { A_(u_hist, Bm * (size_t)desc->MAX_O_ITER * nn); A_(qu_hist, Bm * (size_t)desc->MAX_O_ITER * nn); }
    A_(noise_row, Bm); A_(linkid, Bm * desc->nobs * H); A_(order, Bm); A_(okey, Bm); A_(order_user, Bm);
#undef A_
#define U_(buf, src, count) if (e == hipSuccess) e = hipMemcpy(p->buf.p, src, (count) * sizeof(*p->buf.p), hipMemcpyHostToDevice)
    U_(rb, &p->hrobot, 1); U_(QQ, desc->QQ, (size_t)nn * nn); U_(Hinv, Hinv.data(), (size_t)nn * nn);
    U_(M1n, M1n.data(), (size_t)nn * nn); U_(M2n, M2n.data(), (size_t)nn * nn); U_(Hq, Hq.data(), (size_t)nn * nn);
    for (int m = 0; m < 6; ++m) { U_(Mr[m], Mroll[m].data(), (size_t)nn * nn); }
    U_(lim, desc->lim, nj); U_(margin, desc->margin, desc->nobs);
    if (desc->mode == CFS_MODE_CFS) { U_(maxin, desc->MAX_input, nn); }
    else if (e == hipSuccess) e = hipMemset(p->maxin.p, 0, nn * sizeof(double));
    if (e == hipSuccess) e = hipMemset(p->pool_flag.p, 0, Pn * 16 * sizeof(int));
#undef U_
    if (e != hipSuccess) {
        p->release_all();
        delete p;
        return fail(CFS_ERR_HIP, "device setup failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return CFS_SUCCESS;
}

void cfs_problem_destroy(cfs_problem *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    p->release_all();
    delete p;
}

int cfs_problem_family(const cfs_problem *p, double *QQ, double *alpha)
{
    if (!p) return fail(CFS_ERR_INVALID_ARG, "NULL handle");
    if (QQ) memcpy(QQ, p->QQ_host.data(), p->QQ_host.size() * sizeof(double));
    if (alpha) *alpha = p->d.alpha;
    return CFS_SUCCESS;
}

// largest eigenvalue of a symmetric matrix by cyclic Jacobi rotations (once per family, n <= 384)
static double sym_lambda_max(int n, std::vector<double> A)
{
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) (i == j ? dia : off) += A[i + (size_t)j * n] * A[i + (size_t)j * n];
        if (off <= 1e-30 * dia) break;
        for (int p_ = 0; p_ < n - 1; ++p_)
            for (int q_ = p_ + 1; q_ < n; ++q_) {
                const double apq = A[p_ + (size_t)q_ * n];
                if (apq == 0.0) continue;
                const double theta = (A[q_ + (size_t)q_ * n] - A[p_ + (size_t)p_ * n]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; ++k) {               // columns p, q
                    const double akp = A[k + (size_t)p_ * n], akq = A[k + (size_t)q_ * n];
                    A[k + (size_t)p_ * n] = c * akp - sn * akq;
                    A[k + (size_t)q_ * n] = sn * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {               // rows p, q
                    const double apk = A[p_ + (size_t)k * n], aqk = A[q_ + (size_t)k * n];
                    A[p_ + (size_t)k * n] = c * apk - sn * aqk;
                    A[q_ + (size_t)k * n] = sn * apk + c * aqk;
                }
            }
    }
    double m = A[0];
    for (int i = 1; i < n; ++i) m = std::max(m, A[i + (size_t)i * n]);
    return m;
}

int cfs_problem_create_from_weights(const cfs_problem_desc *desc, const cfs_cost_weights *w, cfs_problem **out)
{
    if (!desc || !w || !out) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (!w->Qp || !w->Qv || !w->Rblk) return fail(CFS_ERR_INVALID_ARG, "Qp/Qv/Rblk must be given");
    const int nj = desc->njoint, H = desc->H;
    if (nj < 2 || nj > 6 || H < 1 || H > CFS_MAX_H) return fail(CFS_ERR_INVALID_ARG, "njoint %d / H %d unsupported", nj, H);
    const double dt = desc->robot.delta_t;
    if (!(dt > 0)) return fail(CFS_ERR_INVALID_ARG, "robot.delta_t must be positive");
    const int ns = 2 * nj, nn = H * nj, nx = H * ns;
    // Q = [Qp qc*I; qc*I Qv] (main_FANUC.m:65-77), Qaug = blkdiag(Q*w_stage, ..., Q*w_terminal) (:81-84)
    std::vector<double> Q((size_t)ns * ns, 0.0), Qaug((size_t)nx * nx, 0.0);
    for (int c = 0; c < nj; ++c)
        for (int r = 0; r < nj; ++r) {
            Q[r + (size_t)c * ns] = w->Qp[r + c * nj];
            Q[(nj + r) + (size_t)(nj + c) * ns] = w->Qv[r + c * nj];
            if (r == c) { Q[r + (size_t)(nj + c) * ns] = w->q_cross; Q[(nj + r) + (size_t)c * ns] = w->q_cross; }
        }
    for (int i = 0; i < H; ++i) {
        const double wi = i == H - 1 ? w->w_terminal : w->w_stage;
        for (int c = 0; c < ns; ++c)
            for (int r = 0; r < ns; ++r) Qaug[(i * ns + r) + (size_t)(i * ns + c) * nx] = Q[r + (size_t)c * ns] * wi;
    }
    // T = Qaug*Baug (block rows), QQ = Baug'*T + cR*(R + R')  (:96-97); Baug(i,k) = [((i-k)+1/2) dt^2 I; dt I] for k <= i
    std::vector<double> T((size_t)nx * nn, 0.0), QQ((size_t)nn * nn, 0.0);
    for (int k = 0; k < H; ++k)
        for (int c = 0; c < nj; ++c) {
            const int col = k * nj + c;
            for (int i = k; i < H; ++i) {
                const double bp = ((double)(i - k) + 0.5) * dt * dt, bv = dt;
                for (int r = 0; r < ns; ++r)
                    T[(i * ns + r) + (size_t)col * nx] = Qaug[(i * ns + r) + (size_t)(i * ns + c) * nx] * bp + Qaug[(i * ns + r) + (size_t)(i * ns + nj + c) * nx] * bv;
            }
        }
    for (int col = 0; col < nn; ++col)
        for (int k = 0; k < H; ++k)
            for (int c = 0; c < nj; ++c) {
                double sacc = 0.0;
                for (int i = k; i < H; ++i)
                    sacc += (((double)(i - k) + 0.5) * dt * dt) * T[(i * ns + c) + (size_t)col * nx] + dt * T[(i * ns + nj + c) + (size_t)col * nx];
                QQ[(k * nj + c) + (size_t)col * nn] = sacc;
            }
    DevCost hc;
    memset(&hc, 0, sizeof hc);
    for (int c = 0; c < nj; ++c)
        for (int r = 0; r < nj; ++r) {
            hc.Qp[r + c * nj] = w->Qp[r + c * nj];
            hc.Qv[r + c * nj] = w->Qv[r + c * nj];
            hc.Rs[r + c * nj] = (w->Rblk[r + c * nj] + w->Rblk[c + r * nj]) * w->cR;
        }
    hc.qc = w->q_cross; hc.ws = w->w_stage; hc.wt = w->w_terminal;
    for (int i = 0; i < H; ++i)
        for (int c = 0; c < nj; ++c)
            for (int r = 0; r < nj; ++r) QQ[(i * nj + r) + (size_t)(i * nj + c) * nn] += hc.Rs[r + c * nj];
    cfs_problem_desc d2 = *desc;
    d2.QQ = QQ.data();
    d2.Aaug = d2.Baug = nullptr;                           // implied: the double integrator
    if (d2.alpha == 0.0 && d2.mode == CFS_MODE_PSGCFS) {   // alpha = 1/max(svd(QQ))  (main_FANUC.m:120)
        std::vector<double> sym((size_t)nn * nn);
        for (int j = 0; j < nn; ++j)
            for (int i = 0; i < nn; ++i) sym[i + (size_t)j * nn] = 0.5 * (QQ[i + (size_t)j * nn] + QQ[j + (size_t)i * nn]);
        d2.alpha = 1.0 / sym_lambda_max(nn, sym);
    }
    cfs_problem *p = nullptr;
    int rc = cfs_problem_create(&d2, &p);
    if (rc) return rc;
    rc = cfs_set_state_cost(p, Qaug.data());
    if (rc == CFS_SUCCESS) {
        hipError_t e = p->cost.alloc(1);
        if (e == hipSuccess) e = hipMemcpy(p->cost.p, &hc, sizeof hc, hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(CFS_ERR_HIP, "device setup failed: %s", hipGetErrorString(e));
    }
    if (rc) { cfs_problem_destroy(p); return rc; }
    *out = p;
    return CFS_SUCCESS;
}

// family-level fields of the fused kernel's parameter block (everything that does not change across a batch)
static void fill_fused_family(const cfs_problem *p, FusedParams &fp, int B)
{
    memset(&fp, 0, sizeof fp);
    fp.rb = p->rb.p; fp.B = B; fp.H = p->d.H; fp.nobs = p->d.nobs; fp.mode = p->d.mode;
    fp.has_bounds = p->d.mode == CFS_MODE_CFS; fp.max_o_iter = p->d.MAX_O_ITER;
    fp.dt = p->d.robot.delta_t; fp.alpha = p->d.alpha; fp.epsilon_O = p->d.epsilon_O; fp.lmax_vel = p->lmax_vel; fp.lmax_H = p->lmax_H;
    fp.M1 = p->M1n.p; fp.M2 = p->M2n.p; fp.M3 = p->Hq.p; fp.QQ = p->QQ.p; fp.cost = p->cost.p;
    fp.M1v = p->Mr[0].p; fp.M1p = p->Mr[1].p; fp.M2v = p->Mr[2].p; fp.M2p = p->Mr[3].p; fp.M3v = p->Mr[4].p; fp.M3p = p->Mr[5].p;
    fp.lim = p->lim.p; fp.maxin = p->maxin.p; fp.margin = p->margin.p;
    fp.x0 = p->x0.p;
    fp.Yg = p->Yg.p; fp.Pt = p->Pt.p; fp.pt_stride = pt_stride(p->nn); fp.pool_flag = p->pool_flag.p; fp.pool_n = p->pool_n;
    // kernel switch word: bit 0 = roll w = H^{-1} n_p out in LDS (the default since round 2: +2-4 % on config 3 CFS, a third of the
    // gather's L2 loads; CFS_DBG_GATHER_ROLLOUTS loads the precomputed rollouts of the family matrices instead), bit 1 = no
    // refinement, bit 3 = no warm start, bit 4 = no step-free certificate
    fp.opt = ((p->dbg_mask & CFS_DBG_GATHER_ROLLOUTS) ? 0 : 1) | (p->dbg_mask & (CFS_DBG_NO_REFINE | CFS_DBG_NO_WARM_START | CFS_DBG_NO_CERTIFICATE));
    fp.polish_tol = p->dbg_polish_tol;
    fp.warm_max = p->dbg_warm_max;
    fp.no_prune = (p->dbg_mask & CFS_DBG_NO_PRUNE) ? 1 : 0;
    fp.dbg = p->trace.p; fp.dbg_b = p->trace_b; fp.dbg_cap = p->trace_cap;
    fp.stamps = (p->stamps.p && B <= p->stamps_B) ? p->stamps.p : nullptr;
    fp.u_log = p->u_log.p;
}
static bool force_w1(const cfs_problem *p) { return (p->dbg_mask & CFS_DBG_TIER_W1) != 0; }

int cfs_set_launch_order(cfs_problem *p, const int *order, int n)
{
    if (!p) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    if (!order) { p->order_mode = n < 0 ? 2 : 0; p->order_n = 0; return CFS_SUCCESS; }
    if (n < 1 || n > p->d.max_batch) return fail(CFS_ERR_INVALID_ARG, "n=%d outside 1..max_batch=%d", n, p->d.max_batch);
    std::vector<char> seen((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        if (order[i] < 0 || order[i] >= n || seen[order[i]]) return fail(CFS_ERR_INVALID_ARG, "order is not a permutation of 0..%d (entry %d)", n - 1, i);
        seen[order[i]] = 1;
    }
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipDeviceSynchronize());      // a solve in flight may still be reading the previous order
    HIPCHK(hipMemcpy(p->order_user.p, order, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    p->order_mode = 1; p->order_n = n;
    return CFS_SUCCESS;
}

static int enqueue_solve(cfs_problem *p, const cfs_batch_in *in, const cfs_batch_out *out, hipStream_t s, hipEvent_t *e4);

int cfs_solve_batch_device(cfs_problem *p, const cfs_batch_in *in, const cfs_batch_out *out, void *stream)
{
    if (!p || !in || !out) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    const int B = in->B;
    if (B < 1 || B > p->d.max_batch) return fail(CFS_ERR_INVALID_ARG, "B=%d outside 1..max_batch=%d", B, p->d.max_batch);
    if (!in->x_init || !in->xR1 || !in->ff || !in->caug || !in->obs) return fail(CFS_ERR_INVALID_ARG, "NULL input array");
    if (!out->u || !out->x_ || !out->cost_all || !out->e_cost_all || !out->e_u_all || !out->iter_O || !out->total_iter || !out->status)
        return fail(CFS_ERR_INVALID_ARG, "NULL output array");
    HIPCHK(hipSetDevice(p->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);

    hipEvent_t e4[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = CFS_SUCCESS;
    if (p->prof)
        for (int k = 0; k < 4 && rc == CFS_SUCCESS; ++k) {
            if (!p->ev_free.empty()) { e4[k] = p->ev_free.back(); p->ev_free.pop_back(); }
            else if (hipEventCreate(&e4[k]) != hipSuccess) { e4[k] = nullptr; rc = fail(CFS_ERR_HIP, "hipEventCreate failed"); }
        }
    if (rc == CFS_SUCCESS) {
        rc = enqueue_solve(p, in, out, s, e4);
        if (rc != CFS_SUCCESS) p->pool_dirty = true;
    }
    if (p->prof)                         // recorded events are read by cfs_profile_read; after an error they go back to the pool
        for (int k = 0; k < 4; ++k)
            if (e4[k]) (rc == CFS_SUCCESS ? p->ev : p->ev_free).push_back(e4[k]);
    return rc;
}

static int enqueue_solve(cfs_problem *p, const cfs_batch_in *in, const cfs_batch_out *out, hipStream_t s, hipEvent_t *e4)
{
    const int B = in->B, nj = p->d.njoint, nn = p->nn, K = p->d.MAX_O_ITER;
    // Every workgroup gives its spill slot back on every exit path, so the flags are all clear after a solve that ran.  They are
    // cleared at creation and again after a solve whose enqueue failed half way -- not per solve: with the chip held by another
    // solve's fused workgroups even a memset node waits ~0.08 ms for a slot on this stream.
    if (p->pool_dirty) {
        HIPCHK(hipMemsetAsync(p->pool_flag.p, 0, (size_t)p->pool_n * 16 * sizeof(int), s));
        p->pool_dirty = false;
    }
    if (p->prof) HIPCHK(hipEventRecord(e4[0], s));
    if (p->d.mode == CFS_MODE_CFS) {     // unconstrained minimiser -H^{-1} ff (MFMA), constant over the outer loop
        GemvParams g;
        g.B = B; g.nn = nn; g.M = p->Hinv.p; g.X = in->ff; g.Y = p->x0.p; g.scale = -1.0;
        launch_batched_gemv(g, s);
    }
    if (p->prof) HIPCHK(hipEventRecord(e4[1], s));
    FusedParams fp;
    fill_fused_family(p, fp, B);
    fp.noise_rows = in->noise ? in->noise_rows : 0;
    fp.x_init = in->x_init; fp.xR1 = in->xR1; fp.ff = in->ff; fp.caug = in->caug; fp.obs = in->obs; fp.noise = in->noise;
    fp.u = out->u; fp.x_ = out->x_; fp.cost_all = out->cost_all; fp.e_cost_all = out->e_cost_all; fp.e_u_all = out->e_u_all;
    fp.iter_O = out->iter_O; fp.total_iter = out->total_iter; fp.status = out->status;
    fp.u_hist = (p->d.mode == CFS_MODE_CFS && K > 0) ? p->u_hist.p : nullptr;
    // Launch order.  Workgroups are dispatched in blockIdx order and a launch ends with its longest problem, so problems
    // that will run long active sets should not wait for a free compute unit behind short ones.  Automatic: a pre-pass
    // counts the clearances the initial trajectory violates (two small kernels on the same stream, ~10 us) -- only when the
    // batch cannot start all at once anyway.
    const int nline = p->d.nobs - p->nmesh;
    if (p->order_mode == 1 && p->order_n == B) fp.order = p->order_user.p;
    else if (p->order_mode != 2 && !(p->dbg_mask & CFS_DBG_NO_AUTO_ORDER) && B > p->n_cu && nline > 0) {   // also when a given order is for another batch size
        OrderParams op;
        op.rb = p->rb.p; op.B = B; op.H = p->d.H; op.nj = nj; op.nobs = nline; op.obs_stride = p->d.nobs;
        op.x_init = in->x_init; op.obs = in->obs; op.margin = p->margin.p; op.key = p->okey.p; op.order = p->order.p;
        launch_order(op, s);
        fp.order = p->order.p;
    }
    if (p->prof) HIPCHK(hipEventRecord(e4[2], s));   // after the launch-order pre-pass: [e4[2], e4[3]] brackets the fused kernel alone (mesh handles: the loop of launches)
    if (p->nmesh == 0) {
        HIPCHK(launch_fused(nj, fp, s, force_w1(p)));
    } else {
        // Mesh obstacles are linearised by their own kernel (hierarchy traversals do not fit the fused kernel's register
        // budget), which needs the current iterate: one outer iteration per launch, state carried through HBM.  Every
        // launch is enqueued up front; finished problems return at once, so there is still no host round trip.
        LinMeshParams lm;
        lm.rb = p->rb.p; lm.B = B; lm.H = p->d.H; lm.nmesh = p->nmesh; lm.meshes = p->meshes_d.p;
        lm.dist = p->dist.p; lm.grad = p->grad.p;
        lm.ends = p->m_ends.p; lm.base_d = p->m_base.p; lm.upper_d = p->m_upper.p; lm.base_t = p->m_tri.p; lm.shift_d = p->m_shift.p; lm.near = p->m_near.p; lm.piece_d = p->m_pd.p; lm.piece_i = p->m_pi.p; lm.piece_nd = p->m_pnd.p;
        fp.nmesh = p->nmesh; fp.ext_dist = p->dist.p; fp.ext_grad = p->grad.p; fp.max_launch_iters = 1;
        fp.st_qu = p->qu.p; fp.st_cost = p->st_cost.p; fp.st_noise = p->noise_row.p; fp.st_done = p->st_done.p;
        for (int it = 0; it < std::max(K, 1); ++it) {
            lm.x_ = it == 0 ? in->x_init : out->x_;
            lm.status_done = it == 0 ? nullptr : p->st_done.p;
            lm.seed_prev = it > 0;
            HIPCHK(launch_linearize_mesh(nj, lm, s));
            fp.resume = it > 0;
            HIPCHK(launch_fused(nj, fp, s, force_w1(p)));
        }
    }
    if (p->prof) HIPCHK(hipEventRecord(e4[3], s));
    if (fp.u_hist) {                     // CFS cost history: QQ * (all logged u) on the matrix cores, then the dots
        if (p->u_hist.n) {
            GemvParams g;
            g.B = B * K; g.nn = nn; g.M = p->QQ.p; g.X = p->u_hist.p; g.Y = p->qu_hist.p; g.scale = 1.0;
            launch_batched_gemv(g, s);
            CostHistParams ch;
            ch.B = B; ch.nn = nn; ch.max_o_iter = K; ch.u_hist = p->u_hist.p; ch.qu_hist = p->qu_hist.p;
            ch.ff = in->ff; ch.caug = in->caug; ch.iter_O = out->iter_O; ch.cost_all = out->cost_all; ch.e_cost_all = out->e_cost_all;
            launch_cost_history(ch, s);
        }
    }
    HIPCHK(hipGetLastError());
    return CFS_SUCCESS;
}

int cfs_solve_batch(cfs_problem *p, const cfs_batch_in *in, const cfs_batch_out *out)
{
    if (!p || !in || !out) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    const int B = in->B;
    if (B < 1 || B > p->d.max_batch) return fail(CFS_ERR_INVALID_ARG, "B=%d outside 1..max_batch=%d", B, p->d.max_batch);
    if (!in->x_init || !in->xR1 || !in->ff || !in->caug || !in->obs) return fail(CFS_ERR_INVALID_ARG, "NULL input array");
    HIPCHK(hipSetDevice(p->device));
    const size_t nn = p->nn, nx = p->nx, ns = p->ns, K = p->d.MAX_O_ITER, nobs = p->d.nobs;
    Stage st;
    cfs_batch_in din = *in;
    din.x_init = st.up(in->x_init, B * nx);
    din.xR1 = st.up(in->xR1, B * ns);
    din.ff = st.up(in->ff, B * nn);
    din.caug = st.up(in->caug, B);
    din.obs = st.up(in->obs, B * nobs * 6);
    din.noise = in->noise ? st.up(in->noise, (size_t)B * in->noise_rows * nn) : nullptr;
    cfs_batch_out dout;
    dout.u = st.up<double>(nullptr, B * nn);
    dout.x_ = st.up<double>(nullptr, B * nx);
    dout.cost_all = st.up<double>(nullptr, B * K);
    dout.e_cost_all = st.up<double>(nullptr, B * K);
    dout.e_u_all = st.up<double>(nullptr, B * K);
    dout.iter_O = st.up<int>(nullptr, B);
    dout.total_iter = st.up<int>(nullptr, B);
    dout.status = st.up<int>(nullptr, B);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "staging failed: %s", hipGetErrorString(st.err));
    HIPCHK(hipMemset(dout.cost_all, 0, B * K * sizeof(double)));
    HIPCHK(hipMemset(dout.e_cost_all, 0, B * K * sizeof(double)));
    HIPCHK(hipMemset(dout.e_u_all, 0, B * K * sizeof(double)));
    int rc = cfs_solve_batch_device(p, &din, &dout, nullptr);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    st.down(out->u, dout.u, B * nn);
    st.down(out->x_, dout.x_, B * nx);
    st.down(out->cost_all, dout.cost_all, B * K);
    st.down(out->e_cost_all, dout.e_cost_all, B * K);
    st.down(out->e_u_all, dout.e_u_all, B * K);
    st.down(out->iter_O, dout.iter_O, B);
    st.down(out->total_iter, dout.total_iter, B);
    st.down(out->status, dout.status, B);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "copy back failed: %s", hipGetErrorString(st.err));
    return CFS_SUCCESS;
}

// ---- EVAL.get_Cost_b (Lib/EVAL.m:75-78; main_FANUC.m:131-132) ----------------------------------------------------------
// u_b = quadprog(Qaug, paug) with no constraints = -H^{-1} ff (H = QQ symmetrised, as quadprog does), cost = get_cost(u_b):
// two products on the matrix cores (the first is the one every CFS solve starts with) and one dot per problem.
int cfs_cost_b(cfs_problem *p, int B, const double *ff, const double *caug, double *cost_b, double *u_b)
{
    if (!p || !ff || !caug || !cost_b) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    if (B < 1 || B > p->d.max_batch) return fail(CFS_ERR_INVALID_ARG, "B=%d outside 1..max_batch=%d", B, p->d.max_batch);
    HIPCHK(hipSetDevice(p->device));
    const size_t nn = p->nn;
    Stage st;
    const double *d_ff = st.up(ff, B * nn), *d_caug = st.up(caug, (size_t)B);
    double *d_cost = st.up<double>(nullptr, (size_t)B);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "staging failed: %s", hipGetErrorString(st.err));
    GemvParams g;
    g.B = B; g.nn = (int)nn; g.M = p->Hinv.p; g.X = d_ff; g.Y = p->x0.p; g.scale = -1.0;
    launch_batched_gemv(g, nullptr);
    g.M = p->QQ.p; g.X = p->x0.p; g.Y = p->qu.p; g.scale = 1.0;
    launch_batched_gemv(g, nullptr);
    CostHistParams ch;
    memset(&ch, 0, sizeof ch);
    ch.B = B; ch.nn = (int)nn; ch.max_o_iter = 1; ch.u_hist = p->x0.p; ch.qu_hist = p->qu.p; ch.ff = d_ff; ch.caug = d_caug;
    ch.iter_O = nullptr; ch.cost_all = d_cost; ch.e_cost_all = nullptr;      // iter_O == NULL: one logged iterate per problem
    launch_cost_history(ch, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    st.down(cost_b, d_cost, (size_t)B);
    if (u_b) st.down(u_b, p->x0.p, B * nn);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "copy back failed: %s", hipGetErrorString(st.err));
    return CFS_SUCCESS;
}

// cost = self.eval.get_cost(u) (Lib/EVAL.m:51-53) for B given u: QQ*u on the matrix cores, one dot per problem
int cfs_get_cost(cfs_problem *p, int B, const double *u, const double *ff, const double *caug, double *cost)
{
    if (!p || !u || !ff || !caug || !cost) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    if (B < 1 || B > p->d.max_batch) return fail(CFS_ERR_INVALID_ARG, "B=%d outside 1..max_batch=%d", B, p->d.max_batch);
    HIPCHK(hipSetDevice(p->device));
    const size_t nn = p->nn;
    Stage st;
    const double *d_u = st.up(u, B * nn), *d_ff = st.up(ff, B * nn), *d_caug = st.up(caug, (size_t)B);
    double *d_cost = st.up<double>(nullptr, (size_t)B);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "staging failed: %s", hipGetErrorString(st.err));
    GemvParams g;
    g.B = B; g.nn = (int)nn; g.M = p->QQ.p; g.X = d_u; g.Y = p->qu.p; g.scale = 1.0;
    launch_batched_gemv(g, nullptr);
    CostHistParams ch;
    memset(&ch, 0, sizeof ch);
    ch.B = B; ch.nn = (int)nn; ch.max_o_iter = 1; ch.u_hist = d_u; ch.qu_hist = p->qu.p; ch.ff = d_ff; ch.caug = d_caug; ch.cost_all = d_cost;
    launch_cost_history(ch, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    st.down(cost, d_cost, (size_t)B);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "copy back failed: %s", hipGetErrorString(st.err));
    return CFS_SUCCESS;
}

// ---- developer / test entry points (declared in include/cfs_hip.h; per handle) --------------------------------------------
int cfs_debug_set_options(cfs_problem *p, int mask, int warm_max, double polish_tol)
{
    if (!p) return fail(CFS_ERR_INVALID_ARG, "NULL handle");
    const int known = CFS_DBG_GATHER_ROLLOUTS | CFS_DBG_NO_REFINE | CFS_DBG_NO_WARM_START | CFS_DBG_NO_CERTIFICATE | CFS_DBG_NO_PRUNE |
                      CFS_DBG_NO_AUTO_ORDER | CFS_DBG_TIER_W1;
    if (mask & ~known) return fail(CFS_ERR_INVALID_ARG, "unknown option bits 0x%x", mask & ~known);
    if (warm_max < 0 || warm_max > 64) return fail(CFS_ERR_INVALID_ARG, "warm_max %d outside 0..64", warm_max);
    p->dbg_mask = mask; p->dbg_warm_max = warm_max;
    p->dbg_polish_tol = polish_tol > 0.0 ? polish_tol : 1e-11;
    return CFS_SUCCESS;
}

int cfs_debug_stamps(cfs_problem *p, int B, unsigned long long *out)
{
    if (!p) return fail(CFS_ERR_INVALID_ARG, "NULL handle");
    HIPCHK(hipSetDevice(p->device));
    if (out) {
        if (!p->stamps.p) return fail(CFS_ERR_INVALID_ARG, "stamps are not enabled");
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(out, p->stamps.p, (size_t)p->stamps_B * 12 * 8, hipMemcpyDeviceToHost));
        return CFS_SUCCESS;
    }
    HIPCHK(hipDeviceSynchronize());
    p->stamps.release(); p->stamps_B = 0;
    if (B <= 0) return CFS_SUCCESS;
    if (B > p->d.max_batch) return fail(CFS_ERR_INVALID_ARG, "B=%d exceeds max_batch=%d", B, p->d.max_batch);
    HIPCHK(p->stamps.alloc((size_t)B * 12));
    HIPCHK(hipMemset(p->stamps.p, 0, (size_t)B * 12 * 8));
    p->stamps_B = B;
    return CFS_SUCCESS;
}

int cfs_debug_trace_begin(cfs_problem *p, int b, int cap)
{
    if (!p) return fail(CFS_ERR_INVALID_ARG, "NULL handle");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipDeviceSynchronize());
    p->trace.release(); p->trace_b = -1; p->trace_cap = 0;
    if (cap <= 0) return CFS_SUCCESS;
    HIPCHK(p->trace.alloc((size_t)(cap + 1) * 8));
    HIPCHK(hipMemset(p->trace.p, 0, (size_t)(cap + 1) * 8 * sizeof(double)));
    p->trace_b = b; p->trace_cap = cap;
    return CFS_SUCCESS;
}

int cfs_debug_trace_read(cfs_problem *p, double *out)
{
    if (!p || !out) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    if (!p->trace.p) return fail(CFS_ERR_INVALID_ARG, "no trace was begun");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, p->trace.p, (size_t)(p->trace_cap + 1) * 8 * sizeof(double), hipMemcpyDeviceToHost));
    return CFS_SUCCESS;
}

int cfs_debug_log_u(cfs_problem *p, int on)
{
    if (!p) return fail(CFS_ERR_INVALID_ARG, "NULL handle");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipDeviceSynchronize());
    p->u_log.release();
    if (!on || p->d.MAX_O_ITER < 1) return CFS_SUCCESS;
    const size_t n = (size_t)p->d.max_batch * p->d.MAX_O_ITER * p->nn;
    HIPCHK(p->u_log.alloc(n));
    HIPCHK(hipMemset(p->u_log.p, 0, n * sizeof(double)));
    return CFS_SUCCESS;
}

int cfs_debug_read_u_log(cfs_problem *p, int B, double *out)
{
    if (!p || !out) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    if (!p->u_log.p) return fail(CFS_ERR_INVALID_ARG, "the u log is not enabled");
    if (B < 1 || B > p->d.max_batch) return fail(CFS_ERR_INVALID_ARG, "B=%d outside 1..max_batch=%d", B, p->d.max_batch);
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, p->u_log.p, (size_t)B * p->d.MAX_O_ITER * p->nn * sizeof(double), hipMemcpyDeviceToHost));
    return CFS_SUCCESS;
}

int cfs_set_state_cost(cfs_problem *p, const double *Qaug)
{
    if (!p || !Qaug) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    HIPCHK(hipSetDevice(p->device));
    const int H = p->d.H, nj = p->d.njoint, ns = 2 * nj, nn = p->nn, nx = p->nx;
    const double dt = p->d.robot.delta_t;
    // E = [Aaug(:,1:nj), -G] (nx x 2nj): state error per unit of x0 / xg;  QE = Qaug*E;  F = Baug'*QE;  Cq = E'*QE
    std::vector<long double> E((size_t)nx * 2 * nj, 0.0L), QE((size_t)nx * 2 * nj, 0.0L);
    for (int i = 0; i < H; ++i)
        for (int c = 0; c < nj; ++c) {
            E[(i * ns + c) + (size_t)c * nx] = 1.0L;                      // A^i(1:nj,1:nj) = I  (x0 has zero velocity)
            E[(i * ns + c) + (size_t)(nj + c) * nx] = -1.0L;              // -gaug
        }
    for (int col = 0; col < 2 * nj; ++col)
        for (int r = 0; r < nx; ++r) {
            long double s = 0.0L;
            for (int k = 0; k < nx; ++k) s += (long double)Qaug[r + (size_t)k * nx] * E[k + (size_t)col * nx];
            QE[r + (size_t)col * nx] = s;
        }
    std::vector<double> F1((size_t)nn * nj), F2((size_t)nn * nj), Cq((size_t)4 * nj * nj);
    for (int col = 0; col < 2 * nj; ++col) {
        for (int k = 0; k < H; ++k)
            for (int c = 0; c < nj; ++c) {                              // row (k,c) of Baug': sum over waypoints i >= k
                long double s = 0.0L;
                for (int i = k; i < H; ++i)
                    s += ((long double)(i - k) + 0.5L) * dt * dt * QE[(i * ns + c) + (size_t)col * nx] + (long double)dt * QE[(i * ns + nj + c) + (size_t)col * nx];
                if (col < nj) F1[(k * nj + c) + (size_t)col * nn] = (double)s;
                else F2[(k * nj + c) + (size_t)(col - nj) * nn] = (double)(-s);   // ff = F1*x0 - F2*xg
            }
        for (int r = 0; r < 2 * nj; ++r) {
            long double s = 0.0L;
            for (int k = 0; k < nx; ++k) s += E[k + (size_t)r * nx] * QE[k + (size_t)col * nx];
            Cq[r + (size_t)col * 2 * nj] = (double)s;
        }
    }
    p->F1.release(); p->F2.release(); p->Cq.release();
    HIPCHK(p->F1.alloc(F1.size())); HIPCHK(p->F2.alloc(F2.size())); HIPCHK(p->Cq.alloc(Cq.size()));
    HIPCHK(hipMemcpy(p->F1.p, F1.data(), F1.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->F2.p, F2.data(), F2.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->Cq.p, Cq.data(), Cq.size() * 8, hipMemcpyHostToDevice));
    return CFS_SUCCESS;
}

int cfs_build_terms_device(cfs_problem *p, int B, const double *x0, const double *xg,
                           double *x_init, double *xR1, double *ff, double *caug, void *stream)
{
    if (!p || !x0 || !xg || !x_init || !xR1 || !ff || !caug) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    if (B < 1) return fail(CFS_ERR_INVALID_ARG, "B must be >= 1");
    if (!p->F1.p) return fail(CFS_ERR_INVALID_ARG, "call cfs_set_state_cost first");
    HIPCHK(hipSetDevice(p->device));
    TermsParams t;
    t.B = B; t.H = p->d.H; t.nj = p->d.njoint; t.F1 = p->F1.p; t.F2 = p->F2.p; t.Cq = p->Cq.p;
    t.x0 = x0; t.xg = xg; t.route = nullptr; t.nwp = 0; t.nwp_b = nullptr; t.nwp_stride = 0; t.dt = p->d.robot.delta_t;
    t.x_init = x_init; t.xR1 = xR1; t.ff = ff; t.caug = caug;
    launch_build_terms(t, reinterpret_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return CFS_SUCCESS;
}

int cfs_build_terms_from_routes_device(cfs_problem *p, int B, const double *routes, int nwp,
                                       double *x_init, double *xR1, double *ff, double *caug, void *stream)
{
    if (!p || !routes || !x_init || !xR1 || !ff || !caug) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    if (B < 1 || nwp < 2) return fail(CFS_ERR_INVALID_ARG, "B >= 1 and at least two route waypoints are needed");
    if (!p->F1.p) return fail(CFS_ERR_INVALID_ARG, "call cfs_set_state_cost first");
    HIPCHK(hipSetDevice(p->device));
    TermsParams t;
    t.B = B; t.H = p->d.H; t.nj = p->d.njoint; t.F1 = p->F1.p; t.F2 = p->F2.p; t.Cq = p->Cq.p;
    t.x0 = nullptr; t.xg = nullptr; t.route = routes; t.nwp = nwp; t.nwp_b = nullptr; t.nwp_stride = 0; t.dt = p->d.robot.delta_t;
    t.x_init = x_init; t.xR1 = xR1; t.ff = ff; t.caug = caug;
    launch_build_terms(t, reinterpret_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return CFS_SUCCESS;
}

int cfs_build_terms_from_ragged_routes_device(cfs_problem *p, int B, const double *routes, int nwp_stride, const int *nwp,
                                              double *x_init, double *xR1, double *ff, double *caug, void *stream)
{
    if (!p || !routes || !nwp || !x_init || !xR1 || !ff || !caug) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    if (B < 1 || nwp_stride < 1) return fail(CFS_ERR_INVALID_ARG, "B >= 1 and nwp_stride >= 1 are needed");
    if (!p->F1.p) return fail(CFS_ERR_INVALID_ARG, "call cfs_set_state_cost first");
    HIPCHK(hipSetDevice(p->device));
    TermsParams t;
    t.B = B; t.H = p->d.H; t.nj = p->d.njoint; t.F1 = p->F1.p; t.F2 = p->F2.p; t.Cq = p->Cq.p;
    t.x0 = nullptr; t.xg = nullptr; t.route = routes; t.nwp = nwp_stride; t.nwp_b = nwp; t.nwp_stride = nwp_stride; t.dt = p->d.robot.delta_t;
    t.x_init = x_init; t.xR1 = xR1; t.ff = ff; t.caug = caug;
    launch_build_terms(t, reinterpret_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return CFS_SUCCESS;
}

int cfs_chomp_batch(cfs_problem *p, const cfs_batch_in *in, const double *u0, const double *D, const double *epsilon,
                    const cfs_batch_out *out)
{
    if (!p || !in || !out || !u0 || !D || !epsilon) return fail(CFS_ERR_INVALID_ARG, "NULL argument");
    const int B = in->B;
    if (B < 1 || B > p->d.max_batch) return fail(CFS_ERR_INVALID_ARG, "B=%d outside 1..max_batch=%d", B, p->d.max_batch);
    if (!in->x_init || !in->xR1 || !in->ff || !in->caug || !in->obs) return fail(CFS_ERR_INVALID_ARG, "NULL input array");
    if (!out->u || !out->x_ || !out->cost_all || !out->e_cost_all || !out->e_u_all || !out->iter_O) return fail(CFS_ERR_INVALID_ARG, "NULL output array");
    if (p->nmesh > 0) return fail(CFS_ERR_INVALID_ARG, "CHOMP_FANUC measures line obstacles only (Lib/CHOMP_FANUC.m:119)");
    if (!chomp_fits(p->d.njoint, p->d.H, p->d.nobs)) return fail(CFS_ERR_INVALID_ARG, "H x nobs too large for the CHOMP kernel's 64 KB of LDS");
    HIPCHK(hipSetDevice(p->device));
    const size_t nn = p->nn, nx = p->nx, ns = p->ns, K = p->d.MAX_O_ITER, nobs = p->d.nobs;
    Stage st;
    ChompParams c;
    memset(&c, 0, sizeof c);
    c.rb = p->rb.p; c.B = B; c.H = p->d.H; c.nobs = p->d.nobs; c.max_o_iter = p->d.MAX_O_ITER;
    c.dt = p->d.robot.delta_t; c.alpha = p->d.alpha; c.epsilon_O = p->d.epsilon_O; c.QQ = p->QQ.p;
    c.x_init = st.up(in->x_init, B * nx); c.xR1 = st.up(in->xR1, B * ns); c.ff = st.up(in->ff, B * nn); c.caug = st.up(in->caug, B);
    c.obs = st.up(in->obs, B * nobs * 6); c.u0 = st.up(u0, B * nn); c.D = st.up(D, nobs); c.eps = st.up(epsilon, nobs);
    c.u = st.up<double>(nullptr, B * nn); c.x_ = st.up<double>(nullptr, B * nx);
    c.cost_all = st.up<double>(nullptr, B * std::max<size_t>(K, 1)); c.e_cost_all = st.up<double>(nullptr, B * std::max<size_t>(K, 1));
    c.e_u_all = st.up<double>(nullptr, B * std::max<size_t>(K, 1));
    c.iter_O = st.up<int>(nullptr, B); c.total_iter = st.up<int>(nullptr, B); c.status = st.up<int>(nullptr, B);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "staging failed: %s", hipGetErrorString(st.err));
    HIPCHK(hipMemset(c.cost_all, 0, B * std::max<size_t>(K, 1) * 8)); HIPCHK(hipMemset(c.e_cost_all, 0, B * std::max<size_t>(K, 1) * 8));
    HIPCHK(hipMemset(c.e_u_all, 0, B * std::max<size_t>(K, 1) * 8));
    chomp_derivest_tables(c);
    HIPCHK(launch_chomp(p->d.njoint, c, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    st.down(out->u, c.u, B * nn); st.down(out->x_, c.x_, B * nx);
    st.down(out->cost_all, c.cost_all, B * K); st.down(out->e_cost_all, c.e_cost_all, B * K); st.down(out->e_u_all, c.e_u_all, B * K);
    st.down(out->iter_O, c.iter_O, B); st.down(out->total_iter, c.total_iter, B); st.down(out->status, c.status, B);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "copy back failed: %s", hipGetErrorString(st.err));
    return CFS_SUCCESS;
}

int cfs_problem_set_meshes(cfs_problem *p, int nmesh, const cfs_mesh *const *meshes)
{
    if (!p) return fail(CFS_ERR_INVALID_ARG, "NULL handle");
    if (nmesh < 0 || nmesh > p->d.nobs) return fail(CFS_ERR_INVALID_ARG, "nmesh %d outside 0..nobs=%d", nmesh, p->d.nobs);
    if (nmesh > 0 && !meshes) return fail(CFS_ERR_INVALID_ARG, "meshes is NULL");
    HIPCHK(hipSetDevice(p->device));
    std::vector<DevMesh> v(nmesh);
    for (int i = 0; i < nmesh; ++i) {
        if (!meshes[i]) return fail(CFS_ERR_INVALID_ARG, "meshes[%d] is NULL", i);
        if (meshes[i]->device != p->device) return fail(CFS_ERR_INVALID_ARG, "meshes[%d] lives on device %d, the problem on %d", i, meshes[i]->device, p->device);
        v[i] = meshes[i]->view();
    }
    p->meshes_d.release(); p->st_cost.release(); p->st_done.release();
    p->m_ends.release(); p->m_base.release(); p->m_shift.release(); p->m_tri.release(); p->m_near.release(); p->m_upper.release();
    p->m_pd.release(); p->m_pnd.release(); p->m_pi.release();
    p->nmesh = 0;
    if (nmesh > 0) {
        size_t we, wb, ws, wn, wpd, wpi, wpn;
        linearize_mesh_workspace(p->d.njoint, nmesh, &we, &wb, &ws, &wn, &wpd, &wpi, &wpn);
        const size_t bh = (size_t)p->d.max_batch * p->d.H;
        HIPCHK(p->m_ends.alloc(bh * we)); HIPCHK(p->m_base.alloc(bh * wb)); HIPCHK(p->m_upper.alloc(bh * wb)); HIPCHK(p->m_tri.alloc(bh * wb)); HIPCHK(p->m_shift.alloc(bh * ws)); HIPCHK(p->m_near.alloc(bh * wn));
        HIPCHK(p->m_pd.alloc(bh * wpd)); HIPCHK(p->m_pi.alloc(bh * wpi)); HIPCHK(p->m_pnd.alloc(bh * wpn));
        HIPCHK(p->meshes_d.alloc(nmesh));
        HIPCHK(p->st_cost.alloc(2 * (size_t)p->d.max_batch));
        HIPCHK(p->st_done.alloc(p->d.max_batch));
        HIPCHK(hipMemcpy(p->meshes_d.p, v.data(), sizeof(DevMesh) * nmesh, hipMemcpyHostToDevice));
        p->nmesh = nmesh;
    }
    return CFS_SUCCESS;
}

int cfs_profile_enable(cfs_problem *p, int on)
{
    if (!p) return fail(CFS_ERR_INVALID_ARG, "NULL handle");
    p->prof = on != 0;
    return CFS_SUCCESS;
}

int cfs_profile_read(cfs_problem *p, double *solve_kernel_ms, double *gemm_kernel_ms, int *solves)
{
    if (!p) return fail(CFS_ERR_INVALID_ARG, "NULL handle");
    HIPCHK(hipSetDevice(p->device));
    double fused = 0.0, gemm = 0.0;
    const int n = (int)(p->ev.size() / 4);
    for (int k = 0; k < n; ++k) {
        float ms = 0.f;
        HIPCHK(hipEventSynchronize(p->ev[4 * k + 3]));
        HIPCHK(hipEventElapsedTime(&ms, p->ev[4 * k + 0], p->ev[4 * k + 1]));
        gemm += ms;
        HIPCHK(hipEventElapsedTime(&ms, p->ev[4 * k + 2], p->ev[4 * k + 3]));
        fused += ms;
    }
    for (hipEvent_t e : p->ev) p->ev_free.push_back(e);
    p->ev.clear();
    if (solve_kernel_ms) *solve_kernel_ms = fused;
    if (gemm_kernel_ms) *gemm_kernel_ms = gemm;
    if (solves) *solves = n;
    return CFS_SUCCESS;
}

int cfs_dist_arm(const cfs_robot *robot, int njoint, int N, const double *theta, int nobs, const double *obs,
                 double *d, int *linkid, double *pos)
{
    int rc = check_robot(robot, njoint);
    if (rc) return rc;
    if (N < 0 || nobs < 0 || !theta || !obs || !d) return fail(CFS_ERR_INVALID_ARG, "bad argument");
    if (N == 0 || nobs == 0) return CFS_SUCCESS;
    if (have_device() == 0) return fail(CFS_ERR_NO_DEVICE, "no HIP device visible");
    HIPCHK(hipSetDevice(g_device));
    DevRobot hr;
    build_dev_robot(*robot, hr);
    Stage st;
    DistArmParams P;
    P.rb = st.up(&hr, 1);
    P.N = N; P.nobs = nobs; P.nj = njoint;
    P.theta = st.up(theta, (size_t)N * njoint);
    P.obs = st.up(obs, (size_t)nobs * 6);
    P.d = st.up<double>(nullptr, (size_t)N * nobs);
    P.linkid = st.up<int>(nullptr, (size_t)N * nobs);
    P.pos = pos ? st.up<double>(nullptr, (size_t)N * njoint * 6) : nullptr;
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "staging failed: %s", hipGetErrorString(st.err));
    launch_dist_arm(P, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    st.down(d, P.d, (size_t)N * nobs);
    st.down(linkid, P.linkid, (size_t)N * nobs);
    if (pos) st.down(pos, P.pos, (size_t)N * njoint * 6);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "copy back failed: %s", hipGetErrorString(st.err));
    return CFS_SUCCESS;
}

static int check_batch(const cfs_problem *p, int B)
{
    if (!p) return fail(CFS_ERR_INVALID_ARG, "NULL handle");
    if (B < 1 || B > p->d.max_batch) return fail(CFS_ERR_INVALID_ARG, "B=%d outside 1..max_batch=%d", B, p->d.max_batch);
    return CFS_SUCCESS;
}

// The pieces below run the SAME kernel as the whole solve (cfs_solve_fused_kernel, FusedParams::piece), so that the
// kernel-level parity tests certify the code that is benchmarked.

// stage what the fused kernel's constructor reads; arrays the piece does not use are zero-filled
struct PieceBuffers {
    double *x_, *xR1, *ff, *caug, *obs;
};
static int stage_piece(cfs_problem *p, Stage &st, int B, const double *x_, const double *xR1, const double *obs, PieceBuffers &pb)
{
    const size_t nobs = p->d.nobs;
    pb.x_ = st.up(x_, (size_t)B * p->nx);
    pb.xR1 = st.up(xR1, (size_t)B * p->ns);
    pb.ff = st.up<double>(nullptr, (size_t)B * p->nn);
    pb.caug = st.up<double>(nullptr, (size_t)B);
    pb.obs = st.up(obs, (size_t)B * nobs * 6);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "staging failed: %s", hipGetErrorString(st.err));
    if (!x_) HIPCHK(hipMemset(pb.x_, 0, (size_t)B * p->nx * 8));
    if (!xR1) HIPCHK(hipMemset(pb.xR1, 0, (size_t)B * p->ns * 8));
    if (!obs) HIPCHK(hipMemset(pb.obs, 0, (size_t)B * nobs * 6 * 8));
    HIPCHK(hipMemset(pb.ff, 0, (size_t)B * p->nn * 8));
    HIPCHK(hipMemset(pb.caug, 0, (size_t)B * 8));
    return CFS_SUCCESS;
}

// linearisation of B trajectories into p->dist / p->grad (B x nobs x H [x nj]) and, optionally, linkid
static int linearize_piece(cfs_problem *p, int B, const PieceBuffers &pb, double *d_dist, double *d_grad, int *d_linkid)
{
    const int nj = p->d.njoint;
    FusedParams fp;
    fill_fused_family(p, fp, B);
    fp.x_init = pb.x_; fp.xR1 = pb.xR1; fp.ff = pb.ff; fp.caug = pb.caug; fp.obs = pb.obs;
    fp.piece = 1;
    fp.dump_dist = d_dist; fp.dump_grad = d_grad; fp.dump_linkid = d_linkid;
    if (d_linkid) HIPCHK(hipMemsetAsync(d_linkid, 0, (size_t)B * p->d.nobs * p->d.H * sizeof(int), nullptr));
    if (p->nmesh > 0) {          // rows of the mesh obstacles come from the hierarchy kernels, as in the whole solve
        LinMeshParams lm;
        lm.rb = p->rb.p; lm.B = B; lm.H = p->d.H; lm.nmesh = p->nmesh; lm.meshes = p->meshes_d.p;
        lm.dist = p->dist.p; lm.grad = p->grad.p;
        lm.ends = p->m_ends.p; lm.base_d = p->m_base.p; lm.upper_d = p->m_upper.p; lm.base_t = p->m_tri.p; lm.shift_d = p->m_shift.p; lm.near = p->m_near.p; lm.piece_d = p->m_pd.p; lm.piece_i = p->m_pi.p; lm.piece_nd = p->m_pnd.p;
        lm.x_ = pb.x_; lm.status_done = nullptr; lm.seed_prev = 0;
        HIPCHK(launch_linearize_mesh(nj, lm, nullptr));
        fp.nmesh = p->nmesh; fp.ext_dist = p->dist.p; fp.ext_grad = p->grad.p;
    }
    HIPCHK(launch_fused(nj, fp, nullptr, force_w1(p)));
    return CFS_SUCCESS;
}

int cfs_linearize(cfs_problem *p, int B, const double *x_, const double *obs, double *dist, int *linkid, double *grad)
{
    int rc = check_batch(p, B);
    if (rc) return rc;
    if (!x_ || !obs || !dist || !grad) return fail(CFS_ERR_INVALID_ARG, "NULL array");
    HIPCHK(hipSetDevice(p->device));
    const size_t nobs = p->d.nobs, H = p->d.H, nj = p->d.njoint;
    Stage st;
    PieceBuffers pb;
    rc = stage_piece(p, st, B, x_, nullptr, obs, pb);
    if (rc) return rc;
    double *d_dist = st.up<double>(nullptr, (size_t)B * nobs * H);
    double *d_grad = st.up<double>(nullptr, (size_t)B * nobs * H * nj);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "staging failed: %s", hipGetErrorString(st.err));
    rc = linearize_piece(p, B, pb, d_dist, d_grad, p->linkid.p);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    st.down(dist, d_dist, (size_t)B * nobs * H);
    st.down(linkid, p->linkid.p, (size_t)B * nobs * H);
    st.down(grad, d_grad, (size_t)B * nobs * H * nj);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "copy back failed: %s", hipGetErrorString(st.err));
    return CFS_SUCCESS;
}

int cfs_get_con(cfs_problem *p, int B, const double *x_, const double *u, const double *xR1, const double *obs,
                double *Ainq, double *binq)
{
    int rc = check_batch(p, B);
    if (rc) return rc;
    if (!x_ || !u || !xR1 || !obs || !Ainq || !binq) return fail(CFS_ERR_INVALID_ARG, "NULL array");
    HIPCHK(hipSetDevice(p->device));
    const size_t nobs = p->d.nobs, H = p->d.H, nj = p->d.njoint, nn = p->nn;
    const size_t rows = nobs * H * (1 + 2 * nj);
    Stage st;
    PieceBuffers pb;
    rc = stage_piece(p, st, B, x_, xR1, obs, pb);
    if (rc) return rc;
    DenseConParams dc;
    dc.B = B; dc.H = p->d.H; dc.nj = p->d.njoint; dc.nobs = p->d.nobs; dc.dt = p->d.robot.delta_t;
    double *d_dist = st.up<double>(nullptr, (size_t)B * nobs * H);
    double *d_grad = st.up<double>(nullptr, (size_t)B * nobs * H * nj);
    dc.dist = d_dist; dc.grad = d_grad;
    dc.u = st.up(u, (size_t)B * nn);
    dc.xR1 = pb.xR1;
    dc.lim = p->lim.p; dc.margin = p->margin.p;
    dc.Ainq = st.up<double>(nullptr, (size_t)B * rows * nn);
    dc.binq = st.up<double>(nullptr, (size_t)B * rows);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "staging failed: %s", hipGetErrorString(st.err));
    rc = linearize_piece(p, B, pb, d_dist, d_grad, nullptr);
    if (rc) return rc;
    launch_dense_con(dc, nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    st.down(Ainq, dc.Ainq, (size_t)B * rows * nn);
    st.down(binq, dc.binq, (size_t)B * rows);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "copy back failed: %s", hipGetErrorString(st.err));
    return CFS_SUCCESS;
}

int cfs_qp(cfs_problem *p, int B, const double *lin, const double *u_lin, const double *xR1,
           const double *dist, const double *grad, double *u, double *lambda, int *qp_iter, int *status)
{
    int rc = check_batch(p, B);
    if (rc) return rc;
    if (!lin || !u_lin || !xR1 || !dist || !grad || !u) return fail(CFS_ERR_INVALID_ARG, "NULL array");
    HIPCHK(hipSetDevice(p->device));
    const size_t nobs = p->d.nobs, H = p->d.H, nj = p->d.njoint, nn = p->nn;
    const size_t nlam = nobs * H + 4 * nn;
    Stage st;
    PieceBuffers pb;
    rc = stage_piece(p, st, B, nullptr, xR1, nullptr, pb);
    if (rc) return rc;
    double *d_lin = st.up(lin, (size_t)B * nn);
    double *d_u = st.up(u_lin, (size_t)B * nn);
    double *d_dist = st.up(dist, (size_t)B * nobs * H);
    double *d_grad = st.up(grad, (size_t)B * nobs * H * nj);
    double *d_lam = lambda ? st.up<double>(nullptr, (size_t)B * nlam) : nullptr;
    int *d_it = st.up<int>(nullptr, (size_t)B), *d_st = st.up<int>(nullptr, (size_t)B);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "staging failed: %s", hipGetErrorString(st.err));
    FusedParams fp;
    fill_fused_family(p, fp, B);
    fp.x_init = pb.x_; fp.xR1 = pb.xR1; fp.ff = pb.ff; fp.caug = pb.caug; fp.obs = pb.obs;
    fp.piece = 2;
    fp.nmesh = p->d.nobs; fp.ext_dist = d_dist; fp.ext_grad = d_grad;     // every row of the linearisation is given
    fp.u = d_u; fp.total_iter = d_it; fp.status = d_st; fp.dump_lambda = d_lam;
    if (p->d.mode == CFS_MODE_CFS) {      // start point: the unconstrained minimiser -H^{-1} ff (CFS) | u_ itself (projection)
        GemvParams g;
        g.B = B; g.nn = (int)nn; g.M = p->Hinv.p; g.X = d_lin; g.Y = p->x0.p; g.scale = -1.0;
        launch_batched_gemv(g, nullptr);
        fp.x0 = p->x0.p;
    } else fp.x0 = d_lin;
    HIPCHK(launch_fused(p->d.njoint, fp, nullptr, force_w1(p)));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    st.down(u, d_u, (size_t)B * nn);
    if (lambda) st.down(lambda, d_lam, (size_t)B * nlam);
    st.down(qp_iter, d_it, (size_t)B);
    st.down(status, d_st, (size_t)B);
    if (st.err != hipSuccess) return fail(CFS_ERR_HIP, "copy back failed: %s", hipGetErrorString(st.err));
    return CFS_SUCCESS;
}

}  // extern "C"
