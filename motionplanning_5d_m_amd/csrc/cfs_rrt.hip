// cfs_rrt.hip -- RRT / RRT* tree growth in joint space, one wavefront per tree, whole trees on the device (row f1).
//
// Reference behaviour restated (not translated): Lib/RRT_FANUC.m
//   find_route   :63-91    tree = [x0]; until goal_reached: getNode, addNode, [arrangeNode for RRT*]; back-track the route
//   getRandNode  :106-131  pp = rand; pp < bi: sample = (rand(nstate,1)-0.5).*region_s*2 + sample_off, else sample = goal_th;
//                          nearest node under the `ratial`-weighted 2-norm (first minimum wins, strict <, :124);
//                          newNode = near + (sample-near)*0.1/norm(near-sample)  (UNWEIGHTED norm)
//   feasible     :146-181  FK (theta(2) - pi/2 for M200i) + distLinSeg of every link against every obstacle axis, near-zero
//                          surrogate, reject if any distance < obs{j}.D; getNode :95-103 keeps sampling until feasible
//   addNode      :184-190  edge cost = total_dis(parent) + toNode_dis(parent): the parent -> SAMPLE distance (quirk kept)
//   arrangeNode  :134-142  RRT*: nodes within 0.2 of the SAMPLE are re-parented to the new node when that is cheaper; costs
//                          of their descendants are not propagated (quirk kept)
//   goal_reached :193-207  all(goal - region_g < newNode) & all(newNode < goal + region_g); fail when node_num > MAX_ITER
// and Lib/functions/s_Parallel_rrt.m:14-28 (independent seeds; here any number of trees per launch instead of parfor's 6).
//
// MI355X mapping.  A tree is a strictly sequential chain of proposals; trees are independent.  One 64-lane wavefront owns one
// tree for its whole life: the tree (nodes, parents, costs, the per-proposal distances) lives in LDS (<= 28 KB at
// MAX_ITER = 400: several trees per CU), the nearest-neighbour search is a wave argmin over the nodes (lane i takes nodes
// i, i + 64, ...), the feasibility test runs one (obstacle, link) pair per lane on the shared forward-kinematics result, RRT*'s
// re-parenting is one lane per node.  All control flow is wave-uniform (decisions come from ballots / wave reductions), there
// is no block barrier and no host round trip; >= 1024 trees per launch fill the chip.
// Random numbers: MATLAB's rand stream cannot be reproduced, so the caller either passes the uniforms (S x ndraw, consumed
// exactly as the reference consumes rand: one per proposal + nstate more when the sample is random) or a seed for the
// counter-based generator below (splitmix64 of (seed, tree, counter): reproducible on any host in integer arithmetic).
//
// This translation unit is compiled with -ffp-contract=off (Makefile): tree arithmetic (weighted norms, the 0.1-rad extension,
// cost sums) is then plain IEEE mul / add / div / sqrt in the order written here, so that parents, nodes, costs and routes are
// BIT-identical to the CPU restatement (oracle/rrt_oracle.py); only sin / cos differ from libm by <= 1 ulp (all_ee: 1e-14).
#include "cfs_geom_dev.h"
#include "cfs_host.h"
#include <cstring>
#include <vector>

namespace {

constexpr int WV = 64;

__device__ __forceinline__ double rrt_uniform(const RrtParams &P, int tree, unsigned long long counter)
{
    if (P.uniforms) return P.uniforms[(size_t)tree * P.ndraw + counter];
    unsigned long long z = P.seed + (unsigned long long)tree * 0x9E3779B97F4A7C15ull + (counter + 1ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * 0x1.0p-53;
}

template <int NJ>
__global__ __launch_bounds__(WV) void cfs_rrt_kernel(const RrtParams P)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = blockIdx.x, lane = threadIdx.x;
    const int NMAX = P.max_iter + 1;                          // node_num may reach MAX_ITER + 1 before the failure test fires (:201)
    double *s_nodes = lds;                                    // [NMAX][NJ]
    double *s_tot = s_nodes + (size_t)NMAX * NJ;              // total_dis
    double *s_to = s_tot + NMAX;                              // toNode_dis of the current proposal
    int *s_par = reinterpret_cast<int *>(s_to + NMAX);        // parent (1-based, -1 for the root), all_nodes(1,:)
    const DevRobot *rb = &P.rb;                               // robot constants travel in the kernel's parameter block (scalar loads)
    const double *x0 = P.x0 + (P.per_tree ? (size_t)t * NJ : 0), *goal = P.goal + (P.per_tree ? (size_t)t * NJ : 0);
    const double *goal_th = P.goal_th + (P.per_tree ? (size_t)t * NJ : 0);

    double newNode[NJ];
#pragma unroll
    for (int c = 0; c < NJ; ++c) newNode[c] = x0[c];
    if (lane < NJ) s_nodes[lane] = x0[lane];
    if (lane == 0) { s_tot[0] = 0.0; s_par[0] = -1; }
    int node_num = 1, parent = 1, fail = 0;
    unsigned long long cursor = 0, proposals = 0;
    const unsigned long long draw_cap = P.uniforms ? (unsigned long long)P.ndraw : (unsigned long long)P.max_draws;

    auto reached = [&]() {                                    // goal_reached (:193-199): both vector tests must hold for every joint
        bool in = true;
#pragma unroll
        for (int c = 0; c < NJ; ++c) in = in && ((goal[c] - P.region_g[c]) < newNode[c]) && (newNode[c] < (goal[c] + P.region_g[c]));
        return in;
    };
    bool done = reached();
    if (node_num > P.max_iter) { fail = 1; done = true; }
    __builtin_amdgcn_wave_barrier();

    while (!done) {
        // ---- getNode (:95-103): propose until feasible -------------------------------------------------------------------
        double e_first[3] = {0.0, 0.0, 0.0};
        for (;;) {
            if (cursor + 1ull + (unsigned long long)NJ > draw_cap) { fail = 2; break; }      // the random stream is exhausted: every wave gets here
            ++proposals;
            // getRandNode (:106-114)
            const double pp = rrt_uniform(P, t, cursor++);
            double sample[NJ];
            if (pp < P.bi) {
#pragma unroll
                for (int c = 0; c < NJ; ++c) sample[c] = (rrt_uniform(P, t, cursor + c) - 0.5) * P.region_s[c] * 2.0 + P.sample_off[c];
                cursor += NJ;
            } else {
#pragma unroll
                for (int c = 0; c < NJ; ++c) sample[c] = goal_th[c];
            }
            // nearest node under the weighted norm (:116-127): lane i scans nodes i, i + 64, ... in increasing order
            double best = INFINITY;
            int bidx = 0x7fffffff;
            for (int i = lane; i < node_num; i += WV) {
                double s2 = 0.0;
#pragma unroll
                for (int c = 0; c < NJ; ++c) { const double w = (s_nodes[i * NJ + c] - sample[c]) * P.ratial[c]; s2 += w * w; }
                const double dis = sqrt(s2);
                s_to[i] = dis;
                if (dis < best) { best = dis; bidx = i; }
            }
            // wave argmin, first minimum wins (:124 replaces only on a strictly smaller distance): smaller distance, then smaller index
#pragma unroll
            for (int m = 1; m < WV; m <<= 1) {
                const double ob = __shfl_xor(best, m, WV);
                const int oi = __shfl_xor(bidx, m, WV);
                if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // s_to is read across lanes below
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the reference starts from node 1 and `dis_ < dis` is false for NaN: a NaN first distance keeps node 1, NaN elsewhere
            // never wins (NaN arises only after a zero-length extension, sample == nearest node)
            if (bidx == 0x7fffffff || s_to[0] != s_to[0]) bidx = 0;
            parent = bidx + 1;
            double near[NJ], n2 = 0.0;
#pragma unroll
            for (int c = 0; c < NJ; ++c) { near[c] = s_nodes[bidx * NJ + c]; const double w = near[c] - sample[c]; n2 += w * w; }
            const double nrm = sqrt(n2);
#pragma unroll
            for (int c = 0; c < NJ; ++c) newNode[c] = near[c] + (sample[c] - near[c]) * 0.1 / nrm;      // :129, left to right
            // feasible (:146-181): FK once per lane (uniform work), one (obstacle, link) pair per lane
            double ends[NJ * 6], M[12], Mn[12];
#pragma unroll
            for (int k = 0; k < NJ; ++k) {
                double sn, cs;
                sincos(newNode[k] - rb->th_off[k], &sn, &cs);
                fk_step(rb, k, sn, cs, k == 0 ? nullptr : M, Mn);
#pragma unroll
                for (int q = 0; q < 12; ++q) M[q] = Mn[q];
                link_ends(rb, k, M, ends + k * 6);
            }
            bool hit = false;
            for (int p = lane; p < P.nobs * NJ; p += WV) {
                const int j = p / NJ, k = p - j * NJ;
                double o6[6], a6[6];
#pragma unroll
                for (int q = 0; q < 6; ++q) o6[q] = P.obs[j * 6 + q];
#pragma unroll
                for (int kk = 0; kk < NJ; ++kk)
                    if (kk == k) {
#pragma unroll
                        for (int q = 0; q < 6; ++q) a6[q] = ends[kk * 6 + q];
                    }
                const double dis = seg_seg_dist(a6, o6);
                hit = hit || (dis < P.D[j]);
            }
            const bool infeasible = __ballot(hit) != 0ull;
            e_first[0] = ends[(NJ - 1) * 6]; e_first[1] = ends[(NJ - 1) * 6 + 1]; e_first[2] = ends[(NJ - 1) * 6 + 2];   // pos{nstate}.p(:,1) (:186)
            if (!infeasible) break;
        }
        if (fail) break;
        // ---- addNode (:184-190) ------------------------------------------------------------------------------------------
        const int nn_ = node_num;                             // 0-based index of the new node
        if (nn_ >= NMAX) { fail = 1; break; }                // cannot happen: the failure test below fires first
        const double tot_new = s_tot[parent - 1] + s_to[parent - 1];
        __builtin_amdgcn_wave_barrier();
        if (lane < NJ) s_nodes[nn_ * NJ + lane] = newNode[lane];
        if (lane == 0) {
            s_par[nn_] = parent; s_tot[nn_] = tot_new;
            if (P.all_ee) { double *ee = P.all_ee + ((size_t)t * P.max_iter + (nn_ - 1)) * 3; ee[0] = e_first[0]; ee[1] = e_first[1]; ee[2] = e_first[2]; }
        }
        node_num = nn_ + 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- arrangeNode (:134-142), RRT* only: toNode_dis has one entry per node that existed at the proposal ----------------
        if (P.solver == 1) {
            for (int i = lane; i < nn_; i += WV) {
                const double td = s_to[i];
                if (td < P.rewire) {
                    const double via = tot_new + td;
                    if (s_tot[i] > via) { s_par[i] = node_num; s_tot[i] = via; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- goal_reached (:193-207) -------------------------------------------------------------------------------------
        done = reached();
        if (node_num > P.max_iter) { fail = 1; done = true; }
    }

    // ---- results: the tree, then the route by back-tracking (:85-90) --------------------------------------------------------
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < node_num * NJ; e += WV) P.nodes[(size_t)t * NMAX * NJ + e] = s_nodes[e];
    for (int i = lane; i < node_num; i += WV) { P.parent[(size_t)t * NMAX + i] = s_par[i]; P.total_dis[(size_t)t * NMAX + i] = s_tot[i]; }
    // route = [ancestors ..., newNode]; `parent` is the parent chosen by the LAST proposal (for a tree of one node: none)
    int len = 1, p = node_num > 1 ? parent : -1;
    while (p != -1 && len <= node_num) { ++len; p = s_par[p - 1]; }          // length first (uniform); a re-parenting cycle ends at node_num + 1
    if (p != -1) { fail = fail ? fail : 3; len = 1; }                        // RRT* may close a cycle (the reference would never return): report
    if (lane == 0) {
        double *rt = P.route + (size_t)t * NMAX * NJ;
        int pos = len - 1;
#pragma unroll
        for (int c = 0; c < NJ; ++c) rt[pos * NJ + c] = newNode[c];
        p = (node_num > 1 && len > 1) ? parent : -1;
        while (p != -1 && pos > 0) {
            --pos;
            for (int c = 0; c < NJ; ++c) rt[pos * NJ + c] = s_nodes[(p - 1) * NJ + c];
            p = s_par[p - 1];
        }
        P.node_num[t] = node_num; P.fail[t] = fail; P.route_len[t] = len;
        if (P.draws_used) P.draws_used[t] = (long long)cursor;
        if (P.proposals) P.proposals[t] = (long long)proposals;
    }
}

}  // namespace

size_t rrt_lds_bytes(int nj, int max_iter)
{
    const size_t NMAX = (size_t)max_iter + 1;
    return NMAX * nj * 8 + NMAX * 8 * 2 + ((NMAX * 4 + 7) & ~(size_t)7);
}

hipError_t launch_rrt(int nj, const RrtParams &p, hipStream_t s)
{
    const size_t lds = rrt_lds_bytes(nj, p.max_iter);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    switch (nj) {
    case 2: hipLaunchKernelGGL(cfs_rrt_kernel<2>, dim3(p.S), dim3(WV), lds, s, p); break;
    case 3: hipLaunchKernelGGL(cfs_rrt_kernel<3>, dim3(p.S), dim3(WV), lds, s, p); break;
    case 4: hipLaunchKernelGGL(cfs_rrt_kernel<4>, dim3(p.S), dim3(WV), lds, s, p); break;
    case 5: hipLaunchKernelGGL(cfs_rrt_kernel<5>, dim3(p.S), dim3(WV), lds, s, p); break;
    case 6: hipLaunchKernelGGL(cfs_rrt_kernel<6>, dim3(p.S), dim3(WV), lds, s, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- C ABI (include/cfs_hip.h, "RRT / RRT*") -----------------------------------------------------------------------------------
namespace {
int check_rrt(const cfs_rrt_desc *d, int S)
{
    if (!d) return cfs_fail(CFS_ERR_INVALID_ARG, "NULL descriptor");
    int rc = cfs_check_robot(&d->robot, d->nstate);
    if (rc) return rc;
    if (d->nstate < 2) return cfs_fail(CFS_ERR_INVALID_ARG, "nstate %d unsupported (2..6)", d->nstate);
    if (S < 1) return cfs_fail(CFS_ERR_INVALID_ARG, "at least one tree is needed");
    if (d->solver != CFS_RRT && d->solver != CFS_RRT_STAR) return cfs_fail(CFS_ERR_INVALID_ARG, "unknown solver %d", d->solver);
    if (d->max_iter < 1 || rrt_lds_bytes(d->nstate, d->max_iter) > 64 * 1024) return cfs_fail(CFS_ERR_INVALID_ARG, "MAX_ITER %d outside 1..%d", d->max_iter, 1000);
    if (d->nobs < 0 || d->nobs > CFS_MAX_OBS) return cfs_fail(CFS_ERR_INVALID_ARG, "nobs %d outside 0..%d", d->nobs, CFS_MAX_OBS);
    if (!d->x0 || !d->goal || !d->goal_th || !d->region_g || !d->region_s || !d->sample_off || !d->ratial) return cfs_fail(CFS_ERR_INVALID_ARG, "NULL array in the descriptor");
    if (d->nobs > 0 && (!d->obs || !d->D)) return cfs_fail(CFS_ERR_INVALID_ARG, "obs / D must be given");
    if (!d->uniforms && d->max_draws < 1) return cfs_fail(CFS_ERR_INVALID_ARG, "max_draws must be >= 1 in generator mode");
    if (d->uniforms && d->ndraw < 1) return cfs_fail(CFS_ERR_INVALID_ARG, "ndraw must be >= 1");
    return CFS_SUCCESS;
}
}  // namespace

extern "C" int cfs_rrt_grow_device(const cfs_rrt_desc *d, int S, const cfs_rrt_out *out, void *stream)
{
    int rc = check_rrt(d, S);
    if (rc) return rc;
    if (!out || !out->node_num || !out->fail || !out->parent || !out->nodes || !out->total_dis || !out->route_len || !out->route)
        return cfs_fail(CFS_ERR_INVALID_ARG, "NULL output array");
    if (cfs_device_count() <= 0) return cfs_fail(CFS_ERR_NO_DEVICE, "no HIP device visible");
    CFS_HIPCHK(hipSetDevice(cfs_current_device()));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    RrtParams P;
    memset(&P, 0, sizeof P);
    cfs_build_dev_robot(d->robot, P.rb);
    P.S = S; P.nobs = d->nobs; P.solver = d->solver; P.max_iter = d->max_iter; P.per_tree = d->per_tree;
    P.bi = d->bi; P.rewire = d->rewire;
    P.x0 = d->x0; P.goal = d->goal; P.goal_th = d->goal_th; P.region_g = d->region_g; P.region_s = d->region_s; P.sample_off = d->sample_off; P.ratial = d->ratial;
    P.obs = d->obs; P.D = d->D; P.uniforms = d->uniforms; P.ndraw = d->ndraw; P.seed = d->seed; P.max_draws = d->max_draws;
    P.node_num = out->node_num; P.fail = out->fail; P.parent = out->parent; P.route_len = out->route_len;
    P.nodes = out->nodes; P.total_dis = out->total_dis; P.all_ee = out->all_ee; P.route = out->route;
    P.draws_used = out->draws_used; P.proposals = out->proposals;
    hipError_t e = launch_rrt(d->nstate, P, s);
    if (e != hipSuccess) return cfs_fail(CFS_ERR_HIP, "RRT launch failed: %s", hipGetErrorString(e));
    return CFS_SUCCESS;
}

extern "C" int cfs_rrt_grow(const cfs_rrt_desc *d, int S, const cfs_rrt_out *out)
{
    int rc = check_rrt(d, S);
    if (rc) return rc;
    if (!out || !out->node_num || !out->fail || !out->parent || !out->nodes || !out->total_dis || !out->route_len || !out->route)
        return cfs_fail(CFS_ERR_INVALID_ARG, "NULL output array");
    if (cfs_device_count() <= 0) return cfs_fail(CFS_ERR_NO_DEVICE, "no HIP device visible");
    CFS_HIPCHK(hipSetDevice(cfs_current_device()));
    const size_t nj = d->nstate, N = (size_t)d->max_iter + 1, per = d->per_tree ? (size_t)S : 1;
    std::vector<void *> bufs;
    hipError_t err = hipSuccess;
    auto up = [&](const void *h, size_t bytes) -> void * {
        if (err != hipSuccess) return nullptr;
        void *p = nullptr;
        err = hipMalloc(&p, bytes ? bytes : 8);
        if (err != hipSuccess) return nullptr;
        bufs.push_back(p);
        if (h) err = hipMemcpy(p, h, bytes, hipMemcpyHostToDevice);
        return p;
    };
    cfs_rrt_desc dd = *d;
    dd.x0 = (const double *)up(d->x0, per * nj * 8); dd.goal = (const double *)up(d->goal, per * nj * 8); dd.goal_th = (const double *)up(d->goal_th, per * nj * 8);
    dd.region_g = (const double *)up(d->region_g, nj * 8); dd.region_s = (const double *)up(d->region_s, nj * 8);
    dd.sample_off = (const double *)up(d->sample_off, nj * 8); dd.ratial = (const double *)up(d->ratial, nj * 8);
    dd.obs = (const double *)up(d->obs, (size_t)d->nobs * 6 * 8); dd.D = (const double *)up(d->D, (size_t)d->nobs * 8);
    if (d->uniforms) dd.uniforms = (const double *)up(d->uniforms, (size_t)S * d->ndraw * 8);
    cfs_rrt_out o;
    memset(&o, 0, sizeof o);
    o.node_num = (int *)up(nullptr, (size_t)S * 4); o.fail = (int *)up(nullptr, (size_t)S * 4); o.route_len = (int *)up(nullptr, (size_t)S * 4);
    o.parent = (int *)up(nullptr, S * N * 4); o.nodes = (double *)up(nullptr, S * N * nj * 8); o.total_dis = (double *)up(nullptr, S * N * 8);
    o.route = (double *)up(nullptr, S * N * nj * 8);
    if (out->all_ee) o.all_ee = (double *)up(nullptr, (size_t)S * d->max_iter * 3 * 8);
    if (out->draws_used) o.draws_used = (long long *)up(nullptr, (size_t)S * 8);
    if (out->proposals) o.proposals = (long long *)up(nullptr, (size_t)S * 8);
    if (err == hipSuccess) err = hipMemset(o.parent, 0, S * N * 4);
    if (err == hipSuccess) err = hipMemset(o.nodes, 0, S * N * nj * 8);
    if (err == hipSuccess) err = hipMemset(o.total_dis, 0, S * N * 8);
    if (err == hipSuccess) err = hipMemset(o.route, 0, S * N * nj * 8);
    if (err == hipSuccess && o.all_ee) err = hipMemset(o.all_ee, 0, (size_t)S * d->max_iter * 3 * 8);
    if (err == hipSuccess) {
        rc = cfs_rrt_grow_device(&dd, S, &o, nullptr);
        if (rc == CFS_SUCCESS) err = hipStreamSynchronize(nullptr);
    }
    auto down = [&](void *h, const void *dv, size_t bytes) { if (err == hipSuccess && h) err = hipMemcpy(h, dv, bytes, hipMemcpyDeviceToHost); };
    if (rc == CFS_SUCCESS) {
        down(out->node_num, o.node_num, (size_t)S * 4); down(out->fail, o.fail, (size_t)S * 4); down(out->route_len, o.route_len, (size_t)S * 4);
        down(out->parent, o.parent, S * N * 4); down(out->nodes, o.nodes, S * N * nj * 8); down(out->total_dis, o.total_dis, S * N * 8);
        down(out->route, o.route, S * N * nj * 8);
        if (out->all_ee) down(out->all_ee, o.all_ee, (size_t)S * d->max_iter * 3 * 8);
        if (out->draws_used) down(out->draws_used, o.draws_used, (size_t)S * 8);
        if (out->proposals) down(out->proposals, o.proposals, (size_t)S * 8);
    }
    for (void *p : bufs) (void)hipFree(p);
    if (rc) return rc;
    if (err != hipSuccess) return cfs_fail(CFS_ERR_HIP, "RRT staging failed: %s", hipGetErrorString(err));
    return CFS_SUCCESS;
}
