"""Row f1 of the scope table: RRT / RRT* tree growth in joint space with capsule feasibility checks.

Host-side mirror of ``RRT_FANUC(obs, sys_info, goal, region_g, region_s, sample_off, ROBOT, SOLVER).find_route()``
(Lib/RRT_FANUC.m:48-207) and of the seed-parallel wrapper ``s_Parallel_rrt`` (Lib/functions/s_Parallel_rrt.m:9-28) over
``cfs_rrt_grow``: whole trees grow on the GPU, one wavefront per tree, any number of trees per launch (the reference grows
6 seeds under ``parfor``).  This module only packs arguments and unpacks results into the reference's field names
(``route``, ``all_nodes``, ``total_dis``, ``all_ee``, ``node_num``, ``fail``); there is no host-side tree code.

MATLAB's ``rand`` stream cannot be reproduced, so the random numbers are explicit: either numpy Generators (one per tree;
``ndraw`` uniforms are drawn from each up front and consumed on the device exactly as the reference consumes ``rand`` -- one
per proposal for the goal bias (:108), ``nstate`` more when the sample is random (:111)), or a seed for the library's
counter-based generator (``seed=``; nothing but the seed crosses the boundary).

Quirks kept (csrc/cfs_rrt.hip): the edge cost added is the parent->SAMPLE weighted distance (:187); RRT*
re-parents nodes within 0.2 of the SAMPLE without propagating cost changes (:134-142); failure when node_num > MAX_ITER = 400
(:201-205).  The near-zero branch of the reference's feasibility test is ill-formed (6x1 minus 3x1, :170); it is defined as in
dist_arm_3D_200i_2.m:23.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

from . import _lib
from .robotproperty2 import to_c_robot
from .solvers import _f64, _ptr, obs_to_array

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

FAIL = {0: "route found", 1: "node_num > MAX_ITER", 2: "uniforms exhausted", 3: "re-parenting cycle"}


class RRT_FANUC:
    MAX_ITER = 400      # RRT_FANUC.m:37
    bi = 0.5            # RRT_FANUC.m:38
    REWIRE = 0.2        # RRT_FANUC.m:135

    def __init__(self, obs, sys_info, goal, region_g, region_s, sample_off, ROBOT="M16iB", SOLVER="RRT*"):
        self.obs, self.sys_info = obs, sys_info
        self.goal, self.region_g = _f64(np.asarray(goal, float).reshape(-1)), _f64(np.asarray(region_g, float).reshape(-1))
        self.region_s, self.sample_off = _f64(np.asarray(region_s, float).reshape(-1)), _f64(np.asarray(sample_off, float).reshape(-1))
        self.ROBOT, self.SOLVER = ROBOT, SOLVER
        if SOLVER not in ("RRT", "RRT*"):
            raise ValueError("SOLVER must be 'RRT' or 'RRT*'")
        self._obs_arr = obs_to_array(obs) if len(obs) else np.zeros((0, 6))
        self._D = _f64([o["D"] for o in obs])

    # ---- argument packing ----------------------------------------------------------------------------------------
    def _desc(self, conv, x0=None, goal=None, goal_th=None, per_tree=False):
        """(descriptor, arrays kept alive); conv turns an array-like into what the entry point reads (host numpy | CUDA tensor)"""
        s = self.sys_info
        d = _lib.cfs_rrt_desc()
        d.robot = to_c_robot(s.robot)
        d.nstate, d.solver, d.max_iter = int(s.nstate), (1 if self.SOLVER == "RRT*" else 0), int(self.MAX_ITER)
        d.bi, d.rewire, d.per_tree = float(self.bi), float(self.REWIRE), 1 if per_tree else 0
        arrs = dict(x0=s.x0 if x0 is None else x0, goal=self.goal if goal is None else goal,
                    goal_th=s.goal_th if goal_th is None else goal_th, region_g=self.region_g, region_s=self.region_s,
                    sample_off=self.sample_off, ratial=s.ratial, obs=self._obs_arr, D=self._D)
        keep = {k: conv(v) for k, v in arrs.items()}
        for k, v in keep.items():
            setattr(d, k, _ptr(v))
        d.nobs = int(self._obs_arr.shape[0])
        return d, keep

    # ---- S trees, host arrays in and out (cfs_rrt_grow) -----------------------------------------------------------------
    def grow(self, rngs=None, *, uniforms=None, seed=None, S=None, ndraw=None, max_draws=None, x0=None, goal=None, goal_th=None):
        """find_route for S independent trees; returns one result per tree with the reference's field names.
        rngs: numpy Generators (ndraw uniforms are drawn from each); or uniforms: (S, ndraw) array; or seed + S: the library's
        counter-based generator.  x0 / goal / goal_th: optional (S, nstate) per-tree start and goal (default: shared)."""
        s = self.sys_info
        nj, N = int(s.nstate), self.MAX_ITER + 1
        if ndraw is None:
            ndraw = (1 + nj) * 8 * N
        if rngs is not None:
            uniforms = np.stack([r.random(ndraw) for r in rngs])
        if uniforms is not None:
            uniforms = _f64(uniforms)
            S = uniforms.shape[0]
        elif seed is None or S is None:
            raise ValueError("pass rngs, uniforms, or seed and S")
        per_tree = x0 is not None
        if per_tree:
            x0, goal = _f64(x0), _f64(goal)
            goal_th = goal if goal_th is None else _f64(goal_th)
            assert x0.shape == (S, nj) and goal.shape == (S, nj) and goal_th.shape == (S, nj)
        d, keep = self._desc(lambda v: _f64(np.asarray(v, float)), x0, goal, goal_th, per_tree)
        if uniforms is not None:
            d.uniforms, d.ndraw = _ptr(uniforms), int(uniforms.shape[1])
        else:
            d.seed, d.max_draws = int(seed), int(max_draws if max_draws is not None else (1 + nj) * 8 * N)
        r = SimpleNamespace(node_num=np.zeros(S, np.int32), fail=np.zeros(S, np.int32), parent=np.zeros((S, N), np.int32),
                            nodes=np.zeros((S, N, nj)), total_dis=np.zeros((S, N)), all_ee=np.zeros((S, self.MAX_ITER, 3)),
                            route_len=np.zeros(S, np.int32), route=np.zeros((S, N, nj)), draws_used=np.zeros(S, np.int64),
                            proposals=np.zeros(S, np.int64))
        o = _lib.cfs_rrt_out()
        for k in ("node_num", "fail", "parent", "nodes", "total_dis", "all_ee", "route_len", "route", "draws_used", "proposals"):
            setattr(o, k, _ptr(getattr(r, k)))
        _lib.check(_lib.lib().cfs_rrt_grow(C.byref(d), S, C.byref(o)))
        out = []
        for t in range(S):
            n, L = int(r.node_num[t]), int(r.route_len[t])
            all_nodes = np.concatenate([r.parent[t, :n, None].astype(float), r.nodes[t, :n]], axis=1).T       # rows [parent; node] (RRT_FANUC.m:66)
            out.append(SimpleNamespace(route=r.route[t, :L].T.copy(), all_nodes=all_nodes, total_dis=r.total_dis[t, :n].copy(),
                                       all_ee=r.all_ee[t, :n - 1].T.copy(), node_num=n, fail=bool(r.fail[t] != 0), fail_code=int(r.fail[t]),
                                       draws_used=int(r.draws_used[t]), proposals=int(r.proposals[t])))
        return out

    # ---- S trees, device-resident results (cfs_rrt_grow_device) ----------------------------------------------------------
    def grow_device(self, S, seed, device, max_draws=None, x0=None, goal=None, goal_th=None, stream=None, want_tree=False):
        """S trees from the library's generator with everything left on the GPU: returns a namespace of CUDA tensors
        (route (S, MAX_ITER+1, nstate), route_len, fail, node_num, proposals [, parent, nodes, total_dis]) -- routes go straight
        into CFSBatch.build_terms_from_ragged_routes_device.  x0 / goal [/ goal_th]: optional (S, nstate) CUDA tensors."""
        s = self.sys_info
        nj, N = int(s.nstate), self.MAX_ITER + 1
        conv = lambda a: (a if isinstance(a, torch.Tensor) else torch.tensor(np.asarray(a, float), dtype=torch.float64, device=device)).contiguous()  # noqa: E731
        per_tree = x0 is not None
        if per_tree:
            goal_th = goal if goal_th is None else goal_th
            for v in (x0, goal, goal_th):
                assert v.is_cuda and v.dtype == torch.float64 and tuple(v.shape) == (S, nj)
        d, keep = self._desc(conv, x0, goal, goal_th, per_tree)
        d.seed, d.max_draws = int(seed), int(max_draws if max_draws is not None else (1 + nj) * 8 * N)
        z = lambda *sh, dt=torch.float64: torch.zeros(*sh, dtype=dt, device=device)  # noqa: E731
        r = SimpleNamespace(node_num=z(S, dt=torch.int32), fail=z(S, dt=torch.int32), route_len=z(S, dt=torch.int32), parent=z(S, N, dt=torch.int32),
                            nodes=z(S, N, nj), total_dis=z(S, N), route=z(S, N, nj), proposals=z(S, dt=torch.int64), _keep=keep)
        o = _lib.cfs_rrt_out()
        for k in ("node_num", "fail", "parent", "nodes", "total_dis", "route_len", "route", "proposals"):
            setattr(o, k, _ptr(getattr(r, k)))
        if stream is None:
            stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(_lib.lib().cfs_rrt_grow_device(C.byref(d), S, C.byref(o), C.c_void_p(stream)))
        if not want_tree:
            r.parent = r.nodes = r.total_dis = None
        return r

    def find_route(self, rng=None):
        r = self.grow([rng if rng is not None else np.random.default_rng()])[0]
        self.route, self.all_nodes, self.total_dis, self.all_ee = r.route, r.all_nodes, r.total_dis, r.all_ee
        self.node_num, self.fail = r.node_num, r.fail
        return self


def s_Parallel_rrt(obs, sys_info, goal, region_g, region_s, sample_off, ROBOT="M200i", num_seed=6, seed=0, max_rounds=50):
    """Lib/functions/s_Parallel_rrt.m:9-28: rounds of `num_seed` RRT seeds (solver 'RRT', :17) until at least one
    succeeds; the route with the fewest nodes wins.  Returns (best result, iter_rrt, all results of the last round)."""
    planner = RRT_FANUC(obs, sys_info, goal, region_g, region_s, sample_off, ROBOT, "RRT")
    ss = np.random.SeedSequence(seed)
    for it in range(1, max_rounds + 1):
        rngs = [np.random.default_rng(c) for c in ss.spawn(num_seed)]
        res = planner.grow(rngs)
        routeL = np.array([1000 if r.fail else r.route.shape[1] for r in res])
        if not all(r.fail for r in res):
            return res[int(np.argmin(routeL))], it, res
    raise RuntimeError("no RRT seed found a route")


def RRTstar_problem():
    """The RRT stage of RRTstar_CFS.m:16-64 as checked in: start, goal, obstacles, sampling regions."""
    from .robotproperty2 import robotproperty2
    from .sysinfo import cylinder
    robot = robotproperty2("M200i")
    s = SimpleNamespace(robot=robot, DH=robot.DH, nstate=5, base=robot.base,
                        x0=np.array([0.421, 0, -0.0092, -0.0010, -1.5786]), ratial=np.array([1, 1, 0.5, 0.1, 0.1]),
                        goal_th=np.array([-1.4090, 0.8873, 0.4008, 0.0, 0.4430]))
    obs = [cylinder((3606, 8413, 1), (3606, 8413, 1038), 0.2, 0.2), cylinder((3406, 7813, 800), (3406, 7813, 1538), 0.2, 0.2)]
    region_g = np.array([np.pi / 20, np.pi / 20, np.pi / 10, np.pi / 2, np.pi / 2])
    region_s = np.array([np.pi / 2, np.pi / 2, np.pi / 2, np.pi / 1.5, np.pi / 1.5])
    return obs, s, s.goal_th.copy(), region_g, region_s, np.zeros(5)
