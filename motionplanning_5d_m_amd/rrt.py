"""Row f1 of the scope table: RRT / RRT* tree growth in joint space with capsule feasibility checks.

Host-side mirror of ``RRT_FANUC(obs, sys_info, goal, region_g, region_s, sample_off, ROBOT, SOLVER)
.find_route()`` (Lib/RRT_FANUC.m:48-207) and of the seed-parallel wrapper ``s_Parallel_rrt``
(Lib/functions/s_Parallel_rrt.m:9-28).  The tree bookkeeping is sequential per seed and stays on the
host, exactly as in the reference; the feasibility test -- forward kinematics + segment distances of
the candidate against every obstacle (RRT_FANUC.m:146-181) -- is the same geometry kernel as the CFS
path and runs on the GPU (``cfs_dist_arm``), batched over all seeds that are growing in lock-step
(the reference grows 6 seeds under ``parfor``; here any number).

MATLAB's ``rand`` stream cannot be reproduced, so every seed draws from its own
``numpy.random.Generator``: one uniform for the goal bias (RRT_FANUC.m:108), and ``nstate`` more when
the sample is random (:111) -- the same consumption pattern as the reference.

Quirks kept: the edge cost added is the parent->SAMPLE weighted distance (:187);
RRT* re-parents nodes within 0.2 of the SAMPLE without propagating cost changes (:134-142); failure
when node_num > MAX_ITER = 400 (:201-205).  The near-zero branch of the reference's feasibility test is
ill-formed (6x1 minus 3x1, :170); it is defined as in dist_arm_3D_200i_2.m:23.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from .solvers import dist_arm, obs_to_array


class _Tree:
    def __init__(self, x0, rng):
        self.rng = rng
        self.newNode = np.asarray(x0, float).copy()
        self.all_nodes = [np.concatenate([[-1.0], self.newNode])]     # rows of [parent; node] (RRT_FANUC.m:66)
        self.total_dis = [0.0]
        self.all_ee = []
        self.node_num = 1
        self.parent = 1
        self.toNode_dis = np.zeros(0)
        self.done = False
        self.fail = False
        self.pos_last = None


class RRT_FANUC:
    MAX_ITER = 400      # RRT_FANUC.m:37
    bi = 0.5            # RRT_FANUC.m:38

    def __init__(self, obs, sys_info, goal, region_g, region_s, sample_off, ROBOT="M16iB", SOLVER="RRT*"):
        self.obs, self.sys_info = obs, sys_info
        self.goal, self.region_g = np.asarray(goal, float).reshape(-1), np.asarray(region_g, float).reshape(-1)
        self.region_s, self.sample_off = np.asarray(region_s, float).reshape(-1), np.asarray(sample_off, float).reshape(-1)
        self.ROBOT, self.SOLVER = ROBOT, SOLVER
        self._obs_arr = obs_to_array(obs)
        self._D = np.array([o["D"] for o in obs], float)

    # ---- one lock-step growth of S trees --------------------------------------------------------------
    def _propose(self, t: _Tree):
        """getRandNode (RRT_FANUC.m:106-131)."""
        s = self.sys_info
        pp = t.rng.random()
        if pp < self.bi:
            sample = (t.rng.random(s.nstate) - 0.5) * self.region_s * 2 + self.sample_off
        else:
            sample = np.asarray(s.goal_th, float).reshape(-1)
        nodes = np.array([n[1:] for n in t.all_nodes])
        dis = np.linalg.norm((nodes - sample) * np.asarray(s.ratial, float).reshape(-1), axis=1)
        t.toNode_dis = dis
        t.parent = int(np.argmin(dis)) + 1                      # first minimum wins (strict <, :124)
        near = nodes[t.parent - 1]
        t.newNode = near + (sample - near) * 0.1 / np.linalg.norm(near - sample)

    def _goal_reached(self, t: _Tree):
        """goal_reached (RRT_FANUC.m:193-207)."""
        reached = bool(np.all((self.goal - self.region_g) < t.newNode) and np.all(t.newNode < (self.goal + self.region_g)))
        if t.node_num > self.MAX_ITER:
            t.fail = True
            reached = True
        t.done = reached

    def _add(self, t: _Tree):
        """addNode (+ arrangeNode for RRT*) (RRT_FANUC.m:134-142,184-190)."""
        t.all_nodes.append(np.concatenate([[float(t.parent)], t.newNode]))
        t.all_ee.append(t.pos_last)
        t.total_dis.append(t.total_dis[t.parent - 1] + t.toNode_dis[t.parent - 1])
        t.node_num += 1
        if self.SOLVER == "RRT*":
            for i in np.nonzero(t.toNode_dis < 0.2)[0]:
                if t.total_dis[i] > t.total_dis[-1] + t.toNode_dis[i]:
                    t.all_nodes[i][0] = float(t.node_num)
                    t.total_dis[i] = t.total_dis[-1] + t.toNode_dis[i]

    def grow(self, rngs):
        """find_route for len(rngs) independent seeds in lock-step; returns one result per seed."""
        s = self.sys_info
        nj = s.nstate
        trees = [_Tree(s.x0, r) for r in rngs]
        for t in trees:
            self._goal_reached(t)
        need = [t for t in trees if not t.done]                 # trees that must produce a feasible node
        while need:
            for t in need:
                self._propose(t)
            th = np.stack([t.newNode for t in need])
            d, _, pos = dist_arm(s.robot, th, self._obs_arr, want_pos=True)      # GPU: (S, nobs), (S, nj, 2, 3)
            ok = (d >= self._D[None, :]).all(axis=1)           # feasible(): no link closer than obs{j}.D (:146-181)
            nxt = []
            for k, t in enumerate(need):
                if not ok[k]:
                    nxt.append(t)                               # getNode keeps sampling (:95-103)
                    continue
                t.pos_last = pos[k, nj - 1, 0].copy()           # pos{nstate}.p(:,1) (:186)
                self._add(t)
                self._goal_reached(t)
                if not t.done:
                    nxt.append(t)
            need = nxt
        out = []
        for t in trees:
            nodes = np.array(t.all_nodes)
            route = [t.newNode]
            parent = t.parent if t.node_num > 1 else -1
            while parent != -1:                                 # :86-90
                route.insert(0, nodes[parent - 1, 1:])
                parent = int(nodes[parent - 1, 0])
            out.append(SimpleNamespace(route=np.array(route).T, all_nodes=nodes.T, total_dis=np.array(t.total_dis),
                                       all_ee=np.array(t.all_ee).T if t.all_ee else np.zeros((3, 0)),
                                       node_num=t.node_num, fail=t.fail))
        return out

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
