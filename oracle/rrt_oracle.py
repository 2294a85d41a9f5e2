"""CPU restatement of Lib/RRT_FANUC.m (TEST INFRASTRUCTURE ONLY; see oracle/cfs_oracle.c for the rules).

Sequential, one seed, literal: getRandNode :106-131, feasible :146-181, addNode :184-190, arrangeNode
:134-142, goal_reached :193-207, route back-tracking :86-90.  Distances come from the C oracle.
"parity unpinned": the reference has no RRT tests and MATLAB's rand stream cannot be reproduced; the
random numbers are drawn from an explicit source with the reference's consumption pattern (one uniform per
proposal, nstate more when the sample is random): a numpy Generator, a pre-drawn array (`ArrayRng`), or the
counter-based generator the library documents in include/cfs_hip.h (`splitmix_uniforms`).

Every norm is a plain left-to-right sum of squares in IEEE double (Python floats: no BLAS, no FMA), so that the tree
arithmetic is well defined to the last bit; csrc/cfs_rrt.hip is compiled without FMA contraction and must reproduce parents,
nodes, costs and routes exactly.
"""
import math

import numpy as np

from . import oracle as O

MASK = (1 << 64) - 1


def splitmix_uniforms(seed, tree, n, start=0):
    """u[k] of the library's generator for k = start .. start+n-1 (include/cfs_hip.h, "RRT / RRT*")"""
    out = np.empty(n)
    for i in range(n):
        z = (seed + tree * 0x9E3779B97F4A7C15 + (start + i + 1) * 0xBF58476D1CE4E5B9) & MASK
        z ^= z >> 30
        z = (z * 0xBF58476D1CE4E5B9) & MASK
        z ^= z >> 27
        z = (z * 0x94D049BB133111EB) & MASK
        z ^= z >> 31
        out[i] = (z >> 11) * 2.0 ** -53
    return out


class ArrayRng:
    """a finite stream of uniforms consumed like numpy's Generator.random"""

    def __init__(self, u):
        self.u, self.k = np.asarray(u, float), 0

    def room(self, n):
        return self.k + n <= self.u.size

    def random(self, n=None):
        if n is None:
            self.k += 1
            return float(self.u[self.k - 1])
        self.k += n
        return self.u[self.k - n:self.k].copy()


def _norm(v):
    s = 0.0
    for x in v:
        s += float(x) * float(x)
    return math.sqrt(s)


def find_route(robot, obs, x0, goal, goal_th, region_g, region_s, sample_off, ratial, rng, solver="RRT*", max_iter=400, bi=0.5):
    nstate = len(x0)
    newNode = np.asarray(x0, float).copy()
    all_nodes = [np.concatenate([[-1.0], newNode])]
    total_dis, all_ee = [0.0], []
    node_num, parent, fail = 1, 1, 0
    toNode_dis = np.zeros(0)
    proposals = 0

    def reached(nn):
        return bool(np.all((goal - region_g) < nn) and np.all(nn < (goal + region_g)))

    done = reached(newNode)
    if node_num > max_iter:
        fail, done = 1, True
    while not done:
        while True:                                              # getNode
            if hasattr(rng, "room") and not rng.room(1 + nstate):
                fail = 2                                          # the finite stream is exhausted (the device reports the same)
                break
            proposals += 1
            pp = rng.random()
            sample = (rng.random(nstate) - 0.5) * region_s * 2 + sample_off if pp < bi else np.asarray(goal_th, float)
            nodes = np.array([n[1:] for n in all_nodes])
            toNode_dis = np.array([_norm((n - sample) * ratial) for n in nodes])
            parent, dis = 1, toNode_dis[0]
            for i in range(1, node_num):
                if toNode_dis[i] < dis:
                    dis, parent = toNode_dis[i], i + 1
            near = nodes[parent - 1]
            with np.errstate(divide="ignore", invalid="ignore"):
                newNode = near + (sample - near) * 0.1 / _norm(near - sample)
            feasible = True
            for o in obs:                                        # feasible()
                pos = O.arm_pos(robot, newNode)
                for i in range(nstate):
                    d, pts = O.dist_lin_seg(pos[i, 0], pos[i, 1], o["l"][:, 0], o["l"][:, 1])
                    if abs(d) < 0.0001:
                        d = -np.linalg.norm(pts[:3] - pos[i, 1])
                    if d < o["D"]:
                        feasible = False
                        break
            if feasible:
                break
        if fail:
            break
        all_nodes.append(np.concatenate([[float(parent)], newNode]))
        all_ee.append(O.arm_pos(robot, newNode)[nstate - 1, 0])
        total_dis.append(total_dis[parent - 1] + toNode_dis[parent - 1])
        node_num += 1
        if solver == "RRT*":
            for i in np.nonzero(toNode_dis < 0.2)[0]:
                if total_dis[i] > total_dis[-1] + toNode_dis[i]:
                    all_nodes[i][0] = float(node_num)
                    total_dis[i] = total_dis[-1] + toNode_dis[i]
        done = reached(newNode)
        if node_num > max_iter:
            fail, done = 1, True
    nodes = np.array(all_nodes)
    route = [newNode]
    p = parent if node_num > 1 else -1
    steps = 0
    while p != -1 and steps <= node_num:
        route.insert(0, nodes[p - 1, 1:])
        p = int(nodes[p - 1, 0])
        steps += 1
    if p != -1:                                                   # RRT* re-parenting closed a cycle: the reference would never return
        fail, route = fail or 3, [newNode]
    return dict(route=np.array(route).T, all_nodes=nodes.T, total_dis=np.array(total_dis), node_num=node_num, fail=bool(fail), fail_code=fail,
                all_ee=np.array(all_ee).T if all_ee else np.zeros((3, 0)), proposals=proposals)
