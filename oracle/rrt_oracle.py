"""CPU restatement of Lib/RRT_FANUC.m (TEST INFRASTRUCTURE ONLY; see oracle/cfs_oracle.c for the rules).

Sequential, one seed, literal: getRandNode :106-131, feasible :146-181, addNode :184-190, arrangeNode
:134-142, goal_reached :193-207, route back-tracking :86-90.  Distances come from the C oracle.
"parity unpinned": the reference has no RRT tests and MATLAB's rand stream cannot be reproduced; the
random numbers are drawn from a numpy Generator with the reference's consumption pattern.
"""
import numpy as np

from . import oracle as O


def find_route(robot, obs, x0, goal, goal_th, region_g, region_s, sample_off, ratial, rng, solver="RRT*", max_iter=400, bi=0.5):
    nstate = len(x0)
    newNode = np.asarray(x0, float).copy()
    all_nodes = [np.concatenate([[-1.0], newNode])]
    total_dis, all_ee = [0.0], []
    node_num, parent, fail = 1, 1, False
    toNode_dis = np.zeros(0)

    def reached(nn):
        return bool(np.all((goal - region_g) < nn) and np.all(nn < (goal + region_g)))

    done = reached(newNode)
    if node_num > max_iter:
        fail, done = True, True
    while not done:
        while True:                                              # getNode
            pp = rng.random()
            sample = (rng.random(nstate) - 0.5) * region_s * 2 + sample_off if pp < bi else np.asarray(goal_th, float)
            nodes = np.array([n[1:] for n in all_nodes])
            toNode_dis = np.array([np.linalg.norm((n - sample) * ratial) for n in nodes])
            parent, dis = 1, toNode_dis[0]
            for i in range(1, node_num):
                if toNode_dis[i] < dis:
                    dis, parent = toNode_dis[i], i + 1
            near = nodes[parent - 1]
            newNode = near + (sample - near) * 0.1 / np.linalg.norm(near - sample)
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
            fail, done = True, True
    nodes = np.array(all_nodes)
    route = [newNode]
    p = parent if node_num > 1 else -1
    while p != -1:
        route.insert(0, nodes[p - 1, 1:])
        p = int(nodes[p - 1, 0])
    return dict(route=np.array(route).T, all_nodes=nodes.T, total_dis=np.array(total_dis), node_num=node_num, fail=fail,
                all_ee=np.array(all_ee).T if all_ee else np.zeros((3, 0)))
