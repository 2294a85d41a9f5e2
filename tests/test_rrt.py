"""Row f1: RRT / RRT* (Lib/RRT_FANUC.m, Lib/functions/s_Parallel_rrt.m).  CPU: the oracle restatement is
self-consistent; GPU: the host mirror with batched GPU feasibility grows the same trees node for node."""
import numpy as np
import pytest


def _setup(O):
    robot = O.robotproperty2("M200i")
    obs = [dict(l=np.array([[3606, 8413, 1], [3606, 8413, 1038]], float).T / 1000, D=0.2),
           dict(l=np.array([[3406, 7813, 800], [3406, 7813, 1538]], float).T / 1000, D=0.2)]
    x0 = np.array([0.421, 0, -0.0092, -0.0010, -1.5786])
    goal = np.array([-1.4090, 0.8873, 0.4008, 0.0, 0.4430])
    rg = np.array([np.pi / 20, np.pi / 20, np.pi / 10, np.pi / 2, np.pi / 2])
    rs = np.array([np.pi / 2, np.pi / 2, np.pi / 2, np.pi / 1.5, np.pi / 1.5])
    return robot, obs, x0, goal, rg, rs, np.array([1, 1, 0.5, 0.1, 0.1])


def test_oracle_rrt_routes_are_valid(O):
    from oracle import rrt_oracle as R
    robot, obs, x0, goal, rg, rs, ratial = _setup(O)
    found = 0
    for sd in (1, 2):
        r = R.find_route(robot, obs, x0, goal, goal, rg, rs, np.zeros(5), ratial, np.random.default_rng(sd), "RRT")
        assert r["node_num"] == r["all_nodes"].shape[1] and r["all_nodes"][0, 0] == -1
        route = r["route"]
        np.testing.assert_array_equal(route[:, 0], x0)
        if not r["fail"]:
            found += 1
            assert np.all(np.abs(route[:, -1] - goal) < rg)                           # goal_reached (RRT_FANUC.m:195-199)
            np.testing.assert_allclose(np.linalg.norm(np.diff(route, axis=1), axis=0), 0.1, atol=1e-12)   # fixed 0.1 rad step (:129)
        for k in range(1, r["node_num"], 7):                                           # every node passed feasible()
            for o in obs:
                assert O.dist_arm(robot, r["all_nodes"][1:, k], o["l"])[0] >= o["D"]
        par = r["all_nodes"][0, 1:].astype(int)
        assert np.all(par >= 1) and np.all(par <= np.arange(1, r["node_num"]))        # parents precede children (plain RRT)
    assert found >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["RRT", "RRT*"])
def test_rrt_trees_match_oracle_node_for_node(gpu, O, solver):
    from oracle import rrt_oracle as R
    robot, obs, x0, goal, rg, rs, ratial = _setup(O)
    pobs, s, g, region_g, region_s, off = gpu.RRTstar_problem()
    planner = gpu.RRT_FANUC(pobs, s, g, region_g, region_s, off, "M200i", solver)
    seeds = [1, 2, 3, 5]
    got = planner.grow([np.random.default_rng(sd) for sd in seeds])                  # 4 trees in lock-step, batched feasibility
    for sd, r in zip(seeds, got):
        w = R.find_route(robot, obs, x0, goal, goal, rg, rs, np.zeros(5), ratial, np.random.default_rng(sd), solver)
        assert r.node_num == w["node_num"] and r.fail == w["fail"]
        np.testing.assert_array_equal(r.all_nodes[0], w["all_nodes"][0])              # parent indices
        np.testing.assert_allclose(r.all_nodes[1:], w["all_nodes"][1:], rtol=0, atol=1e-12)
        np.testing.assert_allclose(r.total_dis, w["total_dis"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(r.route, w["route"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(r.all_ee, w["all_ee"], rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_parallel_rrt_then_cfs_pipeline(gpu):
    """RRTstar_CFS.m end to end with its own start, goal, obstacles and weights: s_Parallel_rrt -> cubic resampling to 41
    points -> CFS_FANUC.optimizer().  The only whole-loop numbers the reference holds are its hand-kept run logs of exactly
    this script (M200i/test.xlsx rows 5-18 and row 19, M200i/test.csv:1; RRT is random, so they are a band, not a value):
    final cost eval.cost_new between 1.5e5 and 4e5 (mean of 14 runs 1.964e5), iter_O - 1 between 2 and 21.  Runs whose
    linearisation becomes infeasible have no counterpart there (the reference ignores quadprog's exitflag and would crash at
    Lib/CFS_FANUC.m:92): they are reported, and must stay a minority."""
    pobs, s, g, region_g, region_s, off = gpu.RRTstar_problem()
    costs, its, infeasible = [], [], 0
    for seed in (1, 3, 6, 9):
        best, iter_rrt, res = gpu.s_Parallel_rrt(pobs, s, g, region_g, region_s, off, "M200i", num_seed=6, seed=seed)
        assert not best.fail and iter_rrt >= 1 and len(res) == 6
        assert best.route.shape[1] == min(r.route.shape[1] for r in res if not r.fail)
        R, sys_info, obs = gpu.RRTstar_CFS_problem(best.route)
        out = gpu.CFS_FANUC(obs, sys_info, R).optimizer()
        assert out.status in (0, 1, 2) and out.iter_O >= 2
        if out.status < 2:
            x = out.x_.reshape(40, 10)
            assert np.abs(x[:, 5:]).max() <= 1 + 1e-6
            costs.append(out.eval.cost_new)
            its.append(out.iter_O - 1)
        else:
            infeasible += 1
    print(f"[RRT*-CFS pipeline] final costs {costs}, outer iterations {its}, infeasible linearisations {infeasible}/4 "
          "(reference logs: 1.5e5-4e5, mean 1.964e5; 2-21 iterations)")
    assert len(costs) >= 2
    assert all(1.5e5 <= c <= 4e5 for c in costs), costs                 # M200i/test.xlsx col C rows 5-18
    assert all(2 <= k <= 21 for k in its), its                           # col E minus iter_rrt = 1
