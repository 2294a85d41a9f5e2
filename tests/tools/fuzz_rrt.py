"""Developer fuzz of row f1: random planning problems (robot, joint count, 0-4 line obstacles, start, goal, regions, weights, solver),
trees grown on the device by cfs_rrt_grow against oracle/rrt_oracle.py fed the same uniforms: parents, nodes, costs and routes
must be the same BITS, failures included.  usage: python tests/tools/fuzz_rrt.py [seed] [problems] [trees per problem]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


def problem(rng):
    rid = str(rng.choice(["M200i", "M16iB", "2L"], p=[0.5, 0.35, 0.15]))
    nj = 2 if rid == "2L" else int(rng.integers(3, 7))
    solver = str(rng.choice(["RRT", "RRT*"]))
    nobs = int(rng.integers(0, 5))
    if rid == "2L":
        x0, goal = np.zeros(2), np.array([rng.uniform(0.6, 1.6), rng.uniform(-0.4, 0.4)])
        obs = [np.array([c[0], c[1], 0.0, c[0], c[1], 0.0]) for c in rng.uniform(0.15, 0.5, (nobs, 2))]
        D = rng.uniform(0.02, 0.06, nobs)
    else:
        base6 = np.array([0.421, 0, -0.0092, -0.0010, -1.5786, 0.2]) if rid == "M200i" else np.array([0.5, 1.2, 0.1, 0.0, -1.2, 0.2])
        g6 = np.array([-1.4090, 0.8873, 0.4008, 0.0, 0.4430, -0.3]) if rid == "M200i" else np.array([-0.6, 1.0, 0.3, 0.1, -1.0, 0.4])
        x0, goal = (base6 + rng.uniform(-0.1, 0.1, 6))[:nj], (g6 + rng.uniform(-0.15, 0.15, 6))[:nj]
        c0 = np.array([3.150, 8.500]) if rid == "M200i" else np.array([3.250, 8.500])
        obs = []
        for _ in range(nobs):
            ang, rad = rng.uniform(0, 2 * np.pi), rng.uniform(0.45, 1.2)
            x, y = c0 + rad * np.array([np.cos(ang), np.sin(ang)])
            z1 = rng.uniform(0.0, 0.8)
            obs.append(np.array([x, y, z1, x + rng.uniform(-0.2, 0.2), y + rng.uniform(-0.2, 0.2), z1 + rng.uniform(0.2, 0.9)]))
        D = rng.uniform(0.1, 0.25, nobs)
    region_g = np.array([np.pi / 20, np.pi / 20, np.pi / 10, np.pi / 2, np.pi / 2, np.pi / 2])[:nj] * rng.uniform(0.7, 1.5)
    region_s = np.array([np.pi / 2, np.pi / 2, np.pi / 2, np.pi / 1.5, np.pi / 1.5, np.pi / 1.5])[:nj] * rng.uniform(0.8, 1.2)
    off = rng.uniform(-0.2, 0.2, nj) * (rng.random() < 0.5)
    ratial = np.array([1, 1, 0.5, 0.1, 0.1, 0.1])[:nj] * rng.uniform(0.5, 1.5, nj)
    return dict(rid=rid, nj=nj, solver=solver, x0=x0, goal=goal, obs=obs, D=D, region_g=region_g, region_s=region_s, off=off, ratial=ratial)


def oracle_tree(args):
    root, p, u = args
    sys.path.insert(0, root)
    from oracle import oracle as O2, rrt_oracle as R
    robot = O2.robotproperty2(p["rid"])
    obs = [dict(l=np.stack([o[:3], o[3:]], axis=1), D=float(d)) for o, d in zip(p["obs"], p["D"])]
    return R.find_route(robot, obs, p["x0"], p["goal"], p["goal"], p["region_g"], p["region_s"], p["off"], p["ratial"], R.ArrayRng(u), p["solver"])


if __name__ == "__main__":
    import concurrent.futures as cf
    import multiprocessing as mp
    from types import SimpleNamespace
    import motionplanning_5d_m_amd as pkg
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    nprob = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    rng = np.random.default_rng(seed)
    probs = [problem(rng) for _ in range(nprob)]
    U = [np.stack([np.random.default_rng(1000 * seed + 10 * k + t).random((1 + p["nj"]) * 8 * 401) for t in range(T)]) for k, p in enumerate(probs)]
    got = []
    for p, u in zip(probs, U):
        robot = pkg.robotproperty2(p["rid"])
        s = SimpleNamespace(robot=robot, DH=robot.DH, nstate=p["nj"], base=robot.base, x0=p["x0"], ratial=p["ratial"], goal_th=p["goal"])
        obs = [dict(l=np.stack([o[:3], o[3:]], axis=1), D=float(d), epsilon=float(d)) for o, d in zip(p["obs"], p["D"])]
        got.append(pkg.RRT_FANUC(obs, s, p["goal"], p["region_g"], p["region_s"], p["off"], p["rid"], p["solver"]).grow(uniforms=u))
    jobs = [(ROOT, p, U[k][t]) for k, p in enumerate(probs) for t in range(T)]
    with cf.ProcessPoolExecutor(min(16, os.cpu_count() or 1), mp_context=mp.get_context("spawn")) as ex:
        want = list(ex.map(oracle_tree, jobs, chunksize=2))
    bad = 0
    for k, p in enumerate(probs):
        line = []
        for t in range(T):
            r, w = got[k][t], want[k * T + t]
            same = (r.node_num == w["node_num"] and r.fail_code == w["fail_code"] and np.array_equal(r.all_nodes, w["all_nodes"])
                    and np.array_equal(r.total_dis, w["total_dis"]) and np.array_equal(r.route, w["route"]) and r.proposals == w["proposals"]
                    and (r.all_ee.size == 0 or np.abs(r.all_ee - w["all_ee"]).max() < 1e-13))
            bad += not same
            line.append(f"{r.node_num}/{w['node_num']}{'' if same else '!'}f{r.fail_code}")
        print(f"problem {k:2d} {p['rid']:6s} nj={p['nj']} nobs={len(p['obs'])} {p['solver']:4s} nodes dev/oracle: {' '.join(line)}", flush=True)
    print(f"flagged {bad} of {nprob * T}")
