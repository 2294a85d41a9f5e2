"""Developer probe: run each stage of the HIP path against the oracle and print the errors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import motionplanning_5d_m_amd as pkg
from oracle import oracle as O

def compare(name, ROBOT, s, obs, P, mode="CFS", noise=None):
    cls = pkg.CFS_FANUC if mode == "CFS" else pkg.PSGCFS_FANUC
    slv = cls(obs, s, ROBOT)
    # stage 1: linearisation at the initial trajectory
    ob = pkg.obs_to_array(obs)[None]
    dist, lid, grad = slv._batch.linearize(np.asarray(s.x_)[None], ob)
    A, b, od, ol, og = O.get_con(P.ROBOT, P.sys_info, P.obs, P.sys_info.x_, np.zeros(s.H * s.nu), mode)
    print(f"[{name}] K1 dist err {np.abs(dist[0]-od).max():.3e} grad err {np.abs(grad[0]-og).max():.3e} linkid mismatches {(lid[0]!=ol).sum()}")
    slv.get_con()
    print(f"[{name}] dense Ainq err {np.abs(slv.Ainq-A).max():.3e} binq err {np.abs(slv.binq-b).max():.3e}")
    t = time.time()
    got = slv.optimizer(noise=noise) if mode != "CFS" else slv.optimizer()
    dt = time.time() - t
    want = O.optimizer(P.ROBOT, P.sys_info, P.obs, mode, noise=noise)
    n = min(len(got.eval.cost_all), len(want.cost_all))
    print(f"[{name}] status gpu={pkg.STATUS[got.status]} oracle={O.STATUS[want.status]} iter_O {got.iter_O}/{want.iter_O} total_iter {got.total_iter}/{want.total_iter} time {dt*1e3:.1f} ms")
    print(f"[{name}] linf x_ {np.abs(got.x_-want.x_).max():.3e}  linf u {np.abs(got.u-want.u).max():.3e}  cost rel {np.abs(got.eval.cost_all[:n]-want.cost_all[:n]).max()/max(1,np.abs(want.cost_all[:n]).max()) if n else 0:.3e} e_u {np.abs(got.eval.e_u_all[:n]-want.e_u_all[:n]).max() if n else 0:.3e}")

print("devices", pkg.device_count())
R, s, obs = pkg.main_FANUC_problem(); compare("FANUC-CFS", R, s, obs, O.problem_main_FANUC())
rng = np.random.default_rng(1); nz = 0.1 * rng.standard_normal((20, 150))
compare("FANUC-PSG", R, s, obs, O.problem_main_FANUC(), "PSGCFS", nz)
R, s, obs = pkg.main_2L_problem(); compare("2L", R, s, obs, O.problem_main_2L())
R, s, obs = pkg.main_2L_problem(lim=(1, 1)); compare("2L-lim1", R, s, obs, O.problem_main_2L(lim=(1, 1)))
rw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "route_wp_200i_xori.npy"))
R, s, obs = pkg.RRTstar_CFS_problem(rw); compare("RRT", R, s, obs, O.problem_RRTstar_CFS(rw))
