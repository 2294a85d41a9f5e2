import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import motionplanning_5d_m_amd as gpu
pobs, s, g, region_g, region_s, off = gpu.RRTstar_problem()
for seed in range(1, 9):
    best, iter_rrt, res = gpu.s_Parallel_rrt(pobs, s, g, region_g, region_s, off, "M200i", num_seed=6, seed=seed)
    R, sys_info, obs = gpu.RRTstar_CFS_problem(best.route)
    out = gpu.CFS_FANUC(obs, sys_info, R).optimizer()
    print(seed, "route wp", best.route.shape[1], "status", out.status, "iter_O-1", out.iter_O - 1, "cost_new %.4e" % out.eval.cost_new, "caug %.4e" % sys_info.caug)
