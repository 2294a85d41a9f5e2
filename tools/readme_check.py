"""Runs the usage snippet of README.md on a GPU (with a generated STL in place of the reference map)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, motionplanning_5d_m_amd as mp
from motionplanning_5d_m_amd import mesh as M
ROBOT, sys_info, obs = mp.main_FANUC_problem()
cfs = mp.CFS_FANUC(obs, sys_info, ROBOT).optimizer()
psg = mp.PSGCFS_FANUC(obs, sys_info, ROBOT).optimizer(noise=0.1 * np.random.default_rng(0).standard_normal((20, 150)))
M.write_stl_binary('/tmp/table.STL', M.box_mesh([3700, 8000, 0], [4200, 9000, 400], n=6))
mesh = mp.Mesh.from_stl('/tmp/table.STL', scale=1e-3, map_from_stl=False)
d, linkid, pts = mp.dist_arm_surf(sys_info.robot, cfs.x_.reshape(30, 10)[:, :5], mesh)
both = mp.CFS_FANUC(obs + [dict(mesh=mesh, D=0.2, epsilon=0.25)], sys_info, ROBOT).optimizer()
sys_info.MAX_O_ITER = 4
chomp = mp.CHOMP_FANUC([dict(num_obs=1)] + obs, sys_info, np.zeros(150), ROBOT).optimizer()
batch = mp.CFSBatch(sys_info, nobs=1, margin=[0.25], mode="CFS", max_batch=1024)
print(cfs.iter_O, psg.iter_O, d.min(), both.iter_O, both.status, chomp.iter_O, 'ok')
