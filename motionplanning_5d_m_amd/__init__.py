"""motionplanning_5d_m_amd -- MI355X-native Convex-Feasible-Set inner loop.

Drop-in for the hot path of JessicaLeu-code/MotionPlanning_5D_m (CFS_FANUC / PSGCFS_FANUC:
forward kinematics -> capsule/obstacle distance + Jacobian -> linearised constraints -> QP),
implemented as hand-written HIP kernels for gfx950 behind the C ABI of include/cfs_hip.h.
"""
from ._lib import CfsError, STATUS, device_count, lib  # noqa: F401  (imports torch first, see _lib)
from .robotproperty2 import robotproperty2, to_c_robot  # noqa: F401
from .sysinfo import (build_sys_info, cubic_resample, cylinder, line_reference, main_2L_problem,  # noqa: F401
                      main_FANUC_problem, RRTstar_CFS_problem)
from .solvers import CFS_FANUC, PSGCFS_FANUC, CHOMP_FANUC, CFSBatch, EVAL, dist_arm, obs_to_array  # noqa: F401
from .rrt import RRT_FANUC, RRTstar_problem, s_Parallel_rrt  # noqa: F401
from .mesh import Mesh, dist_arm_surf  # noqa: F401
from . import mesh  # noqa: F401

__all__ = ["CFS_FANUC", "PSGCFS_FANUC", "CHOMP_FANUC", "CFSBatch", "EVAL", "dist_arm", "robotproperty2", "build_sys_info",
           "line_reference", "cubic_resample", "cylinder", "main_FANUC_problem", "main_2L_problem",
           "RRTstar_CFS_problem", "RRT_FANUC", "s_Parallel_rrt", "RRTstar_problem", "CfsError", "STATUS", "device_count", "lib", "obs_to_array", "to_c_robot", "Mesh", "dist_arm_surf", "mesh"]
