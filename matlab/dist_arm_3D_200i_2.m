% [d, linkid] = dist_arm_3D_200i_2(theta, base, obs, robot) -- drop-in for Lib/200i/dist_arm_3D_200i_2.m over cfs_mex:
% the M200i arm-to-line-obstacle distance (theta(2) - pi/2 offset :11, near-zero surrogate :22-24, first minimum :25-28)
% evaluated by the GPU geometry kernel (cfs_dist_arm).  With this file ahead of Lib/200i on the path
% CFS_FANUC.get_con (Lib/CFS_FANUC.m:115) and any feasibility check written over dist_arm reach the kernel unchanged
% (Lib/RRT_FANUC.m:146-181 inlines the same CapPos + distLinSeg loop; its drop-in is matlab/RRT_FANUC.m).
% theta may be njoint x N (N poses at once: d, linkid are then 1 x N); base is taken from robot.base as the kernel does.
function [d, linkid] = dist_arm_3D_200i_2(theta, base, obs, robot) %#ok<INUSL>
    [d, linkid] = cfs_mex('dist_arm', theta, obs, robot, 'M200i');
end
