% [dis, points] = point2surface_dis(p, obs) -- the function M200i/dist_arm_surf_200i.m:21 and
% Lib/functions/dist_arm_surface.m:43 call but the reference does not contain; here over libcfs_hip.so.
% p = pos{i}.p (3x2 link axis), obs.handle = cfs_mex('mesh_load_stl', file, scale, map_from_stl).
function [dis, points] = point2surface_dis(p, obs)
    [dis, pts] = cfs_mex('mesh_segment_distance', obs.handle, [p(:,1); p(:,2)]);
    points = reshape(pts, 3, 2);   % [closest point on the link axis, closest point on the mesh]
end
