% Drop-in for Lib/RRT_FANUC.m over cfs_mex (see INTEGRATION.md): the whole tree grows on the GPU (cfs_rrt_grow).
% Same constructor and find_route() as Lib/RRT_FANUC.m:48,63; same outputs (route, all_nodes, total_dis, all_ee, fail, node_num).
% rand stays MATLAB's: find_route draws the uniforms here and the kernel consumes them exactly as Lib/RRT_FANUC.m:108,111 do
% (one per proposal for the goal bias, nstate more when the sample is random).
classdef RRT_FANUC
   properties
       obs cell; sys_info struct; goal; region_g; region_s; sample_off; ROBOT = 'M16iB'; SOLVER = 'RRT*'
       parent; newNode; all_nodes; total_dis; route; fail = 0; all_ee; MAX_ITER = 400; bi = 0.5; node_num = 1
   end
   methods
       function self = RRT_FANUC(val, val2, val3, val4, val5, val6, varargin)     % Lib/RRT_FANUC.m:48
            self.obs = val; self.sys_info = val2; self.goal = val3; self.region_g = val4; self.region_s = val5; self.sample_off = val6;
            if ~isempty(varargin), self.ROBOT = varargin{1}; self.SOLVER = varargin{2}; end
       end
       function self = find_route(self)                                            % Lib/RRT_FANUC.m:63
            ndraw = (1 + self.sys_info.nstate) * 8 * (self.MAX_ITER + 1);         % enough for 8 proposals per node
            [self.route, self.all_nodes, self.total_dis, self.all_ee, f, self.node_num] = cfs_mex('rrt', self.obs, self.sys_info, ...
                self.goal, self.region_g, self.region_s, self.sample_off, self.ROBOT, self.SOLVER, rand(ndraw, 1));
            self.fail = f ~= 0;
            if self.fail, disp('Failed to find path.'); end
            self.newNode = self.route(:, end);
       end
   end
end
