% Drop-in for Lib/PSGCFS_FANUC.m over cfs_mex: mode 1, margin obs{j}.D, explicit noise.
classdef PSGCFS_FANUC
   properties
       obs cell; sys_info struct; nn; ROBOT = 'M16iB'; u; x_; Ainq; binq; eval EVAL; iter_O = 1; total_iter = 0; status
   end
   methods
       function self = PSGCFS_FANUC(val, val2, varargin)            % same signature as Lib/PSGCFS_FANUC.m:40
            self.obs = val; self.sys_info = val2; self.nn = val2.H*val2.nu;
            if ~isempty(varargin), self.ROBOT = varargin{1}; end
            self.x_ = val2.x_; self.u = zeros(self.nn,1); self.eval = EVAL(val2);
       end
       function self = get_con(self)                              % public Ainq / binq, reference row order (Lib/PSGCFS_FANUC.m get_con)
            [self.Ainq, self.binq] = cfs_mex('get_con', 1, self.obs, self.sys_info, self.ROBOT, self.x_, self.u);
       end
       function self = optimizer(self)                            % one MEX call instead of the MATLAB loop
            [self.u, self.x_, c, ec, eu, self.iter_O, self.total_iter, self.status] = ...
                cfs_mex('solve', 1, self.obs, self.sys_info, self.ROBOT, 0.1*randn(self.nn, self.sys_info.MAX_O_ITER));
            n = self.iter_O - 1;
            self.eval.cost_all = c(1:n)'; self.eval.e_cost_all = ec(1:n)'; self.eval.e_u_all = eu(1:n)';
            if n > 0, self.eval.cost_new = c(n); end
            self.eval.x_ = self.x_;
       end
   end
end
