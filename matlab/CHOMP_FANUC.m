% Drop-in for Lib/CHOMP_FANUC.m over cfs_mex (see INTEGRATION.md section 6); constructed by Lib/functions/s_Solver.m:12-21.
classdef CHOMP_FANUC
   properties
       obs cell; sys_info struct; nn; ROBOT = 'M16iB'; u; x_; eval EVAL; iter_O = 1; total_iter = 0
   end
   methods
       function self = CHOMP_FANUC(val, val2, uu, varargin)       % same signature as Lib/CHOMP_FANUC.m:34
            self.obs = val; self.sys_info = val2; self.nn = val2.H*val2.nu;
            if ~isempty(varargin), self.ROBOT = varargin{1}; end
            self.x_ = val2.x_; self.u = uu; self.eval = EVAL(val2);
       end
       function self = optimizer(self)                            % one MEX call: cfs_chomp_batch (Lib/CHOMP_FANUC.m:54-69)
            [self.u, self.x_, c, ec, eu, self.iter_O] = cfs_mex('chomp', self.obs, self.sys_info, self.ROBOT, self.u);
            n = self.iter_O - 1;
            self.eval.cost_all = c(1:n)'; self.eval.e_cost_all = ec(1:n)'; self.eval.e_u_all = eu(1:n)';
            if n > 0, self.eval.cost_new = c(n); end
            disp('MAX_ITER')                                       % EVAL.m:70: eval.x_ is never refreshed, the loop always runs out
       end
   end
end
