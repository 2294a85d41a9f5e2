% Cost_b = EVAL_get_Cost_b(sys_info, ROBOT) -- EVAL(sys_info).get_Cost_b() (Lib/EVAL.m:75-78, called at main_FANUC.m:131-132)
% over cfs_mex: cost of the unconstrained minimiser -H^{-1} ff, both products on the GPU's matrix cores (cfs_cost_b).
% The reference's own EVAL.m keeps working unchanged (it calls quadprog); this is the library-side equivalent.
function Cost_b = EVAL_get_Cost_b(sys_info, ROBOT)
    Cost_b = cfs_mex('cost_b', sys_info, ROBOT);
end
