% InitialSetComputation.m -- drop-in for Functions_general/InitialSetComputation.m: sampling of the information set
% and the (alpha, v) LP run on the device (halfspace max-reduction + closed-form homothety LP).
classdef InitialSetComputation < handle
    properties (SetAccess = public)
        N_pre_sam; W; smp; seed = 0; calls = 0;
    end
    methods (Access = public)
        function obj = InitialSetComputation(o)
            obj.N_pre_sam = o.N_pre_sam;  obj.W = o.W;
            EV = o.Xi_true_EV;  LV = o.Xi_true_LV;
            obj.smp = struct('nx', o.nx, 'nu', size(o.B, 2), 'B', o.B, 'min_u_LV', o.min_u_LV, 'max_u_LV', o.max_u_LV, ...
                'EV_A', EV.A, 'EV_b', EV.b, 'EV_lo', min(EV.V)', 'EV_hi', max(EV.V)', ...
                'LV_A', LV.A, 'LV_b', LV.b, 'LV_lo', min(LV.V)', 'LV_hi', max(LV.V)');
        end
        function [alpha_ini, v_ini, samplesinput] = solve(obj)
            samplesinput = lbmpc_mex('sample_sets', obj.smp, obj.W.A, obj.N_pre_sam, obj.seed, obj.calls);
            obj.calls = obj.calls + 1;
            [alpha_ini, v_ini] = lbmpc_mex('learn_set', obj.W.A, obj.W.b, samplesinput);
        end
        function [alpha_opt, v_opt] = solve_opt(obj, samples_opt)
            [alpha_opt, v_opt] = lbmpc_mex('learn_set', obj.W.A, obj.W.b, samples_opt);
        end
    end
end
