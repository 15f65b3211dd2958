% ScenarioMPC.m -- drop-in for Functions_general/ScenarioMPC.m: same constructor struct (A, B, N_SC, Q, R, N, F, G, nu,
% nx, min_u_LV, max_u_LV, Xi_true_EV, Xi_true_LV; :30-48).  solve(x_k) draws N_SC x N fresh scenarios on the device
% and returns u_k (:52-58); monte_carlo(cfg) is the whole loop of Cases_general/Case_6_Compare_With_SCMPC.m:96-150 in
% one call.  Mirrors lbmpc/api.py:ScenarioMPC, which is the tested form (no MATLAB in the build image).
classdef ScenarioMPC < handle
    properties (SetAccess = public)
        p; sampler; calls = 0;
    end
    methods (Access = public)
        function obj = ScenarioMPC(parameters)
            q = parameters;
            obj.p = struct('N', q.N, 'N_SC', q.N_SC, 'A', q.A, 'B', q.B, 'Q', q.Q, 'R', q.R, 'F', q.F, 'G', q.G);
            EV = q.Xi_true_EV;  LV = q.Xi_true_LV;
            obj.sampler = struct('nx', q.nx, 'nu', q.nu, 'EV_A', EV.A, 'EV_b', EV.b, 'EV_lo', min(EV.V, [], 1)', ...
                'EV_hi', max(EV.V, [], 1)', 'LV_A', LV.A, 'LV_b', LV.b, 'LV_lo', min(LV.V, [], 1)', ...
                'LV_hi', max(LV.V, [], 1)', 'B', q.B, 'min_u_LV', q.min_u_LV, 'max_u_LV', q.max_u_LV);
        end
        function u_k = solve(obj, x_k)                      % one closed-loop step of one trial: u_0 of the scenario QP
            cfg = struct('trials', 1, 'trial0', obj.calls, 'steps', 1, 'seed', 0, 'x0', x_k(:), 'acc_bound', inf, ...
                         'distance_error', inf, 'traj_trials', 1);
            obj.calls = obj.calls + 1;
            R = lbmpc_mex('scmpc_mc_run', obj.p, obj.sampler, cfg);
            u_k = R.Control_EV_SCMPC(:, 1);
        end
        function R = monte_carlo(obj, cfg)                  % cfg: trials, trial0, steps, seed, x0, acc_bound, distance_error
            R = lbmpc_mex('scmpc_mc_run', obj.p, obj.sampler, cfg);
        end
    end
end
