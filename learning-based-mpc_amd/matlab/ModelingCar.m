% ModelingCar.m -- drop-in for Functions_general/ModelingCar.m (plant of the Monte-Carlo scripts, Case_3...:58,75):
% same constructor struct and the same [xi, x_next] = EVModeling(x, u) / LVModeling(x) / polytope_sample_EV / _LV
% methods.  The rejection sampling runs in liblbmpc_hip.so through lbmpc_mex('sample_polytope', ...), on the library's
% Philox stream instead of MATLAB's global rand: with parameters.seed = s and parameters.stream = t the k-th step
% draws exactly what trial t of lbmpc_mex('mc_run', ...) with seed s draws at step k (axis-aligned proposal boxes),
% so a script-driven closed loop reproduces the batched kernel's trajectory.  Both fields are optional (default 0).
classdef ModelingCar < handle
    properties (SetAccess = public)
        A; B;
        Xi_true_EV; Xi_true_EV_A; Xi_true_EV_b;
        Xi_true_LV; Xi_true_LV_A; Xi_true_LV_b;
        max_u_LV; min_u_LV;
        seed = 0; stream = 0;   % Philox stream of this plant
        k = 0;                  % closed-loop step of the next EVModeling / LVModeling draw
        n_bulk = 0;             % points already handed out by polytope_sample_EV / _LV
    end
    methods (Access = public)
        function obj = ModelingCar(parameters)
            obj.A = parameters.A;  obj.B = parameters.B;
            obj.Xi_true_EV = parameters.Xi_true_EV;  obj.Xi_true_LV = parameters.Xi_true_LV;
            obj.Xi_true_EV_A = obj.Xi_true_EV.A;  obj.Xi_true_EV_b = obj.Xi_true_EV.b;
            obj.Xi_true_LV_A = obj.Xi_true_LV.A;  obj.Xi_true_LV_b = obj.Xi_true_LV.b;
            obj.max_u_LV = parameters.max_u_LV;  obj.min_u_LV = parameters.min_u_LV;
            if isfield(parameters, 'seed'),   obj.seed = parameters.seed;     end
            if isfield(parameters, 'stream'), obj.stream = parameters.stream; end
        end
        function [xi_EV_k, x_EV_k_next] = EVModeling(obj, x_EV_k, u_EV_k)
            % purpose 5 = step draw of xi_EV; the step index advances here, after the LV draws of the same step
            xi_EV_k = obj.draw(obj.Xi_true_EV_A, obj.Xi_true_EV_b, obj.Xi_true_EV.V, 1, obj.k, 5);
            obj.k = obj.k + 1;
            x_EV_k_next = obj.A * x_EV_k + obj.B * u_EV_k + xi_EV_k;
        end
        function [xi_LV_k, x_LV_k_next] = LVModeling(obj, x_LV_k)
            % purpose 3 = step draw of xi_LV, purpose 4 = u_LV ~ U[min_u_LV, max_u_LV] (the interval as a 1-D polytope)
            xi_LV_k = obj.draw(obj.Xi_true_LV_A, obj.Xi_true_LV_b, obj.Xi_true_LV.V, 1, obj.k, 3);
            u_LV = obj.draw([1; -1], [obj.max_u_LV; -obj.min_u_LV], [obj.min_u_LV; obj.max_u_LV], 1, obj.k, 4);
            x_LV_k_next = obj.A * x_LV_k + xi_LV_k + obj.B * u_LV;
        end
        function EV_samples = polytope_sample_EV(obj, N_sam)
            EV_samples = obj.draw(obj.Xi_true_EV_A, obj.Xi_true_EV_b, obj.Xi_true_EV.V, N_sam, obj.n_bulk, 6);
            obj.n_bulk = obj.n_bulk + N_sam;
        end
        function LV_samples = polytope_sample_LV(obj, N_sam)
            LV_samples = obj.draw(obj.Xi_true_LV_A, obj.Xi_true_LV_b, obj.Xi_true_LV.V, N_sam, obj.n_bulk, 7);
            obj.n_bulk = obj.n_bulk + N_sam;
        end
    end
    methods (Access = private)
        function x = draw(obj, A, b, V, n, index0, purpose)
            [x, status] = lbmpc_mex('sample_polytope', A, b, min(V, [], 1)', max(V, [], 1)', n, obj.seed, obj.stream, index0, purpose);
            if any(status), warning('ModelingCar:sampler', 'rejection sampler hit its attempt cap for %d of %d points', nnz(status), n); end
        end
    end
end
