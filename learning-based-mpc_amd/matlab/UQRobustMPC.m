% UQRobustMPC.m -- drop-in for Functions_general/UQRobustMPC.m: same constructor struct and solve() signature;
% the 2+nc IPOPT solves per step are replaced by one call into liblbmpc_hip.so through lbmpc_mex.
classdef UQRobustMPC < handle
    properties (SetAccess = public)
        W; S; K; dims; h;   % h: uint64 handle of the device-resident controller context
    end
    methods (Access = public)
        function obj = UQRobustMPC(parameters)
            obj.W = parameters.W;  obj.S = parameters.S;  obj.K = parameters.K;
            p = rmfield(parameters, intersect(fieldnames(parameters), {'W','S','E','Ps','num_half_space_S'}));
            p.W_A = obj.W.A;  p.W_b = obj.W.b;  p.S_A = obj.S.A;  p.S_b = obj.S.b;
            obj.dims = double([p.nx p.nu p.nc numel(p.S_b) numel(p.W_b)]);
            obj.h = lbmpc_mex('create', p);
        end
        function [s_k_opt, u_k, alpha_k, v_k, hk, S_hat_star, W_hat] = solve(obj, x_k, alpha_before, v_before, w_new)
            [s_k_opt, u_k, alpha_k, v_k, hk, Sb, Wb] = lbmpc_mex('step', obj.h, 1, obj.dims, x_k, alpha_before, v_before, w_new);
            S_hat_star = Polyhedron('A', obj.S.A, 'b', Sb);   % callers use .A/.b/.volume/+ on these
            W_hat      = Polyhedron('A', obj.W.A, 'b', Wb);
        end
        function delete(obj)
            if ~isempty(obj.h), lbmpc_mex('destroy', obj.h); end
        end
    end
end
