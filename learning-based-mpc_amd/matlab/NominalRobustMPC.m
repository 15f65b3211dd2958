% NominalRobustMPC.m -- drop-in for Functions_general/NominalRobustMPC.m (fixed S and hs: alpha = 1, v = 0, no update).
classdef NominalRobustMPC < handle
    properties (SetAccess = public)
        dims; h; nx;
    end
    methods (Access = public)
        function obj = NominalRobustMPC(parameters)
            p = rmfield(parameters, intersect(fieldnames(parameters), {'S','E','W','Ps','num_half_space_S'}));
            p.S_A = parameters.S.A;  p.S_b = parameters.S.b;  obj.nx = double(p.nx);
            if isfield(parameters, 'W'), p.W_A = parameters.W.A; p.W_b = parameters.W.b;
            else, p.W_A = [eye(obj.nx); -eye(obj.nx)]; p.W_b = ones(2*obj.nx, 1); end
            for f = {'A','B','Phi','F','G'}   % fields the nominal controller's struct may lack
                if ~isfield(p, f{1}), p.(f{1}) = zeros(obj.nx); end
            end
            obj.dims = double([p.nx p.nu p.nc numel(p.S_b) numel(p.W_b)]);
            obj.h = lbmpc_mex('create', p);
        end
        function [s_k_opt, u_k] = solve(obj, x_k)
            [s_k_opt, u_k] = lbmpc_mex('step', obj.h, 0, obj.dims, x_k, 1, zeros(obj.nx, 1), zeros(obj.nx, 1));
        end
        function delete(obj)
            if ~isempty(obj.h), lbmpc_mex('destroy', obj.h); end
        end
    end
end
