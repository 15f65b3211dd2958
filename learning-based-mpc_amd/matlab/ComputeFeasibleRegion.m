% ComputeFeasibleRegion.m -- drop-in for Functions_general/ComputeFeasibleRegion.m: same constructor struct and
% method names.  Every LP the reference hands to IPOPT (Com_hs :447-503, Com_z :505-519, Com_hs_ini) is one batched
% call of lbmpc_mex('polytope_support', ...); the feasible region F_N = F_Ns (+) S is returned as the H-rep template
% polytope {x : d_j'x <= h_FNs(d_j) + h_S(d_j)} of its exact support, with one boundary point per direction in .V,
% instead of the hull of 5000 sampled projections (:153-203).  Mirrors lbmpc/api.py:ComputeFeasibleRegion, which is
% the tested form (no MATLAB in the build image).
classdef ComputeFeasibleRegion < handle
    properties (SetAccess = public)
        S; S_true; S_hat_opt; F; G; K; F_bar; Psi; nc; nx; nu; N; Ps; D; F_Com;
    end
    methods (Access = public)
        function obj = ComputeFeasibleRegion(parameters, D)
            f = {'S','S_true','S_hat_opt','F','G','K','F_bar','Psi','nc','nx','nu','N','Ps'};
            for i = 1:numel(f), if isfield(parameters, f{i}), obj.(f{i}) = parameters.(f{i}); end, end
            obj.F_Com = obj.F + obj.G*obj.K;
            if nargin < 2
                rs = RandStream('mt19937ar', 'Seed', 0);
                if obj.nx == 2
                    th = 2*pi*((0:359)' + 0.5)/360;  T = [cos(th) sin(th)];
                else
                    T = randn(rs, 64*2^obj.nx, obj.nx);  T = T ./ vecnorm(T, 2, 2);
                end
                D = [eye(obj.nx); -eye(obj.nx); T];
            end
            obj.D = D;
        end
        function hs = Com_hs(obj, S)                       % h_S(F_Com): one LP per row of F_Com
            hs = lbmpc_mex('polytope_support', S.A, S.b, obj.F_Com');
        end
        function H = rowsH(obj, Nu)
            H = [];  M = obj.F_bar;
            for i = 0:Nu, H = [H; M];  M = M*obj.Psi; end   %#ok<AGROW>
        end
        function Nu = ComNu_RMPC(obj, hs)                   % smallest Nu >= N with F_bar Psi^(Nu+1) z <= 1-hs on Omega(Nu)
            Nu = obj.N;
            while Nu < 100
                M = obj.F_bar*obj.Psi^(Nu+1);
                [goal, ~, st] = lbmpc_mex('polytope_support', obj.rowsH(Nu), repmat(1-hs, Nu+1, 1), M');
                if all(st == 0) && all(goal <= 1-hs), break; end
                Nu = Nu + 1;
            end
        end
        function [F_Ns, F_N] = region(obj, S, h, Nu)
            nz = size(obj.Psi, 1);  Cd = zeros(nz, size(obj.D, 1));  Cd(1:obj.nx, :) = obj.D';
            [hF, z] = lbmpc_mex('polytope_support', obj.rowsH(Nu), repmat(1-h(:), Nu+1, 1), Cd);
            [hS, e] = lbmpc_mex('polytope_support', S.A, S.b, obj.D');
            F_Ns = struct('A', obj.D, 'b', hF,      'V', z(1:obj.nx, :)');
            F_N  = struct('A', obj.D, 'b', hF + hS, 'V', z(1:obj.nx, :)' + e');
        end
        function [F_N, hs, Nu] = ComFeasibleRegion_RMPC(obj)
            hs = obj.Com_hs(obj.S);  Nu = obj.ComNu_RMPC(hs);  [~, F_N] = obj.region(obj.S, hs, Nu);
        end
        function [F_N, hs] = ComFeasibleRegion_True(obj)
            hs = obj.Com_hs(obj.S_true);  [~, F_N] = obj.region(obj.S_true, hs, obj.ComNu_RMPC(hs));
        end
        function [F_N, hs] = ComFeasibleRegion_UQOPT(obj)
            hs = obj.Com_hs(obj.S_hat_opt);  [~, F_N] = obj.region(obj.S_hat_opt, hs, obj.ComNu_RMPC(hs));
        end
        function [F_Ns, F_N] = ComFeasibleRegion_UQMPC(obj, S_hat, h_k)
            hs = obj.Com_hs(obj.S);  [F_Ns, F_N] = obj.region(S_hat, h_k, obj.ComNu_RMPC(hs));
        end
        function hs_ini = ComFeasibleRegion_UQMPC_Hs_ini(obj, S_hat_ini)
            hs_ini = obj.Com_hs(S_hat_ini);
        end
    end
end
