% MRPISet.m -- drop-in for Functions_general/MRPISet.m: Fs = MRPISet(Ak, W, epsilon).
% (s, alpha) by the reference's loop, support values on the device; Fs is the H-rep template polytope
% {x : d_j'x <= h_S(d_j)} of the exact Minkowski-sum support instead of a hull of projected random samples.
function Fs = MRPISet(Ak, W, epsilon)
nx = size(Ak, 1);
D = [eye(nx); -eye(nx)];
if nx > 2, D = [D; ones(1, nx)/sqrt(nx); -ones(1, nx)/sqrt(nx)]; end
extra = randn(10, nx);  D = [D; extra ./ vecnorm(extra, 2, 2)];
[s, alpha] = lbmpc_mex('mrpi_s_alpha', Ak, W.V, W.A, W.b, D', epsilon, 0);
nT = [2 128 795 1644];  T = randn(nT(min(nx, 4)), nx);  T = [eye(nx); -eye(nx); T ./ vecnorm(T, 2, 2)];
h = lbmpc_mex('support_batch', Ak, W.V, s, alpha, T', 1);
Fs = Polyhedron('A', T, 'b', h);
end
