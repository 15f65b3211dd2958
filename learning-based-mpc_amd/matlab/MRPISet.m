% MRPISet.m -- drop-in for Functions_general/MRPISet.m: Fs = MRPISet(Ak, W, epsilon).
% (s, alpha) by the Rakovic form of the reference's loop (mode 2: alpha from h_W((Ak^s)' f_j), the form for which the
% truncated Minkowski sum is an epsilon-outer approximation and that matches the transposed sum evaluated below; the
% reference's literal loop, MRPISet.m:29,43 without the transpose, is mode 0), support values on the device; Fs is the
% H-rep template polytope {x : d_j'x <= h_S(d_j)} of the exact Minkowski-sum support instead of a hull of projected
% random samples.  Directions come from a seeded stream (the reference uses the unseeded global randn).
% Optional 5th argument `mode` selects the (s, alpha) loop: 2 (default) Rakovic form; 0 the reference's literal loop of
% the general tree (MRPISet.m:27-63: max form, A^i d without the transpose) -- the one that reproduces the (s, alpha)
% behind a stored parameter set (parameters.mat: s = 13, alpha = 0.0053213); 1 the literal sum form of the 2-D tree
% (Functions/MRPISet.m:8-17).
function Fs = MRPISet(Ak, W, epsilon, nTemplate, mode)
nx = size(Ak, 1);
if nargin < 5, mode = 2; end
rs = RandStream('mt19937ar', 'Seed', 0);
D = [eye(nx); -eye(nx)];
if nx > 2, D = [D; ones(1, nx)/sqrt(nx); -ones(1, nx)/sqrt(nx)]; end
extra = randn(rs, 10, nx);  D = [D; extra ./ vecnorm(extra, 2, 2)];
[s, alpha] = lbmpc_mex('mrpi_s_alpha', Ak, W.V, W.A, W.b, D', epsilon, mode);
if nargin < 4 || isempty(nTemplate), nTemplate = 64 * 2^nx; end        % rows of the template beyond +-I
T = randn(rs, nTemplate, nx);  T = [eye(nx); -eye(nx); T ./ vecnorm(T, 2, 2)];
h = lbmpc_mex('support_batch', Ak, W.V, s, alpha, T', 1);
Fs = Polyhedron('A', T, 'b', h);
end
