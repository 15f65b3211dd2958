% lbmpc_poly.m -- turn the {A, b[, V]} structs that lbmpc/matio.py writes (an MPT3 Polyhedron is a MATLAB class object
% that only MATLAB can serialise) back into Polyhedron objects; cell arrays are converted element-wise.
function P = lbmpc_poly(s)
if iscell(s)
    P = cellfun(@lbmpc_poly, s, 'UniformOutput', false);
elseif isfield(s, 'A')
    P = Polyhedron('A', s.A, 'b', s.b);
else
    P = Polyhedron(s.V);
end
end
