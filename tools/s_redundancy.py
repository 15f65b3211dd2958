"""How many rows of the committed 3x3 / 4x4 tube sets are irredundant (api.minHRep on the device) -- diagnostic."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "learning-based-mpc_amd"))
from lbmpc import api, params
for sysname in ("3x3", "4x4"):
    P = params.load(sysname)
    S = api.Polyhedron(A=P["S_A"], b=np.ravel(P["S_b"]))
    Q, keep = api.minHRep(S, return_keep=True)
    print(sysname, "S rows", len(S.b), "irredundant", len(keep))
