"""How many rows of the tube set's H-rep template survive minHRep, against the template size (3x3 / 4x4).  Diagnostic
behind the choice of n_template in lbmpc/pipeline.py (every committed tube-set row must be a facet)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learning-based-mpc_amd"))
from lbmpc import api, core, pipeline
from scipy.linalg import solve_discrete_are
core.init(0)
for nx, sizes in ((3, (3000, 10000, 30000, 60000)), (4, (3000, 6000, 12000, 24000))):
    d = pipeline.definitions(nx)
    A, B = d["A"], d["B"]
    Px = solve_discrete_are(A, B, d["Q"], d["R"])
    K = -np.linalg.solve(d["R"] + B.T @ Px @ B, B.T @ Px @ A)
    Phi = A + B @ K
    W = api.minHRep(api.hull(d["W_V"]))
    for n in sizes:
        D = api.template_directions(nx, n, seed=0)
        t0 = time.perf_counter()
        S, info = api.MRPISet(Phi, W, 0.1, template_directions=D, return_info=True)
        Q, keep = api.minHRep(S, return_keep=True)
        print(f"nx={nx} template {len(D)} -> {len(keep)} facets  (s={info['s']}, {time.perf_counter()-t0:.2f} s)", flush=True)
