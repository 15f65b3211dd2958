#!/usr/bin/env python3
"""Per-solve statistics of the tube QP along closed-loop trajectories (CPU, diagnostic build of the oracle with
-DORC_STATS): final / maximum active-set size, iterations, full scans, drops, active H rows.  Guides the solver design.
  gcc -O2 -ffp-contract=off -mfma -fPIC -fopenmp -std=c11 -DORC_STATS -shared -o /tmp/liborc_stats.so oracle/lbmpc_oracle.c -lm
  python tools/qp_stats.py C1|C3|C4 [trials]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "learning-based-mpc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import oracle_ffi
oracle_ffi.ORACLE_SO = os.environ.get("ORC_STATS_SO", "/tmp/liborc_stats.so")
import oracle_core
from lbmpc import params

which = sys.argv[1] if len(sys.argv) > 1 else "C1"
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if which == "C1":
    P = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz"))
    x0, steps, card, seed = [-12.0, 5.0], 50, 100, 1
elif which == "C3":
    P = params.load("3x3")
    # Case_5 near-boundary start: api.case_5_start needs the GPU; take it from the environment (tools/qp_legs.py prints it)
    x0, steps, card, seed = [float(t) for t in os.environ["X03"].split(",")], 20, 10, 5
else:
    P = params.load("4x4")
    x0, steps, card, seed = [-12.0, 5.0, 0.0, 0.0], 32, 10, 4
lib = oracle_ffi.load()
lib.orc_stats_reset()
o = oracle_core.Context(P).mc_run(params.sampler(P, oriented=True), trials=trials, steps=steps, card_I0=card, seed=seed,
                                  x0=np.asarray(x0), acc_bound=float(np.ravel(P["acc_bound"])[0]),
                                  distance_error=float(np.ravel(P["distance_error"])[0]))
H = (C.c_longlong * (8 * 64)).in_dll(lib, "orc_stats_hist")
H = np.array(H[:]).reshape(8, 64)
names = ["final nq", "max nq", "iterations", "full scans", "drops", "final H rows"]
print(which, "trials", trials, "QPs", int(H[0].sum()), "per trial", H[0].sum() / trials)
for i, n in enumerate(names):
    nzi = np.nonzero(H[i])[0]
    print(f"{n:14s}", {int(k): int(H[i][k]) for k in nzi}, "mean %.2f" % ((H[i] * np.arange(64)).sum() / max(1, H[i].sum())))
