"""The benchmark variant of the oracle's closed loop (orc_set_certificate, bench.py cpu_baseline.with_certificate):
skipping the QP under the inscribed-ellipsoid certificate must not change any result.  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learning-based-mpc_amd"))
sys.path.insert(0, ROOT)

import oracle_core
from lbmpc import params


@pytest.mark.parametrize("system,x0", [("2x2", [-12.0, 5.0]), ("3x3", [-12.0, 5.0, 0.0])])
def test_certificate_changes_nothing(system, x0):
    import bench
    P = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz")) if system == "2x2" else params.load(system)
    smp = params.sampler(P, oriented=True)
    S_A, S_b = np.asarray(P["S_A"]), np.ravel(P["S_b"])
    ME = bench.inscribed_ellipsoid(S_A, S_b)
    # the ellipsoid is inside S: sqrt(a_j' ME^-1 a_j) <= b_j for every row
    supp = np.sqrt(np.einsum("ij,jk,ik->i", S_A, np.linalg.inv(ME), S_A))
    assert np.all(supp <= S_b * (1 + 1e-12))
    octx = oracle_core.Context(P)
    kw = dict(trials=24, steps=60, card_I0=10, seed=3, x0=np.asarray(x0), acc_bound=float(np.ravel(P["acc_bound"])[0]),
              distance_error=float(np.ravel(P["distance_error"])[0]), traj_trials=4)
    plain = octx.mc_run(smp, **kw)
    octx.set_certificate(ME)
    cert = octx.mc_run(smp, **kw)
    octx.set_certificate(None)
    again = octx.mc_run(smp, **kw)
    for k in plain:
        assert np.array_equal(plain[k], cert[k]), k
        assert np.array_equal(plain[k], again[k]), k
    assert plain["qp_iters"].sum() > 0
