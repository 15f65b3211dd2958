"""ScenarioMPC + Case_6 on the GPU against the oracle (flags, statuses, iteration counts exact;
trajectories to the QP tolerance), and the UQ-RMPC-vs-SCMPC comparison of Case_6 on one disturbance realisation."""
import numpy as np
import pytest

import fixtures
import oracle_ffi
from lbmpc import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from lbmpc import api, core, params
    core.init(0)
    return api, core, params


@pytest.mark.parametrize("system,N_SC,trials,steps", [("2x2", 50, 70, 12), ("3x3", 50, 33, 6), ("4x4", 7, 65, 4), ("2x2", 1, 3, 2)])
def test_scmpc_closed_loop_equals_oracle(env, system, N_SC, trials, steps):
    api, core, params = env
    P = fixtures.params_dict() if system == "2x2" else params.load(system)
    smp = _abi.make_sampler(**fixtures.sampler_args(P)) if system == "2x2" else params.sampler(P, oriented=True)
    nx, nu, N, nc = (int(P[k]) for k in ("nx", "nu", "N", "nc"))
    keep = _abi.make_scmpc_params(nx, nu, N, nc, N_SC, P["A"], P["B"], np.eye(nx), 0.1 * np.eye(nu), P["F"], P["G"])
    x0 = np.zeros(nx); x0[:2] = [-12.0, 5.0]
    kw = dict(trials=trials, steps=steps, seed=6, x0=x0, acc_bound=float(np.ravel(P["acc_bound"])[0]),
              distance_error=float(np.ravel(P["distance_error"])[0]), trial0=11, traj_trials=trials)
    rc, g = _abi.scmpc_mc_run(_abi.load(), "lbmpc_", keep, smp, **kw)
    rc2, o = _abi.scmpc_mc_run(oracle_ffi.load(), "orc_", keep, smp, **kw)
    assert rc == 0 and rc2 == 0
    for k in ("infeasible", "status", "qp_iters"):
        np.testing.assert_array_equal(g[k], o[k], err_msg=k)
    for k in ("x_final", "traj_x", "traj_u"):
        np.testing.assert_allclose(g[k], o[k], rtol=1e-6, atol=1e-6, err_msg=k)
    assert system != "2x2" or N_SC == 1 or g["qp_iters"].sum() > 0   # the wide bounds of 3x3 / 4x4 leave the QP unconstrained here


def test_case_6_comparison_on_one_realisation(env):
    """Case_6...:96-150: UQ-RMPC and SCMPC against the same plant disturbances (same seed and trial ids); both keep the
    constraints on this run, the mirrored ScenarioMPC.solve returns the first input of the batched run's QP form"""
    api, core, params = env
    P = fixtures.params_dict()
    from test_gpu_api import opts_from
    opts_UQMPC, opts_Car, _ = opts_from(api, P)
    opts_SC = dict(opts_Car, N_SC=50, Q=np.eye(2), R=0.1, N=8, F=P["F"], G=P["G"], nu=1, nx=2)
    K_N, n = 50, 16
    uq = api.monte_carlo(opts_UQMPC, opts_Car, N_pre_sam=400, N_MC=n, K_N=K_N, x_RM_0=[-12, 5], acc_bound=2.0,
                         distance_error=15.0, seed=6, record_trials=n)
    sc = api.scmpc_monte_carlo(opts_SC, N_MC=n, K_N=K_N, x_RM_0=[-12, 5], acc_bound=2.0, distance_error=15.0, seed=6,
                               record_trials=n)
    assert uq["Infeasible_Index"].sum() == 0 and uq["status"].max() == 0 and sc["status"].max() == 0
    # same realisation: w_k = x_{k+1} - A x_k - B u_k agrees between the two closed loops
    A, B = P["A"], np.reshape(P["B"], (2, 1))
    for r in (uq, sc):
        r["w"] = r["State_RM"][:, :, 1:] - np.einsum("ab,tbk->tak", A, r["State_RM"][:, :, :-1]) - np.einsum("aj,tjk->tak", B, r["Control_EV"])
    np.testing.assert_allclose(uq["w"], sc["w"], atol=1e-9)
    # both regulate the relative state towards the origin
    assert np.abs(uq["State_RM"][:, :, -1]).max() < 1.0 and np.abs(sc["State_RM"][:, :, -1]).max() < 1.5
    u = api.ScenarioMPC(opts_SC, seed=6).solve(np.array([-12.0, 5.0]))
    assert u.shape == (1, 1) and abs(u[0, 0]) <= 2.0 + 1e-9
