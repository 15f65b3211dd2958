"""SURVEY.md 8f-1 on the GPU: the polytope-support LP kernel against the oracle (bit-exact) and the mirrored
ComputeFeasibleRegion class; then config C3 AS WRITTEN: Case_5 on the 3x3 system from the near-boundary initial
state of Case_5...:25-47 (learn a set, S_hat_ini, hs_UQMPC, F_N_Hat, x_RM_0 from its coordinate extremes), 10 000
trials x 20 steps, flags == oracle."""
import numpy as np
import pytest

import fixtures
import oracle_core
from lbmpc import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from lbmpc import api, core, params
    core.init(0)
    return api, core, params


def fr_opts(P, api):
    S = api.Polyhedron(A=P["S_A"], b=P["S_b"])
    return dict(S=S, S_true=S, S_hat_opt=S, F=P["F"], G=P["G"], K=P["K"], F_bar=P["F_bar"], Psi=P["Psi"],
                nc=int(P["nc"]), nx=int(P["nx"]), nu=int(P["nu"]), N=int(P["N"]), Ps=P.get("Ps"))


@pytest.mark.parametrize("system", ["2x2", "3x3", "4x4"])
def test_polytope_support_kernel_equals_oracle(env, system):
    api, core, params = env
    P = fixtures.params_dict() if system == "2x2" else params.load(system)
    nx, nc, Nu = int(P["nx"]), int(P["nc"]), int(P["Nu"])
    fr = api.ComputeFeasibleRegion(fr_opts(P, api))
    H = fr._H(Nu)
    rng = np.random.default_rng(nx)
    nz = H.shape[1]
    nd = 333
    Cd = np.zeros((nz, nd))
    Cd[:nx] = rng.normal(size=(nx, nd))
    Cd[:, -7:] = rng.normal(size=(nz, 7))            # a few directions that involve the input variables too
    Cd[:, 5] = 0.0                                     # the zero direction
    hk = np.vstack([np.ravel(P["hs"]), 0.5 * np.ravel(P["hs"]), 0.9 * np.ravel(P["hs"])])
    b = np.tile(1 - hk, (1, Nu + 1))
    g, gs, gz = core.polytope_support(H, b, Cd, want_z=True)
    o, os_, oz = oracle_core.polytope_support(H, b, Cd, want_z=True)
    np.testing.assert_array_equal(gs, os_)
    np.testing.assert_array_equal(g, o)
    np.testing.assert_array_equal(gz, oz)
    assert gs.max() == 0 and (np.einsum("rk,sik->sir", H, gz) <= b[:, None, :] + 1e-9).all()
    # the tube set: nS rows in nx variables (795 / 1644 rows for 3x3 / 4x4)
    D = rng.normal(size=(nx, 200))
    g2, s2 = core.polytope_support(P["S_A"], np.ravel(P["S_b"]), D)
    o2, _ = oracle_core.polytope_support(P["S_A"], np.ravel(P["S_b"]), D)
    np.testing.assert_array_equal(g2, o2)
    assert s2.max() == 0
    with pytest.raises(core.LbmpcError, match="strictly feasible"):
        core.polytope_support(np.eye(2), np.array([1.0, 0.0]), np.eye(2))
    h, st = core.polytope_support(np.array([[1.0, 0.0]]), np.array([1.0]), np.array([[0.0], [1.0]]))
    assert st[0] == _abi.ST_LP_UNBOUNDED


def test_compute_feasible_region_mirror_2x2(env):
    api, core, params = env
    P = fixtures.params_dict()
    fr = api.ComputeFeasibleRegion(fr_opts(P, api))
    fo = api.ComputeFeasibleRegion(fr_opts(P, api), backend=oracle_core.polytope_support)
    F_N, hs, Nu = fr.ComFeasibleRegion_RMPC()
    F_o, hs_o, Nu_o = fo.ComFeasibleRegion_RMPC()
    assert Nu == Nu_o == 9
    np.testing.assert_array_equal(F_N.b, F_o.b)
    np.testing.assert_array_equal(hs, hs_o)
    V = P["F_N_RMPC_V"]
    assert (F_N.A @ V.T - F_N.b[:, None]).max() < 2e-7 and (F_N.b - (F_N.A @ V.T).max(1)).max() < 3e-7
    # the time-varying region of UQ-RMPC (:380-435) for a learned set: smaller tube, larger nominal region
    W = api.Polyhedron(A=P["W_A"], b=P["W_b"], V=P["W_V"])
    T = np.linalg.inv(np.eye(2) - P["Phi"])
    alpha, v = 0.3, np.array([0.004, -0.002])
    S_hat = alpha * fr.S + (1 - alpha) * (T @ v)
    hk = fr.ComFeasibleRegion_UQMPC_Hs_ini(S_hat)
    np.testing.assert_allclose(hk.ravel(), alpha * hs.ravel() + (1 - alpha) * fr.F_Com @ T @ v, rtol=1e-9)
    F_Ns_hat, F_N_hat = fr.ComFeasibleRegion_UQMPC(S_hat, hk)
    F_Ns, _ = fr._region(fr.S, hs, Nu)
    assert (F_Ns_hat.b >= F_Ns.b - 1e-12).all() and (F_Ns_hat.b > F_Ns.b + 1e-6).any()


def test_c3_case_5_as_written(env):
    """BASELINE config C3: 3x3, card(I0) = 10, x_RM_0 near the boundary of F_N_Hat (Case_5...:25-47), 10 000 trials x
    20 steps; Infeasible_Index, statuses and iteration counts == oracle on every trial"""
    api, core, params = env
    P = params.load("3x3")
    nx = 3
    smp = params.sampler(P, oriented=True)
    fr = api.ComputeFeasibleRegion(fr_opts(P, api))
    # IniSet.solve() once (Case_5...:29): information set 0 of the seed used below
    _, m, _ = core.sample_information_sets(smp, P["W_A"], 10, 1, seed=5, set0=0, want_samples=False)
    a0, v0, _, st = core.homothety_lp(P["W_A"], P["W_b"], m[0])
    assert st == 0
    T = np.linalg.inv(np.eye(nx) - P["Phi"])
    S_hat_ini = a0 * fr.S + (1 - a0) * (T @ v0)
    hs_UQMPC = fr.ComFeasibleRegion_UQMPC_Hs_ini(S_hat_ini)
    _, F_N_Hat = fr.ComFeasibleRegion_UQMPC(S_hat_ini, hs_UQMPC)
    x0 = api.case_5_initial_state(F_N_Hat.V)
    assert x0[0] < -1.0 and x0[1] > 0.1 and x0[2] == 0.0
    # the corner of the bounding box of F_N_Hat need not lie in F_N_Hat (that is what Case_5 probes)
    ctx, octx = core.Context(P), oracle_core.Context(P)
    kw = dict(trials=10000, steps=20, card_I0=10, seed=5, x0=x0, acc_bound=float(P["acc_bound"][0]),
              distance_error=float(P["distance_error"][0]))
    g = ctx.mc_run(smp, **kw)
    o = octx.mc_run(smp, threads=8, **kw)
    for k in ("infeasible", "status", "qp_iters"):
        np.testing.assert_array_equal(g[k], o[k], err_msg=k)
    np.testing.assert_allclose(g["alpha_final"], o["alpha_final"], rtol=1e-9, atol=0)
    np.testing.assert_allclose(g["x_final"], o["x_final"], rtol=1e-6, atol=1e-6)
    assert g["qp_iters"].min() > 0
    print("C3 as written: x_RM_0 =", x0, "violating trials:", int(g["infeasible"].sum()), "QP-infeasible:",
          int((g["status"] == _abi.ST_QP_INFEASIBLE).sum()))
    ctx.close(); octx.close()
