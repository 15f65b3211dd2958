"""The host-side mirror of the MATLAB entry points (lbmpc.api), driven the way the Cases_general scripts drive the
reference, on the GPU."""
import numpy as np
import pytest

import fixtures
import oracle_core
from lbmpc import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from lbmpc import api as a, core
    core.init(0)
    return a


@pytest.fixture(scope="module")
def P():
    return fixtures.params_dict()


def opts_from(api, P):
    W = api.Polyhedron(A=P["W_A"], b=P["W_b"], V=P["W_V"])
    S = api.Polyhedron(A=P["S_A"], b=P["S_b"], V=P["S_V"])
    EV = api.Polyhedron(A=P["Xi_EV_H"][:, :-1], b=P["Xi_EV_H"][:, -1], V=P["Xi_EV_V"])
    LV = api.Polyhedron(A=P["Xi_LV_H"][:, :-1], b=P["Xi_LV_H"][:, -1], V=P["Xi_LV_V"])
    opts_UQMPC = {k: P[k] for k in ("N", "nx", "nu", "nc", "Nu", "A", "B", "F", "G", "K", "Px", "Pc", "F_bar", "Psi",
                                    "Phi", "E", "hs", "Ps", "num_half_space_S")}
    opts_UQMPC.update(W=W, S=S)
    opts_Car = dict(A=P["A"], B=P["B"], min_u_LV=P["min_u_LV"], max_u_LV=P["max_u_LV"], Xi_true_EV=EV, Xi_true_LV=LV)
    opts_ini = dict(B=P["B"], N_pre_sam=100, N_sam_opt=int(P["N_sam_opt"]), nx=2, W=W, min_u_LV=P["min_u_LV"],
                    max_u_LV=P["max_u_LV"], Xi_true_EV=EV, Xi_true_LV=LV)
    return opts_UQMPC, opts_Car, opts_ini


def test_case_3_script_loop_equals_batched_kernel(api, P):
    """Case_3...:24-88 written out with the mirrored classes (one solve per step, plant stepped by ModelingCar in
    absolute coordinates) reproduces the batched closed-loop kernel's trajectory of the same RNG stream."""
    opts_UQMPC, opts_Car, opts_ini = opts_from(api, P)
    K_N, trial, seed = 25, 3, 11
    mc = api.monte_carlo(opts_UQMPC, opts_Car, N_pre_sam=100, N_MC=trial + 1, K_N=K_N, x_RM_0=[-12, 5], acc_bound=2.0,
                         distance_error=15.0, seed=seed, record_trials=trial + 1)
    mpc = api.UQRobustMPC(opts_UQMPC)
    Car = api.ModelingCar(opts_Car, seed=seed, stream=trial)
    A, B, x_des = P["A"], P["B"], P["x_des"].reshape(-1, 1)
    x_LV = np.array([[100.0], [10.0]])
    x_EV = x_LV + x_des + np.array([[-12.0], [5.0]])
    alpha_k, v_k = float(mc["alpha_ini"][trial]), None
    # the information set of trial `trial` is stream (seed, trial): learn it through InitialSetComputation
    ini = api.InitialSetComputation(opts_ini, seed=seed)
    api.InitialSetComputation._calls = iter(range(trial, 10 ** 9))
    a0, v0, samples = ini.solve()
    assert a0 == alpha_k and samples.shape == (2, 100)
    v_k = v0
    X = [x_EV - x_LV - x_des]
    U = []
    for k in range(K_N):
        xi_LV, x_LV_next = Car.LVModeling(x_LV)
        w_new = np.zeros((2, 1)) if k == 0 else X[k] - A @ X[k - 1] - B @ U[k - 1]
        s_k, u_k, alpha_k, v_k, hk, S_hat, W_hat = mpc.solve(X[k], alpha_k, v_k, w_new)
        xi_EV, x_EV = Car.EVModeling(x_EV, u_k)
        x_LV = x_LV_next
        U.append(u_k)
        X.append(x_EV - x_LV - x_des)
        assert S_hat.A.shape == (80, 2) and W_hat.b.shape == (4,)
    X, U = np.hstack(X), np.hstack(U)
    np.testing.assert_allclose(X, mc["State_RM"][trial], rtol=0, atol=1e-9)
    np.testing.assert_allclose(U, mc["Control_EV"][trial], rtol=0, atol=1e-9)
    assert abs(alpha_k - mc["alpha_K"][trial]) < 1e-12


def test_initial_set_solve_opt_golden(api, P):
    _, _, opts_ini = opts_from(api, P)
    st = fixtures.npz("stats.npz")
    a, v = api.InitialSetComputation(opts_ini).solve_opt(P["W_true_V"].T)
    assert abs(a - st["2x2_alpha_opt"][0]) < 1e-7 and abs(v[1, 0] - st["2x2_v_opt"][1]) < 1e-8


def test_mrpiset_template_polytope(api, P):
    """MRPISet drop-in: H-rep template polytope of the exact support.  Geometric parity with the reference's stored S
    (an inner approximation of S(s, alpha) by sampled projections): every stored vertex lies inside the template, and
    the tightening h_S(F_Com) agrees with the stored hs within the epsilon-level slack (1 %)."""
    W = api.Polyhedron(A=P["W_A"], b=P["W_b"], V=P["W_V"])
    # the stored S was built from the reference's literal loop (MRPISet.m:27-63, mode 0): with its (s, alpha) every
    # stored vertex lies inside the template
    Fs, info = api.MRPISet(P["Phi"], W, 0.01, mode=_abi.MS_MAX_LITERAL, return_info=True)
    assert info["status"] == 0 and info["s"] >= 11
    viol = (Fs.A @ P["S_V"].T - Fs.b[:, None]).max()
    assert viol < 1e-3 * np.abs(P["S_V"]).max()
    # the default (Rakovic form) is another epsilon-outer approximation of the same mRPI set: the two agree within
    # the epsilon level (1 % of the set's size)
    Fr, ir = api.MRPISet(P["Phi"], W, 0.01, return_info=True)
    assert ir["status"] == 0
    assert (Fr.A @ P["S_V"].T - Fr.b[:, None]).max() < 1e-2 * np.abs(P["S_V"]).max()
    Fc = P["F"] + P["G"] @ P["K"]
    from scipy.optimize import linprog
    hs = np.array([-linprog(-Fc[i], A_ub=Fs.A, b_ub=Fs.b, bounds=[(None, None)] * 2, method="highs").fun for i in range(4)])
    np.testing.assert_allclose(hs, P["hs"].ravel(), rtol=1.5e-2)


def test_nominal_rmpc_mirror(api, P):
    opts_UQMPC, _, _ = opts_from(api, P)
    rmpc = api.NominalRobustMPC(opts_UQMPC)
    octx = oracle_core.Context(P)
    for x in ([-12.0, 5.0], [3.0, -1.0], [0.1, 0.0]):
        s, u = rmpc.solve(np.array(x))
        o = octx.step(np.array(x), 1.0, np.zeros(2), None, uq_update=False)
        np.testing.assert_allclose(s.ravel(), o["s0"], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(u.ravel(), o["u"], rtol=1e-6, atol=1e-9)


def test_sample_polytope_vs_oracle(api, P):
    import oracle_ffi
    from lbmpc import core
    from lbmpc._abi import dptr, fcol, iptr
    lib, orc = core.init(0), oracle_ffi.load()
    H, V = P["Xi_EV_H"], P["Xi_EV_V"]
    A, b, lo, hi = fcol(H[:, :-1]), fcol(H[:, -1]), fcol(V.min(0)), fcol(V.max(0))
    n = 5000
    out, out2 = np.zeros((n, 2)), np.zeros((n, 2))
    st, st2 = np.zeros(n, np.int32), np.zeros(n, np.int32)
    assert lib.lbmpc_sample_polytope(dptr(A), dptr(b), 4, 2, dptr(lo), dptr(hi), n, 7, 99, 5, 5, 1000, dptr(out), iptr(st)) == 0
    assert orc.orc_sample_polytope(dptr(A), dptr(b), 4, 2, dptr(lo), dptr(hi), n, 7, 99, 5, 5, 1000, dptr(out2), iptr(st2)) == 0
    np.testing.assert_array_equal(out, out2)
    np.testing.assert_array_equal(st, st2)
    assert (out @ H[:, :-1].T <= H[:, -1] + 1e-15).all()
    # uniformity: the sample mean approaches the polygon's centroid
    x, y = V[:, 0], V[:, 1]
    x1, y1 = np.roll(x, -1), np.roll(y, -1)
    cr = x * y1 - x1 * y
    cen = np.array([((x + x1) * cr).sum(), ((y + y1) * cr).sum()]) / (3 * cr.sum())
    assert np.abs(out.mean(0) - cen).max() < 2e-3


@pytest.mark.parametrize("system", ["3x3", "4x4"])
def test_mrpiset_is_an_outer_template_on_generated_systems(api, system):
    """MRPISet on the 3x3 / 4x4 systems: (s, alpha) in the Rakovic form (the default), and the returned template
    polytope {x : d_j'x <= h_j} contains the set whose support the oracle evaluates: for 1e5 random directions d the
    support of the template polytope's defining set, h_S(d) from the oracle, never exceeds what the template allows --
    checked through its dual: every template row's h_j equals the oracle's h_S(d_j), and h_S is sublinear, so
    h_S(d) <= sum_j lam_j h_j for any d = sum_j lam_j d_j, lam >= 0 (verified on random conic combinations)"""
    from lbmpc import params
    Pg = params.load(system)
    nx = int(Pg["nx"])
    W = api.Polyhedron(A=Pg["W_A"], b=Pg["W_b"], V=Pg["W_V"])
    Fs, info = api.MRPISet(Pg["Phi"], W, 0.1, return_info=True)
    assert info["status"] == 0 and 0.0 <= info["alpha"] < 1.0
    s, alpha = info["s"], info["alpha"]
    assert (s, alpha) == oracle_core.mrpi_s_alpha(Pg["Phi"], Pg["W_V"], Pg["W_A"], Pg["W_b"],
                                                  api.reference_directions(nx, np.random.default_rng(0)).T, 0.1, mode=_abi.MS_RAKOVIC)[:2]
    # every row is tight: b_j = h_S(d_j) (oracle)
    np.testing.assert_allclose(Fs.b, oracle_core.support_batch(Pg["Phi"], Pg["W_V"], s, alpha, Fs.A.T, transpose=True), rtol=1e-9)
    rng = np.random.default_rng(8)
    n, k = 100000, 3
    idx = rng.integers(0, Fs.A.shape[0], size=(n, k))
    lam = rng.uniform(0.0, 1.0, size=(n, k))
    D = np.einsum("nk,nkx->nx", lam, Fs.A[idx])
    bound = (lam * Fs.b[idx]).sum(1)
    hS = oracle_core.support_batch(Pg["Phi"], Pg["W_V"], s, alpha, D.T, transpose=True)
    assert (hS <= bound * (1 + 1e-12) + 1e-15).all()
    # epsilon-outer: alpha from the Rakovic loop satisfies the loop condition alpha <= eps / (eps + Ms)
    assert alpha <= 0.1 / (0.1 + info["Ms"]) + 1e-15
