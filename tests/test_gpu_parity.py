"""GPU parity tests: the HIP library (through the C ABI) against the CPU oracle on the same seeded
inputs, and against the reference's stored goldens.  Bars (BASELINE.json north_star): flags, statuses,
active sets, maxima and RNG streams exact; set quantities <= 1e-9 relative; QP primal <= 1e-6."""
import os

import numpy as np
import pytest

import fixtures
import oracle_core
from lbmpc import _abi

pytestmark = pytest.mark.gpu

SET_RTOL = 1e-9   # (alpha, v), support values, h_k, S_hat.b
QP_TOL = 1e-6     # s_0, c, u


@pytest.fixture(scope="module")
def core():
    from lbmpc import core as c
    c.init(0)
    return c


@pytest.fixture(scope="module")
def P():
    return fixtures.params_dict()


@pytest.fixture(scope="module")
def smp(P):
    return _abi.make_sampler(**fixtures.sampler_args(P))


# ---------------------------------------------------------------- subsystem 2
def test_halfspace_max_golden_results_1(core):
    """Results_1 sample sets (5 ... 20000 samples) -> maxima bit-exact, alpha vs stored Alpha_ini"""
    R = fixtures.npz("results_1.npz")
    WA, Wb = R["W_H"][:, :-1], R["W_H"][:, -1]
    for i in range(5):
        S = R[f"Samples_{i}"]
        m_gpu = core.halfspace_max(WA, S)
        m_cpu = oracle_core.halfspace_max(WA, S)
        np.testing.assert_array_equal(m_gpu, m_cpu)
        a, v, mask, st = core.homothety_lp(WA, Wb, m_gpu)
        a2, v2, mask2, st2 = oracle_core.homothety_lp(WA, Wb, m_cpu)
        assert (a, mask, st) == (a2, mask2, st2)
        np.testing.assert_array_equal(v, v2)
        assert abs(a - R["Alpha_ini"][i]) < 1e-7


@pytest.mark.parametrize("nx,nW,N,batch", [(2, 4, 1, 3), (2, 4, 63, 5), (2, 4, 1000, 7), (2, 4, 20000, 33),
                                           (3, 6, 777, 9), (4, 8, 4097, 11), (1, 2, 513, 4), (2, 6, 999, 3),
                                           (4, 5, 300, 2), (3, 6, 200001, 1), (2, 4, 3000000, 1)])
def test_halfspace_max_shapes(core, nx, nW, N, batch):
    """ragged sizes, every nx, box and non-box row counts, split-over-waves and one-wave paths: bit-exact"""
    rng = np.random.default_rng(nx * 1000 + N)
    WA = rng.normal(size=(nW, nx))
    S = rng.normal(size=(batch, nx, N))
    np.testing.assert_array_equal(core.halfspace_max(WA, S), oracle_core.halfspace_max(WA, S))


def test_halfspace_max_property_full_size(core, P):
    """BASELINE config C2 size (N=20000, batch 2048): max over a set == max of maxima over a partition,
    and planting a known extreme sample moves exactly the rows it should."""
    rng = np.random.default_rng(7)
    N, batch = 20000, 1024
    S = rng.uniform(-0.1, 0.1, size=(batch, 2, N))
    S[5, :, 12345] = [0.3, -0.15]
    WA = P["W_A"]
    m = core.halfspace_max(WA, S)
    m2 = np.maximum(core.halfspace_max(WA, S[:, :, :7001]), core.halfspace_max(WA, S[:, :, 7001:]))
    np.testing.assert_array_equal(m, m2)
    np.testing.assert_array_equal(m[5], oracle_core.halfspace_max(WA, S[5]))
    assert np.all(m[5] >= WA @ np.array([0.3, -0.15]) - 1e-18)


def test_halfspace_max_errors(core, P):
    with pytest.raises(core.LbmpcError):
        core.halfspace_max(np.zeros((17, 2)), np.zeros((2, 4)))       # nW beyond limit
    with pytest.raises(core.LbmpcError):
        core.halfspace_max(np.zeros((4, 5)), np.zeros((5, 4)))        # nx beyond limit
    # a W that is not a bounded polytope (four copies of one halfspace): no basic solution exists, reported as status
    a, v, mask, st = core.homothety_lp(np.ones((4, 2)), np.ones(4), np.zeros(4))
    assert st == _abi.ST_LP_INFEASIBLE and a == 1.0
    with pytest.raises(core.LbmpcError):
        core.homothety_lp(np.ones((2, 2)), np.ones(2), np.zeros(2))   # fewer than nx + 1 rows
    # a sample outside the prior W: LP infeasible, reported (the reference returns IPOPT garbage)
    a, v, mask, st = core.homothety_lp(P["W_A"], P["W_b"], P["W_b"] * np.array([1.0, 1.5, 1.0, 1.0]))
    assert st == _abi.ST_LP_INFEASIBLE and a == 1.0


@pytest.mark.parametrize("case", ["W_true", "3d", "4d"])
def test_homothety_lp_general_polytope(core, P, case):
    """prior W that is not an axis-aligned box (InitialSetComputation.m:69-99 takes any W): vertex-enumeration LP on
    the device == the oracle's, bit for bit (alpha, v, binding rows, status); alpha also against HiGHS; the returned v
    lies on the optimal face"""
    from scipy.optimize import linprog
    rng = np.random.default_rng(42)
    if case == "W_true":
        H = P["W_true_H"]
        WA, Wb = H[:, :-1], H[:, -1]
    else:
        nx = 3 if case == "3d" else 4
        nW = nx + 5
        WA = rng.normal(size=(nW, nx))
        WA = np.vstack([WA, -WA.sum(0, keepdims=True)])     # rows that positively span R^nx: bounded
        WA /= np.linalg.norm(WA, axis=1, keepdims=True)
        Wb = rng.uniform(0.5, 1.5, WA.shape[0])
    nW, nx = WA.shape
    B = 37
    # sample clouds inside W (scaled so that 0 < alpha < 1), one far outside (infeasible)
    M = np.zeros((B, nW))
    for b in range(B):
        X = rng.normal(size=(nx, 40)) + 0.5 * rng.normal(size=(nx, 1))     # an off-centre cloud
        X *= rng.uniform(0.1, 0.9) / np.maximum((WA @ X) / Wb[:, None], 1e-9).max()   # gauge of W: inside
        M[b] = (WA @ X).max(1)
    M[-1] = Wb * 1.5
    a, v, mask, st = core.homothety_lp(WA, Wb, M)
    a2, v2, mask2, st2 = oracle_core.homothety_lp(WA, Wb, M)
    np.testing.assert_array_equal(st, st2)
    np.testing.assert_array_equal(a, a2)
    np.testing.assert_array_equal(v, v2)
    np.testing.assert_array_equal(mask, mask2)
    assert st[-1] == _abi.ST_LP_INFEASIBLE and (st[:-1] == 0).all()
    c = np.zeros(nx + 1); c[0] = -1
    A = np.vstack([np.hstack([-Wb[:, None], WA]), np.hstack([Wb[:, None], -WA])])
    for b in range(0, B - 1, 6):
        r = linprog(c, A_ub=A, b_ub=np.concatenate([np.zeros(nW), Wb - M[b]]), bounds=[(0, 1)] + [(None, None)] * nx, method="highs")
        assert abs((1 - r.x[0]) - a[b]) < 1e-9
        x = np.concatenate([[1 - a[b]], (1 - a[b]) * v[b]])
        assert (A @ x <= np.concatenate([np.zeros(nW), Wb - M[b]]) + 1e-9).all()


def test_sampler_stream_bit_exact(core, P, smp):
    """on-device generation of information sets == oracle stream (Philox + rejection), any set offset"""
    for N, batch, set0 in [(1, 2, 0), (100, 5, 0), (1000, 3, 12345678901), (4099, 2, 7)]:
        S, m, st = core.sample_information_sets(smp, P["W_A"], N, batch, seed=2, set0=set0)
        S2, m2, st2 = oracle_core.sample_information_sets(smp, P["W_A"], N, batch, seed=2, set0=set0)
        np.testing.assert_array_equal(S, S2)
        np.testing.assert_array_equal(m, m2)
        np.testing.assert_array_equal(st, st2)
        # samples lie in W_true = Xi_EV + (-Xi_LV) + (-B U_LV)
        H = P["W_true_H"]
        assert (np.einsum("rk,bkn->brn", H[:, :-1], S) <= H[:, -1][None, :, None] + 1e-9).all()


def test_sampler_few_lanes_per_set(core, P, smp):
    """large batches of small information sets (the closed loop's I0 pre-pass: one set per trial) take fewer lanes per set
    than samples -- several sets per wave, several samples per lane, ragged last round; samples, maxima and statuses
    == oracle (set_learning.hip lanes_per_set)"""
    for N, batch, set0 in [(50, 20000, 3), (100, 70000, 0), (10, 140000, 5), (37, 9000, 11)]:
        S, m, st = core.sample_information_sets(smp, P["W_A"], N, batch, seed=9, set0=set0)
        n = 1500   # the oracle on the first and the last sets
        for lo in (0, batch - n):
            S2, m2, st2 = oracle_core.sample_information_sets(smp, P["W_A"], N, n, seed=9, set0=set0 + lo)
            np.testing.assert_array_equal(S[lo:lo + n], S2)
            np.testing.assert_array_equal(m[lo:lo + n], m2)
            np.testing.assert_array_equal(st[lo:lo + n], st2)


def test_case_2_statistics(core, P, smp):
    """Case_2: vol(W_hat_0) = alpha^nx vol(W) vs card(I_0^w); means agree with Results_2 statistically
    (reference RNG unseeded) and the GPU equals the oracle exactly."""
    R = fixtures.npz("results_2.npz")
    volW = 1.0 * 0.4
    for idx in (0, 8, 16, 24):
        N = int(R["N_sam_ini"][idx])
        _, m, _ = core.sample_information_sets(smp, P["W_A"], N, 256, seed=2, set0=1000 * idx, want_samples=False)
        a, v, mask, st = core.homothety_lp(P["W_A"], P["W_b"], m)
        _, m2, _ = oracle_core.sample_information_sets(smp, P["W_A"], N, 256, seed=2, set0=1000 * idx, want_samples=False)
        a2, v2, _, _ = oracle_core.homothety_lp(P["W_A"], P["W_b"], m2)
        np.testing.assert_array_equal(a, a2)
        np.testing.assert_array_equal(v, v2)
        ref = R["W_Hat_Vol"][idx]
        mine = a ** 2 * volW
        assert abs(mine.mean() - ref.mean()) < 4 * ref.std() / np.sqrt(30) + 0.1 * ref.mean()


# ---------------------------------------------------------------- subsystem 1
def test_support_hs_golden(core, P):
    """hs = max F_Com S.V' (stored in parameters_2x2.mat)"""
    Fc = P["F"] + P["G"] @ P["K"]
    h = core.support_batch(np.eye(2), P["S_V"], 1, 0.0, Fc.T)
    np.testing.assert_allclose(h, P["hs"].ravel(), atol=1.5e-8)
    np.testing.assert_array_equal(h, oracle_core.support_batch(np.eye(2), P["S_V"], 1, 0.0, Fc.T))


@pytest.mark.parametrize("nx,nV,s,nd,tr", [(2, 4, 7, 1000, True), (2, 4, 11, 77, False), (3, 8, 9, 5000, True),
                                           (4, 16, 10, 100000, True), (4, 16, 30, 4097, True), (1, 2, 5, 300, True),
                                           (2, 119, 1, 64, True), (4, 16, 8, 3000000, True), (4, 16, 4, 1000001, True)])
def test_support_batch_vs_oracle(core, nx, nV, s, nd, tr):
    rng = np.random.default_rng(nx + nV + s)
    A = 0.6 * rng.normal(size=(nx, nx)) / np.sqrt(nx)
    Vw = rng.uniform(-1, 1, size=(nV, nx))
    D = rng.normal(size=(nx, nd))
    D /= np.linalg.norm(D, axis=0)
    h = core.support_batch(A, Vw, s, 0.05, D, transpose=tr)
    h2 = oracle_core.support_batch(A, Vw, s, 0.05, D, transpose=tr)
    np.testing.assert_allclose(h, h2, rtol=SET_RTOL, atol=0)
    # positive homogeneity and sub-additivity of a support function
    h3 = core.support_batch(A, Vw, s, 0.05, 2.5 * D[:, :1000], transpose=tr)
    np.testing.assert_allclose(h3, 2.5 * h[:1000], rtol=1e-12)


def test_mrpi_s_alpha(core, P):
    """(s, alpha) loop of MRPISet.m:27-63; the exact-formula cross-checks of SURVEY.md Appendix B"""
    Phi, WV, WA, Wb = P["Phi"], P["W_V"], P["W_A"], P["W_b"]
    D = np.hstack([np.eye(2), -np.eye(2)])
    for eps, s_ref, a_ref in ((0.1, 7, 0.123623), (0.01, 11, 0.0151979)):
        s, a, Ms, rc = core.mrpi_s_alpha(Phi, WV, WA, Wb, D, eps, mode=_abi.MS_MAX_LITERAL)
        assert rc == 0 and s == s_ref and abs(a - a_ref) < 1e-6
        assert (s, a, Ms, rc) == oracle_core.mrpi_s_alpha(Phi, WV, WA, Wb, D, eps, mode=_abi.MS_MAX_LITERAL)
        assert core.mrpi_s_alpha(Phi, WV, WA, Wb, D, eps, mode=_abi.MS_RAKOVIC) == oracle_core.mrpi_s_alpha(Phi, WV, WA, Wb, D, eps, mode=_abi.MS_RAKOVIC)
        Fc = P["F"] + P["G"] @ P["K"]
        h = core.support_batch(Phi, WV, s, a, Fc.T, transpose=True)
        ref = {0.1: [0.162328, 1.020390], 0.01: [0.149607, 0.933744]}[eps]
        np.testing.assert_allclose(h[[0, 2]], ref, atol=2e-6)
    # golden pin: the upstream parameters.mat hs (2-D tree loop at eps = 0.01) through the HIP path
    U = fixtures.params_dict("params_upstream.npz")
    s, a, Ms, rc = core.mrpi_s_alpha(U["Phi"], U["W_V"], U["W_A"], U["W_b"], D, 0.01, mode=_abi.MS_SUM_LITERAL)
    assert (s, rc) == (13, 0)
    h = core.support_batch(U["Phi"], U["W_V"], s, a, (U["F"] + U["G"] @ U["K"]).T, transpose=True)
    np.testing.assert_allclose(h, U["hs"].ravel(), atol=1.5e-8)
    rng = np.random.default_rng(3)
    extra = rng.normal(size=(2, 10))
    D2 = np.hstack([D, extra / np.linalg.norm(extra, axis=0)])
    for mode in (_abi.MS_MAX_LITERAL, _abi.MS_SUM_LITERAL):
        assert core.mrpi_s_alpha(Phi, WV, WA, Wb, D2, 0.1, mode=mode) == oracle_core.mrpi_s_alpha(Phi, WV, WA, Wb, D2, 0.1, mode=mode)


# ---------------------------------------------------------------- subsystem 3
def compare_steps(g, o):
    np.testing.assert_array_equal(g["status"], o["status"])
    np.testing.assert_array_equal(g["n_active"], o["n_active"])
    np.testing.assert_array_equal(g["active_set"], o["active_set"])
    np.testing.assert_array_equal(g["qp_iters"], o["qp_iters"])
    for k in ("alpha", "v", "hk", "Sb_hat", "Wb_hat"):
        np.testing.assert_allclose(g[k], o[k], rtol=SET_RTOL, atol=1e-15, err_msg=k)
    ok = o["status"] == 0
    for k in ("s0", "c", "u"):
        np.testing.assert_allclose(g[k][ok], o[k][ok], rtol=QP_TOL, atol=QP_TOL, err_msg=k)


def test_step_vs_oracle_random(core, P):
    ctx, octx = core.Context(P), oracle_core.Context(P)
    rng = np.random.default_rng(0)
    V = P["F_N_RMPC_V"]
    B = 500
    lam = rng.dirichlet(np.ones(3), size=B)
    X = np.einsum("bi,bik->bk", lam, V[rng.integers(0, len(V), (B, 3))]) * rng.uniform(0.1, 1.05, (B, 1))
    X[-3:] = [[40, 10], [-30, 5], [0, 0]]  # infeasible / infeasible / origin
    alpha = rng.uniform(0.05, 1.0, B)
    v = rng.uniform(-0.05, 0.05, (B, 2))
    w = rng.uniform(-0.2, 0.2, (B, 2)) * np.array([1.0, 0.4])
    for uq in (False, True):
        g = ctx.step(X.T, alpha, v.T, w.T, uq_update=uq)
        o = octx.step(X.T, alpha, v.T, w.T, uq_update=uq)
        compare_steps(g, o)
        assert (o["n_active"] > 0).sum() > 100 and (o["status"] == _abi.ST_QP_INFEASIBLE).sum() >= 2
    ctx.close(); octx.close()


@pytest.mark.parametrize("tag", ["small", "large"])
def test_step_golden_results_3(core, P, tag):
    """the 50 stored steps of Case_3 through the HIP path: h_k, S_hat.b, (s_0, u_k), alpha_k"""
    from test_oracle_golden import recover_alpha_v
    from scipy.optimize import linprog
    R = fixtures.npz(f"results_3_{tag}.npz")
    WA, Wb = P["W_A"], P["W_b"]
    T = np.linalg.inv(np.eye(2) - P["Phi"])
    av = [recover_alpha_v(WA, Wb, R["W_Hat_H"][k][:, -1]) for k in range(50)]
    a0, v0 = av[0]
    SA = R["S_Hat_H"][0][:, :-1]
    Sb = (R["S_Hat_H"][0][:, -1] - (1 - a0) * SA @ T @ v0) / a0
    Fc = P["F"] + P["G"] @ P["K"]
    hs = np.array([-linprog(-Fc[i], A_ub=SA, b_ub=Sb, bounds=[(None, None)] * 2, method="highs").fun for i in range(4)])
    d = dict(P)
    d.update(S_A=SA, S_b=Sb, hs=hs)
    ctx, octx = core.Context(d), oracle_core.Context(d)
    X, U, S0, HS = R["State_RM"], R["Control_EV"], R["Nominal_RM_State"], R["Hs"]
    alphas = np.array([a for a, _ in av])
    vs = np.array([v for _, v in av])
    g = ctx.step(X[:, :50], alphas, vs.T, None, uq_update=False)
    o = octx.step(X[:, :50], alphas, vs.T, None, uq_update=False)
    compare_steps(g, o)
    np.testing.assert_allclose(g["hk"], HS.T, atol=1.5e-8)
    np.testing.assert_allclose(g["Sb_hat"], R["S_Hat_H"][:, :, -1], atol=2e-9)
    np.testing.assert_allclose(g["s0"], S0[:, :50].T, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(g["u"][:, 0], U[0], rtol=1e-6, atol=1e-6)
    Wn = X[:, 1:50] - P["A"] @ X[:, :49] - P["B"] @ U[:, :49]
    g2 = ctx.step(X[:, 1:50], alphas[:49], vs[:49].T, Wn, uq_update=True)
    np.testing.assert_allclose(g2["alpha"], alphas[1:], atol=1e-7)
    ctx.close(); octx.close()


def test_nominal_rmpc(core, P):
    """NominalRobustMPC.solve == the same QP with alpha = 1, v = 0, no update (parity unpinned vs reference:
    no stored output; oracle only)"""
    ctx, octx = core.Context(P), oracle_core.Context(P)
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, (2, 64)) * np.array([[12.0], [5.0]])
    g = ctx.step(X, np.ones(64), np.zeros((2, 64)), None, uq_update=False)
    o = octx.step(X, np.ones(64), np.zeros((2, 64)), None, uq_update=False)
    compare_steps(g, o)
    np.testing.assert_allclose(g["hk"], np.tile(P["hs"].ravel(), (64, 1)), rtol=1e-15)
    ctx.close(); octx.close()


def compare_mc(g, o):
    for k in ("infeasible", "status", "qp_iters"):
        np.testing.assert_array_equal(g[k], o[k], err_msg=k)
    for k in ("alpha0", "alpha_final", "v_final", "traj_alpha"):
        np.testing.assert_allclose(g[k], o[k], rtol=SET_RTOL, atol=1e-15, err_msg=k)
    for k in ("x_final", "traj_x", "traj_u", "traj_s0"):
        np.testing.assert_allclose(g[k], o[k], rtol=QP_TOL, atol=QP_TOL, err_msg=k)


def test_mc_case_3_trajectory(core, P, smp):
    """Case_3: K_N = 50 from x_RM_0 = [-12; 5], card(I0) = 100; whole trajectories vs oracle"""
    ctx, octx = core.Context(P), oracle_core.Context(P)
    kw = dict(trials=70, steps=50, card_I0=100, seed=1, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0, traj_trials=70)
    g, o = ctx.mc_run(smp, **kw), octx.mc_run(smp, **kw)
    compare_mc(g, o)
    assert o["qp_iters"].min() > 10 and o["infeasible"].sum() == 0
    ctx.close(); octx.close()


def test_mc_case_5_feasibility_flags(core, P, smp):
    """Case_5: near-boundary initial state, 20 steps; Infeasible_Index exact vs oracle, rate vs Results_5"""
    R = fixtures.npz("results_5.npz")
    ctx, octx = core.Context(P), oracle_core.Context(P)
    for n, trials in ((10, 2000), (100, 300)):
        kw = dict(trials=trials, steps=20, card_I0=n, seed=5, x0=R[f"x_RM_0_{n}"], acc_bound=2.0, distance_error=15.0,
                  traj_trials=5)
        g, o = ctx.mc_run(smp, **kw), octx.mc_run(smp, **kw)
        compare_mc(g, o)
        rate, ref = g["infeasible"].mean(), R[f"Infeasible_Index_{n}"].mean()
        if n == 10:
            assert 0.02 < rate < 0.40 and ref == 0.16   # statistical: reference RNG is unseeded
        else:
            assert rate <= 0.02 and ref == 0.0
    ctx.close(); octx.close()


def test_mc_sharding_invariance(core, P, smp):
    """trial-sharded runs (trial0 offsets, as the multi-GPU driver issues them) reproduce the unsharded run bitwise"""
    ctx = core.Context(P)
    kw = dict(steps=12, card_I0=10, seed=9, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0)
    full = ctx.mc_run(smp, trials=300, **kw)
    a = ctx.mc_run(smp, trials=130, trial0=0, **kw)
    b = ctx.mc_run(smp, trials=170, trial0=130, **kw)
    for k in ("infeasible", "status", "qp_iters", "alpha_final", "v_final", "x_final"):
        np.testing.assert_array_equal(full[k], np.concatenate([a[k], b[k]]), err_msg=k)
    ctx.close()


def test_mc_wide_path_matches_narrow(core, P, smp):
    """2 trials per lane (chosen automatically from 524288 trials; forced here) == 1 trial per lane, bitwise.  1003
    trials: the last wave holds three trials in its first slot and none in its second"""
    import os
    ctx = core.Context(P)
    kw = dict(trials=1003, steps=8, card_I0=10, seed=4, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0)
    try:
        os.environ["LBMPC_MC_T"] = "2"
        wide = ctx.mc_run(smp, **kw)
        os.environ["LBMPC_MC_T"] = "1"
        narrow = ctx.mc_run(smp, **kw)
    finally:
        os.environ.pop("LBMPC_MC_T", None)
    for k in ("infeasible", "status", "qp_iters", "alpha_final", "v_final", "x_final"):
        np.testing.assert_array_equal(wide[k], narrow[k], err_msg=k)
    ctx.close()


@pytest.mark.parametrize("system,x0,steps", [("3x3", [-12.0, 5.0, 0.0], 30), ("4x4", [-12.0, 5.0, 0.0, 0.0], 40)])
def test_mc_generated_systems_vs_oracle(core, system, x0, steps):
    """C3 / C4 systems (build-generated parameters): closed loop through the transient (filter scans + QP solves)
    and into the steady state (ellipsoid early-out), whole trajectories vs oracle"""
    from lbmpc import params
    Pg = params.load(system)
    smpg = params.sampler(Pg)
    ctx, octx = core.Context(Pg), oracle_core.Context(Pg)
    kw = dict(trials=200, steps=steps, card_I0=10, seed=4, x0=x0, acc_bound=float(Pg["acc_bound"][0]),
              distance_error=float(Pg["distance_error"][0]), traj_trials=200)
    g, o = ctx.mc_run(smpg, **kw), octx.mc_run(smpg, **kw)
    compare_mc(g, o)
    assert o["qp_iters"].min() > 0
    cnt = ctx.mc_counters()
    assert cnt["wave_steps"] % steps == 0 and cnt["wave_steps"] >= 4 * steps   # 200 trials in waves of 4 .. 64
    assert cnt["qp_solves"] > 0 and cnt["pivot_scans"] >= cnt["solver_rounds"] >= 0
    # every QP went through one of the two solvers (the one-QP-per-lane solver of the S-row regime first)
    assert cnt["lane_scans"] + cnt["pivot_scans"] > 0 and cnt["row_solver_qps"] <= cnt["qp_solves"]
    assert (cnt["row_solver_qps"] > 0) == (cnt["solver_rounds"] > 0)
    assert 0 < cnt["filter_wave_steps"] < cnt["wave_steps"]   # transient scanned, steady state skipped
    ctx.close(); octx.close()


def test_mc_trial_chunks_match_single_launch(core, P, smp):
    """a small disturbance-scratch cap (LBMPC_MC_XI_GIB) splits the run into trial chunks; results are bitwise equal"""
    import os
    ctx = core.Context(P)
    kw = dict(trials=1500, steps=10, card_I0=10, seed=7, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0, traj_trials=1500)
    one = ctx.mc_run(smp, **kw)
    try:
        os.environ["LBMPC_MC_XI_GIB"] = "0.0001"   # ~105 kB -> 512-trial chunks
        many = ctx.mc_run(smp, **kw)
        os.environ["LBMPC_MC_SPLIT"] = "4"         # chunks x step segments: every chunk reuses the hand-over buffers
        both = ctx.mc_run(smp, **kw)
    finally:
        os.environ.pop("LBMPC_MC_XI_GIB", None)
        os.environ.pop("LBMPC_MC_SPLIT", None)
    for k in one:
        np.testing.assert_array_equal(one[k], many[k], err_msg=k)
        np.testing.assert_array_equal(one[k], both[k], err_msg=k + " (chunks x segments)")
    ctx.close()


@pytest.mark.parametrize("system", ["2x2", "3x3", "4x4"])
def test_step_first_pass_filter_is_conservative(core, P, system):
    """states scattered around the boundary of the tube set alpha*S (gauge 0.5 .. 1 +- 1e-12 .. 1.3): the
    inscribed-ellipsoid early-out and the packed-fp32 filter in front of the exact solver must never hide a violated
    row, so statuses, active sets, iteration counts and solutions equal the oracle's (which scans every row in fp64)"""
    from lbmpc import params
    Pg = P if system == "2x2" else params.load(system)
    nx = int(np.ravel(Pg["nx"])[0])
    SA, Sb = np.asarray(Pg["S_A"], dtype=np.float64), np.asarray(Pg["S_b"], dtype=np.float64).ravel()
    ctx, octx = core.Context(Pg), oracle_core.Context(Pg)
    rng = np.random.default_rng(11)
    B = 1536
    d = rng.normal(size=(B, nx)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    alpha = rng.uniform(0.05, 1.0, B)
    gauge = (d @ SA.T / Sb).max(axis=1)                 # boundary of S along d is at d / gauge
    scale = np.concatenate([rng.uniform(0.5, 0.999, B // 4), 1.0 + rng.choice([-1, 1], B // 4) * 10.0 ** rng.uniform(-13, -6, B // 4),
                            1.0 + rng.choice([-1, 1], B // 4) * 10.0 ** rng.uniform(-6, -2, B // 4), rng.uniform(1.001, 1.3, B // 4)])
    # v = 0: x - s0 must lie in alpha*S; x = -q on the scaled boundary
    X = (d / gauge[:, None]) * (alpha * scale)[:, None]
    v = np.zeros((B, nx)); w = np.zeros((B, nx))
    g = ctx.step(X.T, alpha, v.T, w.T, uq_update=False)
    o = octx.step(X.T, alpha, v.T, w.T, uq_update=False)
    compare_steps(g, o)
    assert (o["n_active"] == 0).sum() > B // 8 and (o["n_active"] > 0).sum() > B // 8
    ctx.close(); octx.close()


@pytest.mark.parametrize("system,x0", [("3x3", [-12.0, 5.0, 0.0]), ("4x4", [-12.0, 5.0, 0.0, 0.0])])
def test_mc_oriented_proposal_vs_oracle(core, system, x0):
    """lbmpc_sampler.EV_U / LV_U (principal-axes proposal boxes): closed loop vs oracle on the same streams, and the
    accepted disturbances keep the distribution of the axis-aligned proposal (uniform on Xi): first two moments of
    the learned w samples agree within Monte-Carlo error"""
    from lbmpc import params
    Pg = params.load(system)
    nx = len(x0)
    ctx, octx = core.Context(Pg), oracle_core.Context(Pg)
    kw = dict(trials=200, steps=25, card_I0=10, seed=4, x0=x0, acc_bound=float(Pg["acc_bound"][0]),
              distance_error=float(Pg["distance_error"][0]), traj_trials=200)
    so = params.sampler(Pg, oriented=True)
    assert so.struct.EV_U and so.struct.LV_U     # both boxes are oriented for these systems
    compare_mc(ctx.mc_run(so, **kw), octx.mc_run(so, **kw))
    WA = np.asarray(Pg["W_H"])[:, :-1]
    N = 40000
    sa, _, _ = core.sample_information_sets(params.sampler(Pg), WA, N, 1, seed=3)
    sb, _, _ = core.sample_information_sets(so, WA, N, 1, seed=3)
    sa, sb = np.asarray(sa)[0], np.asarray(sb)[0]     # nx x N
    sd = sa.std(axis=1)
    assert np.all(np.abs(sa.mean(1) - sb.mean(1)) < 5 * sd / np.sqrt(N) * 1.5)
    assert np.allclose(np.cov(sa), np.cov(sb), rtol=0.06, atol=0.02 * sd.max() ** 2)
    ctx.close(); octx.close()


@pytest.mark.parametrize("trials,steps", [(1, 0), (1, 1), (3, 2), (65, 3), (130, 5)])
def test_mc_tiny_shapes_vs_oracle(core, P, smp, trials, steps):
    """edge shapes of the closed-loop launch: no steps at all, fewer steps than a store group of the pre-pass,
    fewer trials than a 4-trial sampling group / a wave, ragged last wave"""
    ctx, octx = core.Context(P), oracle_core.Context(P)
    kw = dict(trials=trials, steps=steps, card_I0=7, seed=21, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0,
              traj_trials=trials)
    compare_mc(ctx.mc_run(smp, **kw), octx.mc_run(smp, **kw))
    ctx.close(); octx.close()


def test_mc_trials_per_wave_invariance(core, P, smp):
    """small batches run with fewer trials per wave (4 .. 32 lanes) so that the solver rounds spread over more waves;
    every packing gives bitwise the same results (LBMPC_MC_TPW forces one)"""
    import os
    ctx = core.Context(P)
    kw = dict(trials=700, steps=9, card_I0=10, seed=13, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0, traj_trials=700)
    ref = ctx.mc_run(smp, **kw)          # automatic choice (4 per wave at this size)
    try:
        for tpw, waves in (("8", "8"), ("32", "8"), ("64", "8"), ("10", "4"), ("7", "6"), ("1", "8")):   # any packing, powers of two or not
            os.environ["LBMPC_MC_TPW"], os.environ["LBMPC_MC_WAVES"] = tpw, waves
            got = ctx.mc_run(smp, **kw)
            for k in ref:
                np.testing.assert_array_equal(ref[k], got[k], err_msg=f"{k} tpw={tpw} waves={waves}")
    finally:
        os.environ.pop("LBMPC_MC_TPW", None)
        os.environ.pop("LBMPC_MC_WAVES", None)
    ctx.close()


@pytest.mark.parametrize("lean", ["0", "1"])
@pytest.mark.parametrize("split", ["1", "5", "11"])
def test_mc_step_segments_match_single_launch(core, P, smp, split, lean):
    """long horizons run in step segments with the trial state handed over through HBM: mc_kernel for the transient,
    mc_steady_kernel for the rest (lean=1; trials that still need a QP are handed back to a resume launch of
    mc_kernel, which here -- split inside the transient -- is most of them) or mc_kernel again (lean=0).
    LBMPC_MC_SPLIT forces the split point; results are bitwise those of the single launch, trajectories included"""
    import os
    ctx = core.Context(P)
    kw = dict(trials=333, steps=12, card_I0=10, seed=17, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0, traj_trials=333)
    try:
        os.environ["LBMPC_MC_SPLIT"] = "0"      # single launch
        one = ctx.mc_run(smp, **kw)
        os.environ["LBMPC_MC_SPLIT"] = split
        os.environ["LBMPC_MC_LEAN"] = lean
        two = ctx.mc_run(smp, **kw)
    finally:
        os.environ.pop("LBMPC_MC_SPLIT", None)
        os.environ.pop("LBMPC_MC_LEAN", None)
    for k in one:
        np.testing.assert_array_equal(one[k], two[k], err_msg=k)
    ctx.close()


def test_mc_long_horizon_default_split_vs_oracle(core):
    """the default policy (steps >= 64: 32 transient steps, then the steady-state kernel) against the oracle on 4x4"""
    from lbmpc import params
    Pg = params.load("4x4")
    smpg = params.sampler(Pg, oriented=True)
    ctx, octx = core.Context(Pg), oracle_core.Context(Pg)
    kw = dict(trials=192, steps=80, card_I0=10, seed=4, x0=[-12.0, 5.0, 0.0, 0.0], acc_bound=float(Pg["acc_bound"][0]),
              distance_error=float(Pg["distance_error"][0]), traj_trials=192)
    compare_mc(ctx.mc_run(smpg, **kw), octx.mc_run(smpg, **kw))
    ctx.close(); octx.close()


def test_mc_c4_full_size_slices_vs_oracle(core):
    """C4 at BASELINE.json's full size on one GPU (131072 trials x 1000 steps, card(I0) = 10, seed 4, the bench's
    proposal boxes): the first and the last 96 trials of the full launch against the oracle run on those trial ids
    (flags, iteration counts bit-exact; final set and state within the tolerances above), and the two half-size shards
    the 2-GPU driver would issue reproduce the full launch bitwise"""
    from lbmpc import params
    Pg = params.load("4x4")
    smpg = params.sampler(Pg, oriented=True)
    ctx, octx = core.Context(Pg), oracle_core.Context(Pg)
    n, w = 131072, 96
    kw = dict(steps=1000, card_I0=10, seed=4, x0=[-12.0, 5.0, 0.0, 0.0], acc_bound=float(Pg["acc_bound"][0]),
              distance_error=float(Pg["distance_error"][0]))
    full = ctx.mc_run(smpg, trials=n, **kw)
    for t0 in (0, n - w):
        o = octx.mc_run(smpg, trials=w, trial0=t0, **kw)
        sl = slice(t0, t0 + w)
        for k in ("infeasible", "status", "qp_iters"):
            np.testing.assert_array_equal(full[k][sl], o[k], err_msg=f"{k} trial0={t0}")
        for k in ("alpha0", "alpha_final", "v_final"):
            np.testing.assert_allclose(full[k][sl], o[k], rtol=SET_RTOL, atol=1e-15, err_msg=f"{k} trial0={t0}")
        np.testing.assert_allclose(full["x_final"][sl], o["x_final"], rtol=QP_TOL, atol=QP_TOL, err_msg=f"x_final trial0={t0}")
    a = ctx.mc_run(smpg, trials=n // 2, trial0=0, **kw)
    b = ctx.mc_run(smpg, trials=n // 2, trial0=n // 2, **kw)
    for k in ("infeasible", "status", "qp_iters", "alpha0", "alpha_final", "v_final", "x_final"):
        np.testing.assert_array_equal(full[k], np.concatenate([a[k], b[k]]), err_msg=k)
    assert full["infeasible"].sum() == 0 and full["qp_iters"].min() > 0
    ctx.close(); octx.close()


def test_mc_two_trials_per_lane_default_vs_oracle(core, P, smp):
    """from 524288 trials a lane carries two trials (chosen by the library, nothing forced here): the first and the
    last 64 trials of a ragged batch against the oracle on those trial ids"""
    ctx, octx = core.Context(P), oracle_core.Context(P)
    n, w = 524288 + 37, 64
    kw = dict(steps=12, card_I0=10, seed=31, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0)
    full = ctx.mc_run(smp, trials=n, **kw)
    for t0 in (0, n - w):
        o = octx.mc_run(smp, trials=w, trial0=t0, **kw)
        sl = slice(t0, t0 + w)
        for k in ("infeasible", "status", "qp_iters"):
            np.testing.assert_array_equal(full[k][sl], o[k], err_msg=f"{k} trial0={t0}")
        for k in ("alpha0", "alpha_final", "v_final"):
            np.testing.assert_allclose(full[k][sl], o[k], rtol=SET_RTOL, atol=1e-15, err_msg=f"{k} trial0={t0}")
        np.testing.assert_allclose(full["x_final"][sl], o["x_final"], rtol=QP_TOL, atol=QP_TOL, err_msg=f"x_final trial0={t0}")
    assert full["qp_iters"].min() > 0
    ctx.close(); octx.close()


def test_kernel_3_usage_errors(core, P, smp):
    """usage errors of the controller entry points come back as negative codes with a message (no exception crosses
    the ABI, nothing is launched); the reference has no checks here at all"""
    from lbmpc import params
    ctx = core.Context(P)
    kw = dict(trials=8, steps=3, card_I0=10, seed=1, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0)
    with pytest.raises(core.LbmpcError, match="sampler dims"):
        P3 = params.load("3x3")
        ctx.mc_run(params.sampler(P3), **kw)                     # 3x3 disturbance sets on a 2x2 controller
    with pytest.raises(core.LbmpcError, match="bad configuration"):
        ctx.mc_run(smp, **dict(kw, card_I0=0))                   # empty initial information set
    with pytest.raises(core.LbmpcError, match="bad configuration"):
        ctx.mc_run(smp, **dict(kw, sampler_max_attempts=0))
    out = ctx.mc_run(smp, **dict(kw, trials=0))                  # nothing to do is not an error
    assert out["status"].shape == (0,)
    big = dict(P)
    big["S_A"] = np.tile(P["S_A"], (80, 1)); big["S_b"] = np.tile(np.ravel(P["S_b"]), 80)   # 6400 rows: beyond the LDS
    with pytest.raises(core.LbmpcError):
        c2 = core.Context(big)
        c2.mc_run(smp, **kw)
    ctx.close()


def test_device_pointers_without_16_byte_alignment(core, P):
    """device-pointer entry points with buffers that start 8 bytes off a 16-byte boundary (a slice of a tensor):
    the kernels fall back from double2 loads to scalar ones, results are those of the aligned call, bitwise"""
    import torch
    from lbmpc import dev
    rng = np.random.default_rng(12)
    dvc = torch.device("cuda:0")
    # support sweep, nx = 4
    A = 0.6 * rng.normal(size=(4, 4)) / 2
    Vw = rng.uniform(-1, 1, size=(16, 4))
    D = rng.normal(size=(5001, 4))
    buf = torch.zeros(5001 * 4 + 1, dtype=torch.float64, device=dvc)
    buf[1:] = torch.from_numpy(D.ravel()).to(dvc)
    off = buf[1:].view(5001, 4)
    assert off.data_ptr() % 16 == 8
    h_off = dev.support_batch(A, Vw, 6, 0.05, off)
    h_al = dev.support_batch(A, Vw, 6, 0.05, torch.from_numpy(D).to(dvc))
    assert torch.equal(h_off, h_al)
    np.testing.assert_allclose(h_al.cpu().numpy(), oracle_core.support_batch(A, Vw, 6, 0.05, D.T), rtol=SET_RTOL, atol=0)
    # halfspace max, nx = 2
    S = rng.uniform(-0.2, 0.2, size=(3, 4097, 2))
    buf2 = torch.zeros(S.size + 1, dtype=torch.float64, device=dvc)
    buf2[1:] = torch.from_numpy(S.ravel()).to(dvc)
    off2 = buf2[1:].view(3, 4097, 2)
    assert off2.data_ptr() % 16 == 8
    m_off = dev.halfspace_max(P["W_A"], off2)
    m_al = dev.halfspace_max(P["W_A"], torch.from_numpy(S).to(dvc))
    assert torch.equal(m_off, m_al)
    np.testing.assert_array_equal(m_al.cpu().numpy()[1], oracle_core.halfspace_max(P["W_A"], S[1].T))
    torch.cuda.synchronize()


@pytest.mark.parametrize("lean", ["0", "1"])
def test_mc_step_segments_with_two_trials_per_lane(core, P, smp, lean):
    """the combination large long-horizon batches run with: two trials per lane AND step segments (forced here on a
    ragged batch), against the plain single launch with one trial per lane"""
    import os
    ctx = core.Context(P)
    kw = dict(trials=1003, steps=12, card_I0=10, seed=19, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0, traj_trials=1003)
    try:
        os.environ["LBMPC_MC_SPLIT"] = "0"
        os.environ["LBMPC_MC_T"] = "1"
        one = ctx.mc_run(smp, **kw)
        os.environ["LBMPC_MC_SPLIT"] = "4"
        os.environ["LBMPC_MC_T"] = "2"
        os.environ["LBMPC_MC_LEAN"] = lean
        two = ctx.mc_run(smp, **kw)
    finally:
        for k in ("LBMPC_MC_SPLIT", "LBMPC_MC_T", "LBMPC_MC_LEAN"):
            os.environ.pop(k, None)
    for k in one:
        np.testing.assert_array_equal(one[k], two[k], err_msg=k)
    ctx.close()


@pytest.mark.parametrize("system,oriented", [("2x2", False), ("2x2", True), ("3x3", True), ("4x4", False), ("4x4", True)])
def test_sampler_row_pruning_keeps_the_stream(core, system, oriented):
    """pack_sampler drops polytope rows that no proposal of the box can violate (all rows of the 4x4 cuboids in their
    own axes, the cap rows of the 3x3 prisms, the rows on the box faces of the 2x2 trapezoids); the oracle tests every
    row of every candidate.  The accepted samples must be the same bits."""
    from lbmpc import params
    Pg = params.load(os.path.join(os.path.dirname(__file__), "golden", "params_2x2.npz")) if system == "2x2" else params.load(system)
    smp = params.sampler(Pg, oriented=oriented)
    for N, batch, set0 in [(3000, 4, 0), (257, 3, 99)]:
        S, m, st = core.sample_information_sets(smp, Pg["W_A"], N, batch, seed=11, set0=set0)
        S2, m2, st2 = oracle_core.sample_information_sets(smp, Pg["W_A"], N, batch, seed=11, set0=set0)
        np.testing.assert_array_equal(S, S2)
        np.testing.assert_array_equal(m, m2)
        np.testing.assert_array_equal(st, st2)


@pytest.mark.parametrize("hs_scale", [1.0, 6.0, 14.0])
def test_steady_kernel_tightening_bound_and_row_walk(core, P, smp, hs_scale):
    """mc_steady_kernel walks the tightening rows only when its bound cannot rule out a negative 1 - h_k.  With hs
    scaled up the bound fails (6x: the walk runs and finds every row positive or not, depending on alpha; 14x: 1 - h_k
    is negative and every trial is handed to the solver kernel): the steady-state path must stay bitwise equal to
    running the same steps in mc_kernel (LBMPC_MC_LEAN=0), which evaluates the rows at every step."""
    import os
    Ps = dict(P)
    Ps["hs"] = np.asarray(P["hs"]) * hs_scale
    ctx = core.Context(Ps)
    kw = dict(trials=200, steps=70, card_I0=10, seed=23, x0=[-1.0, 0.5], acc_bound=2.0, distance_error=15.0, traj_trials=8)
    try:
        os.environ["LBMPC_MC_SPLIT"] = "5"
        os.environ["LBMPC_MC_LEAN"] = "0"
        ref = ctx.mc_run(smp, **kw)
        os.environ["LBMPC_MC_LEAN"] = "1"
        got = ctx.mc_run(smp, **kw)
    finally:
        os.environ.pop("LBMPC_MC_SPLIT", None)
        os.environ.pop("LBMPC_MC_LEAN", None)
    for k in ref:
        np.testing.assert_array_equal(ref[k], got[k], err_msg=k)
    ctx.close()
