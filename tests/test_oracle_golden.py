"""Pin the CPU oracle (oracle/lbmpc_oracle.c) against every stored output of the reference that
covers the hot path: Results_1 (set learning), stats_2x2 (solve_opt), Results_3_*
(per-step KATs of UQRobustMPC.solve), parameters_2x2 (hs).  Tolerances vs the stored reference are
the accuracy of IPOPT itself (alpha 5e-8, h_k 1.2e-8, s_0 3e-8, u 6e-8)."""
import ctypes as C

import numpy as np
import pytest
from scipy.optimize import linprog

import fixtures
import oracle_ffi
from lbmpc import _abi
from lbmpc._abi import dptr, fcol, iptr


@pytest.fixture(scope="module")
def orc():
    return oracle_ffi.load()


@pytest.fixture(scope="module")
def P():
    return fixtures.params_dict()


def lp_rows(WA, Wb, m):
    """rows R [beta; y] <= rhs of InitialSetComputation.m:74-79 after the max-reduction"""
    nW, nx = WA.shape
    R = np.zeros((2 * nW + 2, nx + 1))
    rhs = np.zeros(2 * nW + 2)
    R[:nW, 0], R[:nW, 1:] = -Wb, WA
    R[nW:2 * nW, 0], R[nW:2 * nW, 1:], rhs[nW:2 * nW] = Wb, -WA, Wb - m
    R[2 * nW, 0] = -1
    R[2 * nW + 1, 0], rhs[2 * nW + 1] = 1, 1
    return R, rhs


def highs_alpha(WA, Wb, m):
    R, rhs = lp_rows(WA, Wb, m)
    c = np.zeros(R.shape[1])
    c[0] = -1
    res = linprog(c, A_ub=R, b_ub=rhs, bounds=[(None, None)] * R.shape[1], method="highs")
    assert res.status == 0
    return 1 - res.x[0]


def halfspace_max(orc, WA, samples):
    nW, nx = WA.shape
    s = fcol(samples)
    m = np.zeros(nW)
    assert orc.orc_halfspace_max(dptr(fcol(WA)), nW, nx, dptr(s), s.shape[1], 1, s.size, dptr(m)) == 0
    return m


def box_lp(orc, WA, Wb, m):
    nW, nx = WA.shape
    a, v, mask, st = np.zeros(1), np.zeros(nx), np.zeros(1, np.uint32), np.zeros(1, np.int32)
    rc = orc.orc_homothety_lp(dptr(fcol(WA)), dptr(fcol(Wb)), nW, nx, dptr(fcol(m)), 1, _abi.TIE_CENTER, dptr(a),
                              dptr(v), mask.ctypes.data_as(_abi.c_uint32_p), iptr(st))
    assert rc == 0
    return a[0], v, int(mask[0]), int(st[0])


def enum_lp(orc, WA, Wb, m):
    nW, nx = WA.shape
    a, v, y = np.zeros(1), np.zeros(nx), np.zeros(nx)
    st = orc.orc_homothety_lp_enum(dptr(fcol(WA)), dptr(fcol(Wb)), nW, nx, dptr(fcol(m)), dptr(a), dptr(v), dptr(y))
    return a[0], v, y, st


EXACT_ALPHA_R1 = [0.0795854905, 0.1990103278, 0.2717102036, 0.3347049807, 0.3273450526]


def test_results_1_set_learning(orc):
    """Samples_ini{k} -> (Alpha_ini, V_ini, W_Hat_ini): InitialSetComputation.solve, Case_1...:29-42"""
    R = fixtures.npz("results_1.npz")
    WA, Wb = R["W_H"][:, :-1], R["W_H"][:, -1]
    for i in range(5):
        S = R[f"Samples_{i}"]
        m = halfspace_max(orc, WA, S)
        np.testing.assert_allclose(m, (WA @ S).max(axis=1), rtol=0, atol=1e-15)
        a_enum, v_enum, y_enum, st = enum_lp(orc, WA, Wb, m)
        assert st == 0
        a_box, v_box, mask, st = box_lp(orc, WA, Wb, m)
        assert st == 0 and mask != 0
        a_highs = highs_alpha(WA, Wb, m)
        # the three exact solvers agree; the reference (IPOPT) is the inexact one
        assert abs(a_enum - a_highs) < 1e-12 and abs(a_box - a_enum) < 1e-13
        assert abs(a_box - EXACT_ALPHA_R1[i]) < 1e-9
        assert abs(a_box - R["Alpha_ini"][i]) < 1e-7
        # canonical v and the reference's v are both in the optimal face (all LP rows hold at beta*)
        Rm, rhs = lp_rows(WA, Wb, m)
        for v, tol in ((v_box, 1e-12), (v_enum, 1e-10), (R["V_ini"][:, i], 1e-7)):
            x = np.concatenate([[1 - a_box], (1 - a_box) * v])
            assert (Rm @ x - rhs).max() < tol
        # binding coordinate is pinned: reference and canonical agree there
        for k in range(2):
            if mask >> k & 1:
                assert abs(v_box[k] - R["V_ini"][k, i]) < 2e-7
        # W_hat = (1-alpha) v + alpha W in H-rep from the STORED (alpha, v)
        a_ref, v_ref = R["Alpha_ini"][i], R["V_ini"][:, i]
        H = R[f"W_Hat_{i}_H"]
        np.testing.assert_allclose(H[:, :-1], WA, atol=1e-12)
        np.testing.assert_allclose(H[:, -1], a_ref * Wb + (1 - a_ref) * WA @ v_ref, atol=1e-10)


def test_stats_alpha_opt(orc, P):
    """solve_opt(W_true.V') of offline_parameters_computation_nxn.m:357 -> stats.alpha_opt, v_opt"""
    st = fixtures.npz("stats.npz")
    WA, Wb = P["W_A"], P["W_b"]
    m = halfspace_max(orc, WA, P["W_true_V"].T)
    a, v, mask, status = box_lp(orc, WA, Wb, m)
    assert status == 0
    assert abs(a - 0.36746915025) < 1e-10
    assert abs(a - st["2x2_alpha_opt"][0]) < 1e-7
    assert mask == 0b10 and abs(v[1] - st["2x2_v_opt"][1]) < 1e-8
    # same (alpha_opt, v_opt) is stored as W_hat_opt's H-rep
    H = P["W_hat_opt_H"]
    a_ref, v_ref = st["2x2_alpha_opt"][0], st["2x2_v_opt"]
    np.testing.assert_allclose(H[:, -1], a_ref * Wb + (1 - a_ref) * WA @ v_ref, atol=1e-9)


def test_hs_from_S_vertices(orc, P):
    """hs = h_S((F+GK)_i) = max F_Com S.V'  (hs_debug_experiment.m:103-111; nxn.m:734)"""
    Fc = P["F"] + P["G"] @ P["K"]
    V = fcol(P["S_V"])
    h = np.zeros(4)
    I = fcol(np.eye(2))
    assert orc.orc_support_batch(dptr(I), 2, dptr(V), V.shape[0], 1, 0.0, dptr(fcol(Fc.T)), 4, 0, dptr(h)) == 0
    np.testing.assert_allclose(h, P["hs"].ravel(), atol=1.5e-8)
    np.testing.assert_allclose(h, [0.1489621924, 0.1489621908, 0.929358463, 0.929358463], atol=1e-9)
    # LP over the H-rep (UQRobustMPC.m:109-122) gives the same numbers
    for i in range(4):
        r = linprog(-Fc[i], A_ub=P["S_A"], b_ub=P["S_b"], bounds=[(None, None)] * 2, method="highs")
        assert abs(-r.fun - h[i]) < 1e-8


def recover_alpha_v(WA, Wb, bh):
    """invert b_hat = alpha b + (1-alpha) W_A v for the 2x2 box W (rows: -y, +x, +y, -x)"""
    c = np.abs(WA).max(axis=1)
    ax = np.abs(WA).argmax(axis=1)
    sg = np.sign(WA[np.arange(len(ax)), ax])
    nx = WA.shape[1]
    alpha, v = None, np.zeros(nx)
    for k in range(nx):
        p = [j for j in range(len(ax)) if ax[j] == k and sg[j] > 0][0]
        q = [j for j in range(len(ax)) if ax[j] == k and sg[j] < 0][0]
        a_k = (bh[p] / c[p] + bh[q] / c[q]) / (Wb[p] / c[p] + Wb[q] / c[q])
        alpha = a_k if alpha is None else alpha
        assert abs(a_k - alpha) < 1e-9
    for k in range(nx):
        p = [j for j in range(len(ax)) if ax[j] == k and sg[j] > 0][0]
        v[k] = (bh[p] / c[p] - alpha * Wb[p] / c[p]) / (1 - alpha)
    return alpha, v


def results3_context(orc, P, R):
    """Results_3_* were produced with a 102-row S that is not in the checkout: take
    S_A from the stored S_hat, recover S_b from step 0, and hs by LP over that S."""
    WA, Wb = P["W_A"], P["W_b"]
    T = np.linalg.inv(np.eye(2) - P["Phi"])
    a0, v0 = recover_alpha_v(WA, Wb, R["W_Hat_H"][0][:, -1])
    SA = R["S_Hat_H"][0][:, :-1]
    Sb = (R["S_Hat_H"][0][:, -1] - (1 - a0) * SA @ T @ v0) / a0
    Fc = P["F"] + P["G"] @ P["K"]
    hs = np.array([-linprog(-Fc[i], A_ub=SA, b_ub=Sb, bounds=[(None, None)] * 2, method="highs").fun for i in range(4)])
    d = dict(P)
    d.update(S_A=SA, S_b=Sb, hs=hs)
    kp = _abi.make_params(d)
    ctx = C.c_void_p()
    assert orc.orc_uqmpc_create(kp.ref(), C.byref(ctx)) == 0
    return ctx, kp, d, T


def step(orc, ctx, d, uq, x, a_prev, v_prev, w):
    nx, nc, nS, nW, ncv = 2, 4, d["S_A"].shape[0], 4, 8
    o = dict(s0=np.zeros(nx), c=np.zeros(ncv), u=np.zeros(1), alpha=np.zeros(1), v=np.zeros(nx), hk=np.zeros(nc),
             Sb=np.zeros(nS), Wb=np.zeros(nW))
    st, na, act, it = (np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(_abi.MAX_NZ, np.int32),
                       np.zeros(1, np.int32))
    rc = orc.orc_uqmpc_step(ctx, 1, uq, dptr(fcol(x)), dptr(np.array([a_prev])), dptr(fcol(v_prev)), dptr(fcol(w)),
                            dptr(o["s0"]), dptr(o["c"]), dptr(o["u"]), dptr(o["alpha"]), dptr(o["v"]), dptr(o["hk"]),
                            dptr(o["Sb"]), dptr(o["Wb"]), iptr(st), iptr(na), iptr(act), iptr(it))
    assert rc == 0
    o.update(status=int(st[0]), n_active=int(na[0]), active=act[:int(na[0])].copy(), iters=int(it[0]))
    return o


@pytest.mark.parametrize("tag", ["small", "large"])
def test_results_3_per_step_kats(orc, P, tag):
    """50 stored steps of Case_3: (x_k, b_hat_{k-1}, w_k) -> alpha_k; (alpha_k, v_k) -> S_hat.b, h_k;
    (x_k, h_k, S_hat_k) -> (s_0, u_k)."""
    R = fixtures.npz(f"results_3_{tag}.npz")
    ctx, kp, d, T = results3_context(orc, P, R)
    WA, Wb = P["W_A"], P["W_b"]
    A, B = P["A"], P["B"]
    X, U, S0, HS = R["State_RM"], R["Control_EV"], R["Nominal_RM_State"], R["Hs"]
    av = [recover_alpha_v(WA, Wb, R["W_Hat_H"][k][:, -1]) for k in range(50)]
    n_active_seen = 0
    for k in range(50):
        a_k, v_k = av[k]
        # stage KATs are fed the reference's own (alpha_k, v_k) (hard part 1: v is not unique)
        o = step(orc, ctx, d, 0, X[:, k], a_k, v_k, np.zeros(2))
        assert o["status"] == 0
        np.testing.assert_allclose(o["Sb"], R["S_Hat_H"][k][:, -1], atol=2e-9)
        np.testing.assert_allclose(o["Wb"], R["W_Hat_H"][k][:, -1], atol=1e-12)
        np.testing.assert_allclose(o["hk"], HS[:, k], atol=1.5e-8)
        np.testing.assert_allclose(o["s0"], S0[:, k], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(o["u"], U[:, k], rtol=1e-6, atol=1e-6)
        n_active_seen = max(n_active_seen, o["n_active"])
        if k >= 1:
            # online update LPforUQ (UQRobustMPC.m:124-139): alpha is pinned, v only where binding
            w = X[:, k] - A @ X[:, k - 1] - B[:, 0] * U[0, k - 1]
            o2 = step(orc, ctx, d, 1, X[:, k], av[k - 1][0], av[k - 1][1], w)
            assert abs(o2["alpha"][0] - a_k) < 1e-7
    assert n_active_seen >= 1  # the transient really exercises the active-set path
    orc.orc_destroy(ctx)


def kkt_check(d, ctxdata, x, alpha, v, o, T):
    """independent KKT verification of the oracle's QP answer (primal/dual feasibility, stationarity)"""
    nx = 2
    SA, Sb, H = d["S_A"], d["S_b"], ctxdata
    z = np.concatenate([o["s0"], o["c"]])
    Sb_hat = alpha * Sb + (1 - alpha) * SA @ T @ v
    Cm = np.vstack([np.hstack([-SA, np.zeros((SA.shape[0], z.size - nx))]), H])
    dv = np.concatenate([Sb_hat - SA @ x, np.tile(1 - o["hk"], H.shape[0] // 4)])
    slack = dv - Cm @ z
    assert slack.min() > -1e-9
    Pm = np.zeros((z.size, z.size))
    Pm[:nx, :nx] = d["Px"]
    Pm[nx:, nx:] = d["Pc"]
    act = o["active"]
    if len(act) == 0:
        assert np.abs(z).max() < 1e-12
        return
    assert np.abs(slack[act]).max() < 1e-9
    lam, res, *_ = np.linalg.lstsq(Cm[act].T, -2 * Pm @ z, rcond=None)
    assert np.abs(Cm[act].T @ lam + 2 * Pm @ z).max() < 1e-8
    assert lam.min() > -1e-9


def test_qp_kkt_random_states(orc, P):
    """The oracle's Goldfarb-Idnani QP satisfies the KKT conditions of UQRobustMPC.m:149-155 on states
    spread over the feasible region (and reports infeasibility outside it)."""
    d = dict(P)
    kp = _abi.make_params(d)
    ctx = C.c_void_p()
    assert orc.orc_uqmpc_create(kp.ref(), C.byref(ctx)) == 0
    nH, nz = 4 * (int(P["Nu"]) + 1), 10
    H = np.zeros((nH, nz))
    assert orc.orc_ctx_export(ctx, dptr(H), None, None, None) == 0
    Href = np.vstack([P["F_bar"] @ np.linalg.matrix_power(P["Psi"], i) for i in range(int(P["Nu"]) + 1)])
    np.testing.assert_allclose(H, Href, atol=1e-12)
    T = np.linalg.inv(np.eye(2) - P["Phi"])
    rng = np.random.default_rng(0)
    V = P["F_N_RMPC_V"]
    n_act, n_inf = 0, 0
    for t in range(200):
        lamb = rng.dirichlet(np.ones(3))
        x = (V[rng.integers(0, len(V), 3)] * lamb[:, None]).sum(0) * rng.uniform(0.2, 1.0)
        alpha, v = rng.uniform(0.1, 1.0), rng.uniform(-0.05, 0.05, 2)
        o = step(orc, ctx, d, 0, x, alpha, v, np.zeros(2))
        assert o["status"] == 0, (t, x)
        kkt_check(d, H, x, alpha, v, o, T)
        n_act += o["n_active"] > 0
    assert n_act > 50
    # far outside the feasible region the QP has no solution: the oracle says so (the reference cannot)
    o = step(orc, ctx, d, 0, np.array([40.0, 10.0]), 0.5, np.zeros(2), np.zeros(2))
    assert o["status"] == _abi.ST_QP_INFEASIBLE
    orc.orc_destroy(ctx)


def test_upstream_hs_pins_s_alpha_and_support_sum():
    """parameters.mat (the upstream 2-D parameter set) stores hs = h_S((F+GK)_i) for the S built by
    Functions/MRPISet.m at epsilon = 0.01.  The 2-D (s, alpha) loop (:8-17, mode SUM_LITERAL) gives s = 13,
    alpha = 0.0053213, and the Minkowski-sum support (1-alpha)^-1 sum_{t<s} h_W((Phi^t)' d) reproduces the
    stored hs to IPOPT accuracy: a golden pin of subsystem 1 (support sweep + (s, alpha) search)."""
    import oracle_core
    U = fixtures.params_dict("params_upstream.npz")
    D = np.hstack([np.eye(2), -np.eye(2)])
    Fc = U["F"] + U["G"] @ U["K"]
    s, a, Ms, rc = oracle_core.mrpi_s_alpha(U["Phi"], U["W_V"], U["W_A"], U["W_b"], D, 0.01, mode=_abi.MS_SUM_LITERAL)
    assert (s, rc) == (13, 0) and abs(a - 0.005321295948) < 1e-11
    h = oracle_core.support_batch(U["Phi"], U["W_V"], s, a, Fc.T, transpose=True)
    np.testing.assert_allclose(h, U["hs"].ravel(), atol=1.5e-8)


def test_oracle_openmp_leg_equals_serial():
    """the all-cores CPU baseline (OpenMP over trials) returns exactly what the serial loop returns"""
    import oracle_core
    from lbmpc import _abi
    P = fixtures.params_dict()
    smp = _abi.make_sampler(**fixtures.sampler_args(P))
    c = oracle_core.Context(P)
    kw = dict(trials=97, steps=8, card_I0=10, seed=3, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0, traj_trials=97)
    a, b = c.mc_run(smp, **kw), c.mc_run(smp, threads=4, **kw)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    c.close()


def _feasible_region_opts(P, api):
    S = api.Polyhedron(A=P["S_A"], b=P["S_b"])
    return dict(S=S, S_true=S, S_hat_opt=S, F=P["F"], G=P["G"], K=P["K"], F_bar=P["F_bar"], Psi=P["Psi"],
                nc=int(P["nc"]), nx=int(P["nx"]), nu=int(P["nu"]), N=int(P["N"]), Ps=P.get("Ps"))


def test_feasible_region_golden_parameters_2x2():
    """ComputeFeasibleRegion on the oracle's LP: hs = stored hs; the exact template of F_N = F_Ns (+) S
    contains the stored F_N_RMPC vertices (5236 x 2, the hull of 5000 IPOPT projections (+) S) and is tight against
    them to IPOPT accuracy; Case_5's initial state from the template == from the stored vertices; the polytope-support
    LP agrees with HiGHS; nu_s: the exact criterion gives 9, and every row block 10..19 (the stored Nu) is redundant,
    so the stored Omega(1 - hs, 19) is the same set."""
    import oracle_core
    from lbmpc import api
    P = fixtures.params_dict()
    fr = api.ComputeFeasibleRegion(_feasible_region_opts(P, api), backend=oracle_core.polytope_support)
    F_N, hs, Nu = fr.ComFeasibleRegion_RMPC()
    np.testing.assert_allclose(hs.ravel(), P["hs"].ravel(), atol=1.5e-8)
    V = P["F_N_RMPC_V"]
    assert (F_N.A @ V.T - F_N.b[:, None]).max() < 2e-7          # stored vertices inside the template
    assert (F_N.b - (F_N.A @ V.T).max(1)).max() < 3e-7          # and the template is tight against them
    np.testing.assert_allclose(api.case_5_initial_state(F_N.V), api.case_5_initial_state(V), atol=2e-7)
    assert Nu == 9 and int(P["Nu"]) == 19
    H9, hsv = fr._H(9), hs.ravel()
    later = np.vstack([P["F_bar"] @ np.linalg.matrix_power(P["Psi"], i) for i in range(10, 20)])
    g, st = oracle_core.polytope_support(H9, np.tile(1 - hsv, 10), later.T)
    assert st.max() == 0 and (g <= np.tile(1 - hsv, 10) + 1e-12).all()
    # the literal LP of ComNu_RMPC (M_before only) does not depend on Nu in exact arithmetic
    vals = []
    for nu in (8, 9, 12, 15):
        gl, _ = oracle_core.polytope_support(P["F_bar"] @ np.linalg.matrix_power(P["Psi"], nu), 1 - hsv,
                                             (P["F_bar"] @ np.linalg.matrix_power(P["Psi"], nu + 1)).T)
        vals.append(gl)
    assert np.allclose(vals, vals[0], rtol=1e-7) and (vals[0] > 1 - hsv).any()
    # LP vs HiGHS on random directions of the 10-variable polytope
    rng = np.random.default_rng(4)
    Cd = rng.normal(size=(10, 25))
    h, st = oracle_core.polytope_support(H9, np.tile(1 - hsv, 10), Cd)
    for i in range(25):
        r = linprog(-Cd[:, i], A_ub=H9, b_ub=np.tile(1 - hsv, 10), bounds=[(None, None)] * 10, method="highs")
        assert st[i] == 0 and abs(-r.fun - h[i]) < 1e-9 * max(1.0, abs(h[i]))
    # an unbounded direction is reported, not looped on
    h, st = oracle_core.polytope_support(np.array([[1.0, 0.0]]), np.array([1.0]), np.array([[0.0], [1.0]]))
    assert st[0] == _abi.ST_LP_UNBOUNDED


def test_scenario_mpc_restatement_against_the_uncondensed_qp():
    """ScenarioMPC (parity unpinned vs the reference: no stored SCMPC output exists).  The oracle's
    condensed QP (N inputs, nc*N rows after the max over scenarios) must give the u_0 of the QP exactly as
    ScenarioMPC.m:70-101 poses it: N_SC*nx*(N+1) states tied by the dynamics, N_SC*N*nc rows, cost
    sum_k sum_i x'Qx + u'Ru -- solved here by SLSQP on the inputs with all 1600 rows, scenarios re-drawn through the
    oracle's stand-alone sampler on the same streams."""
    from scipy.optimize import minimize
    from lbmpc._abi import make_scmpc_params, scmpc_mc_run
    P = fixtures.params_dict()
    lib = oracle_ffi.load()
    smp = _abi.make_sampler(**fixtures.sampler_args(P))
    nx, nu, N, nc, N_SC = 2, 1, 8, 4, 50
    Q, R = np.eye(2), np.array([[0.1]])
    keep = make_scmpc_params(nx, nu, N, nc, N_SC, P["A"], P["B"], Q, R, P["F"], P["G"])
    x0 = np.array([-12.0, 5.0])
    rc, o = scmpc_mc_run(lib, "orc_", keep, smp, trials=2, steps=3, seed=6, x0=x0, acc_bound=2.0, distance_error=15.0, traj_trials=2)
    assert rc == 0 and o["status"].max() == 0
    # scenarios of (trial 1, step 0): draws idx = sc*N + i, purposes 6 (u_LV), 7 (xi_EV), 8 (xi_LV)
    def draw(H, V, purpose, n):
        A, b = fcol(H[:, :-1]), fcol(H[:, -1]).ravel()
        lo, hi = fcol(V.min(0)), fcol(V.max(0))
        out, st = np.zeros((n, A.shape[1])), np.zeros(n, np.int32)
        assert lib.orc_sample_polytope(dptr(A), dptr(b), A.shape[0], A.shape[1], dptr(lo), dptr(hi), n, 6, 1, 0, purpose, 1000, dptr(out), iptr(st)) == 0
        return out
    n = N_SC * N
    ev, lv = draw(P["Xi_EV_H"], P["Xi_EV_V"], 7, n), draw(P["Xi_LV_H"], P["Xi_LV_V"], 8, n)
    umin, umax = float(np.ravel(P["min_u_LV"])[0]), float(np.ravel(P["max_u_LV"])[0])
    ul = draw(np.array([[1.0, umax], [-1.0, -umin]]), np.array([[umin], [umax]]), 6, n)
    Bm = np.reshape(P["B"], (2, 1))
    Wsc = (ev - lv - ul @ Bm.T).reshape(N_SC, N, nx)
    A, F, G = P["A"], P["F"], np.reshape(P["G"], (nc, 1))

    def rollout(U):
        X = np.zeros((N_SC, N, nx))
        x = np.tile(x0, (N_SC, 1))
        for i in range(N):
            x = x @ A.T + U[i] * Bm.T + Wsc[:, i]
            X[:, i] = x
        return X

    def cost(U):
        X = rollout(U)
        return float(np.einsum("kia,ab,kib->", X, Q, X) + N_SC * 0.1 * (U ** 2).sum())

    def cons(U):
        X = rollout(U)
        return (1.0 - (np.einsum("ra,kia->kir", F, X) + (G[:, 0][None, None, :] * U[None, :, None]))).ravel()

    r = minimize(cost, np.zeros(N), method="SLSQP", constraints=[{"type": "ineq", "fun": cons}], options={"maxiter": 500, "ftol": 1e-12})
    # SLSQP may stop with "positive directional derivative" at the optimum: judge it by feasibility and by the cost
    assert cons(r.x).min() > -1e-8
    print("SLSQP u0", r.x[0], "oracle u0", o["traj_u"][1, 0, 0], r.message)
    assert abs(r.x[0] - o["traj_u"][1, 0, 0]) < 2e-5 * max(1.0, abs(r.x[0]))
    # the controller keeps the closed loop inside the bounds on this short run
    assert o["infeasible"].sum() == 0


def test_redundancy_lps_against_highs():
    """orc_polytope_redundant (the LPs behind minHRep) vs scipy's HiGHS on a random polytope"""
    import oracle_core
    rng = np.random.default_rng(77)
    n, m = 3, 60
    D = rng.normal(size=(m, n))
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    b = 1.0 + 0.5 * rng.random(m)
    slack, st = oracle_core.polytope_redundant(D, b)
    for j in range(m):
        others = np.arange(m) != j
        res = linprog(-D[j], A_ub=D[others], b_ub=b[others], bounds=[(None, None)] * n, method="highs")
        if res.status == 3:
            assert st[j] == _abi.ST_LP_UNBOUNDED and slack[j] == -np.inf
        else:
            assert res.status == 0 and st[j] == 0
            assert abs((b[j] + res.fun) - slack[j]) < 1e-9, j


def test_compute_Ps_is_the_centred_minimum_volume_ellipsoid():
    """pipeline.compute_Ps: the vertex form (offline_parameters_computation.m:137-154) reproduces the Ps stored in
    parameters_2x2.mat; the row form (nxn script :827-848) satisfies the KKT conditions of its own problem"""
    from lbmpc import pipeline
    g = fixtures.npz("params_2x2.npz")
    Pv = pipeline.compute_Ps(g["F_bar"], g["Psi"], int(g["Nu"]), g["hs"], over="vertices")
    assert np.abs(Pv - g["Ps"]).max() <= 1e-5 * np.abs(g["Ps"]).max()      # SDPT3 accuracy of the stored value
    Ps = pipeline.compute_Ps(g["F_bar"], g["Psi"], int(g["Nu"]))
    H = np.vstack([g["F_bar"] @ np.linalg.matrix_power(g["Psi"], i) for i in range(int(g["Nu"]) + 1)])
    nr = np.linalg.norm(H @ Ps.T, axis=1)
    assert nr.max() <= 1 + 1e-9
    # KKT of max logdet X s.t. h_i' X h_i <= 1: X^-1 = sum lam_i h_i h_i', lam >= 0 supported on the active rows
    from scipy.optimize import nnls
    X = Ps @ Ps
    act = nr > 1 - 1e-6
    Amat = np.stack([np.outer(h, h).ravel() for h in H[act]], 1)
    lam, res = nnls(Amat, np.linalg.inv(X).ravel())
    assert res <= 1e-5 * np.linalg.norm(np.linalg.inv(X))
