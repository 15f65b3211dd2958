"""The three solver tiers of the tube QP (qp_lane.hpp: S-row regime in registers; qp_lane_gen.hpp: one QP per lane with
rows of F_bar Psi^i in play; qp_group.hpp: one QP per 16-lane row) must be interchangeable: whichever tier solves a QP,
status, active set, iteration count and primal solution are the oracle's.  Environment switches (read when a context
is created / a launch is planned): LBMPC_GEN_SOLVER=0 removes the middle tier, LBMPC_LANE_SOLVER=0 the first,
LBMPC_GEN_LANES caps the lanes of the middle tier's LDS region (a wave is then served in several groups)."""
import os

import numpy as np
import pytest

import fixtures
import oracle_core
from lbmpc import _abi, params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def core():
    from lbmpc import core as c
    c.init(0)
    return c


def _env(**kv):
    class E:
        def __enter__(self):
            self.old = {k: os.environ.get(k) for k in kv}
            os.environ.update({k: str(v) for k, v in kv.items()})
        def __exit__(self, *a):
            for k, v in self.old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    return E()


def _system(name):
    if name == "2x2":
        P = fixtures.params_dict()
        return P, _abi.make_sampler(**fixtures.sampler_args(P)), [-12.0, 5.0], 100, 1
    from lbmpc import api
    P = params.load(name)
    return P, params.sampler(P, oriented=True), [float(t) for t in api.case_5_start(P)[0]], 10, 5   # Case_5 near-boundary start


@pytest.mark.parametrize("system,trials,steps", [("2x2", 1500, 30), ("3x3", 900, 20)])
def test_mc_tiers_agree_with_the_oracle_and_each_other(core, system, trials, steps):
    P, smp, x0, card, seed = _system(system)
    kw = dict(trials=trials, steps=steps, card_I0=card, seed=seed, x0=x0, acc_bound=float(np.ravel(P["acc_bound"])[0]),
              distance_error=float(np.ravel(P["distance_error"])[0]), traj_trials=64)
    ctx = core.Context(P)
    ref = ctx.mc_run(smp, **kw)
    cnt = ctx.mc_counters()
    assert cnt["gen_qps"] > 0.1 * cnt["qp_solves"] and cnt["row_solver_qps"] < 0.01 * cnt["qp_solves"] + 1, cnt
    it_hist, na_hist = ctx.mc_histograms()
    assert sum(it_hist) == cnt["qp_solves"] == sum(na_hist)
    assert sum(i * n for i, n in enumerate(it_hist)) == int(ref["qp_iters"].sum())
    ctx.close()
    n = 96
    o = oracle_core.Context(P).mc_run(smp, **{**kw, "trials": n})
    for k in ("infeasible", "status", "qp_iters"):
        np.testing.assert_array_equal(ref[k][:n], o[k], err_msg=k)
    np.testing.assert_allclose(ref["traj_x"], o["traj_x"][:64], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ref["traj_u"], o["traj_u"][:64], rtol=1e-6, atol=1e-6)
    for env in (dict(LBMPC_GEN_SOLVER=0), dict(LBMPC_GEN_LANES=4), dict(LBMPC_GEN_LANES=8, LBMPC_MC_TPW=64),
                dict(LBMPC_LANE_SOLVER=0), dict(LBMPC_LANE_SOLVER=0, LBMPC_GEN_SOLVER=0)):
        with _env(**env):
            c2 = core.Context(P)
            got = c2.mc_run(smp, **kw)
            c2cnt = c2.mc_counters()
            c2.close()
        if env.get("LBMPC_GEN_SOLVER") == 0:
            assert c2cnt["gen_qps"] == 0 and c2cnt["row_solver_qps"] > 0
        for k in ref:
            np.testing.assert_array_equal(ref[k], got[k], err_msg=f"{k} {env}")


@pytest.mark.parametrize("system", ["2x2", "3x3"])
def test_step_tiers_agree(core, system):
    """lbmpc_uqmpc_step on states spread over the feasible region (active sets with up to 8 rows, H rows among them,
    infeasible states): every tier combination returns the oracle's status / active set / iterations"""
    P, smp, x0, card, seed = _system(system)
    nx = int(P["nx"])
    rng = np.random.default_rng(5)
    B = 600
    x0 = np.asarray(x0, dtype=np.float64)
    X = x0[None, :] * rng.uniform(0.0, 1.08, (B, 1)) + rng.normal(size=(B, nx)) * 0.3
    alpha = rng.uniform(0.05, 1.0, B)
    v = rng.uniform(-0.01, 0.01, (B, nx))
    o = oracle_core.Context(P).step(X.T, alpha, v.T, None, uq_update=False)
    assert (o["n_active"] >= 3).sum() > 20 and (o["active_set"].max(1) >= P["S_A"].shape[0]).sum() > 20
    for env in (dict(), dict(LBMPC_GEN_SOLVER=0), dict(LBMPC_GEN_LANES=4), dict(LBMPC_LANE_SOLVER=0)):
        with _env(**env):
            ctx = core.Context(P)
            g = ctx.step(X.T, alpha, v.T, None, uq_update=False)
            ctx.close()
        for k in ("status", "n_active", "active_set", "qp_iters"):
            np.testing.assert_array_equal(g[k], o[k], err_msg=f"{k} {env}")
        ok = o["status"] == 0
        for k in ("s0", "c", "u"):
            np.testing.assert_allclose(g[k][ok], o[k][ok], rtol=1e-6, atol=1e-6, err_msg=f"{k} {env}")


def test_step_allocates_nothing_after_the_first_call(core):
    """the per-step host entry point (one call per closed-loop step of a script, Case_3...:69-72) stages through
    buffers owned by the context: device memory in use does not move between calls"""
    import torch
    P = fixtures.params_dict()
    ctx = core.Context(P)
    x, vp, w = np.array([-8.0, 3.0]), np.array([0.01, -0.005]), np.array([0.05, 0.02])
    first = ctx.step(x, 0.4, vp, w, uq_update=True)
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(50):
        g = ctx.step(x, 0.4, vp, w, uq_update=True)
    assert torch.cuda.mem_get_info()[0] == free0
    for k in first:
        np.testing.assert_array_equal(first[k], g[k], err_msg=k)
    big = ctx.step(np.tile(x[:, None], (1, 4000)), np.full(4000, 0.4), np.tile(vp[:, None], (1, 4000)), np.tile(w[:, None], (1, 4000)), uq_update=True)
    np.testing.assert_array_equal(big["s0"][17], first["s0"])      # a larger batch regrows the staging once, same answers
    np.testing.assert_array_equal(ctx.step(x, 0.4, vp, w, uq_update=True)["u"], first["u"])
    ctx.close()


def test_information_sets_with_more_inputs_than_states(core):
    """nu > nx: every component of the input draw enters B u_LV (the generator's proposal vector has nu entries);
    generated stream == oracle"""
    rng = np.random.default_rng(3)
    args = fixtures.sampler_args(fixtures.params_dict())     # the 2x2 disturbance polytopes, with three inputs
    args.update(nu=3, B=rng.normal(size=(2, 3)), min_u=np.array([-0.05, -0.2, 0.1]), max_u=np.array([0.0625, 0.3, 0.4]))
    smp = _abi.make_sampler(**args)
    WA = np.vstack([np.eye(2), -np.eye(2)])
    S, m, st = core.sample_information_sets(smp, WA, 300, 5, seed=9, set0=2)
    S2, m2, st2 = oracle_core.sample_information_sets(smp, WA, 300, 5, seed=9, set0=2)
    np.testing.assert_array_equal(S, S2)
    np.testing.assert_array_equal(m, m2)
    assert abs(S - S2).max() == 0 and np.ptp(S[:, 0, :]) > 0.2    # the third input's range [0.1, 0.4] shows in the spread


def test_row_pruning_with_a_skewed_proposal_basis(core):
    """lbmpc_sampler.EV_U / LV_U may be any invertible basis: with skewed, badly conditioned columns the rows of the
    polytope in proposal coordinates (a . U_i) cancel to small numbers while the evaluated a . (U t) still rounds at
    the scale |a||U||t| -- the pruning margin of pack_sampler is built from the latter.  Streams == oracle."""
    P = fixtures.params_dict()
    args = fixtures.sampler_args(P)
    arr = {}
    for name in ("EV", "LV"):
        V = np.asarray(args[name + "_V"], dtype=np.float64)
        U = np.array([[1.0, 0.999], [1e-3, 2e-3]]) if name == "EV" else np.array([[3.0, -2.999], [0.5, 0.5003]])
        Tt = np.linalg.solve(U, V.T).T                 # proposal coordinates of the vertices: x = U t
        arr[name] = (np.ascontiguousarray(U), Tt.min(0), Tt.max(0))
    nx, nu = 2, 1
    a = dict(EV_A=_abi.fcol(args["EV_A"]), EV_b=_abi.fcol(args["EV_b"]).ravel(), EV_lo=_abi.fcol(arr["EV"][1]), EV_hi=_abi.fcol(arr["EV"][2]),
             LV_A=_abi.fcol(args["LV_A"]), LV_b=_abi.fcol(args["LV_b"]).ravel(), LV_lo=_abi.fcol(arr["LV"][1]), LV_hi=_abi.fcol(arr["LV"][2]),
             B=_abi.fcol(np.reshape(args["B"], (nx, nu))), min_u_LV=_abi.fcol(np.ravel(args["min_u"])), max_u_LV=_abi.fcol(np.ravel(args["max_u"])),
             EV_U=_abi.fcol(arr["EV"][0]), LV_U=_abi.fcol(arr["LV"][0]))
    s = _abi.Sampler(nx=nx, nu=nu, nEV=a["EV_A"].shape[0], nLV=a["LV_A"].shape[0], **{k: _abi.dptr(v) for k, v in a.items()})
    smp = _abi.Keep(s, a)
    for N, batch, set0 in [(4000, 3, 0), (513, 2, 41)]:
        S, m, st = core.sample_information_sets(smp, P["W_A"], N, batch, seed=17, set0=set0)
        S2, m2, st2 = oracle_core.sample_information_sets(smp, P["W_A"], N, batch, seed=17, set0=set0)
        np.testing.assert_array_equal(S, S2)
        np.testing.assert_array_equal(m, m2)
        np.testing.assert_array_equal(st, st2)
    kw = dict(trials=200, steps=8, card_I0=10, seed=3, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0)
    g, o = core.Context(P).mc_run(smp, **kw), oracle_core.Context(P).mc_run(smp, **kw)
    for k in ("infeasible", "status", "qp_iters"):
        np.testing.assert_array_equal(g[k], o[k], err_msg=k)
    np.testing.assert_allclose(g["x_final"], o["x_final"], rtol=1e-9, atol=1e-12)
