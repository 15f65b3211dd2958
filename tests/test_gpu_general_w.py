"""Closed loop with a prior W that is not an axis-aligned box (UQRobustMPC.m:124-139 takes any polytope W; every
reference configuration uses a box, so there is no stored output: parity unpinned, GPU against the oracle only).
The set update then solves LPforUQ by vertex enumeration (csrc/homothety_gen.hpp; oracle: orc_homothety_general).
W here is the configuration's box cut by diagonal half-spaces, so the stored tube set S stays a valid outer bound."""
import os

import numpy as np
import pytest

import fixtures
import oracle_core
from lbmpc import _abi, params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def core():
    from lbmpc import core as c
    c.init(0)
    return c


def _cut(P, frac=0.8):
    """W = box ∩ {±(x_0 / r_0 + x_1 / r_1) <= 2 frac}: nW = 2 nx + 2 rows, not a box"""
    WA, Wb = np.asarray(P["W_A"], dtype=np.float64), np.ravel(P["W_b"]).astype(np.float64)
    nx = WA.shape[1]
    lo = np.array([-min(Wb[j] / -WA[j, k] for j in range(len(Wb)) if WA[j, k] < 0) for k in range(nx)])
    hi = np.array([min(Wb[j] / WA[j, k] for j in range(len(Wb)) if WA[j, k] > 0) for k in range(nx)])
    r = 0.5 * (hi - lo)
    d = np.zeros(nx)
    d[0], d[1] = 1.0 / r[0], 1.0 / r[1]
    c = 0.5 * (hi + lo)
    Q = dict(P)
    Q["W_A"] = np.vstack([WA, d, -d])
    Q["W_b"] = np.concatenate([Wb, [2 * frac + d @ c, 2 * frac - d @ c]])
    return Q


def _system(name):
    if name == "2x2":
        P = fixtures.params_dict()
        return _cut(P), _abi.make_sampler(**fixtures.sampler_args(P)), [-12.0, 5.0]
    P = params.load(name)
    return _cut(P), params.sampler(P, oriented=True), [-12.0, 5.0] + [0.0] * (int(P["nx"]) - 2)


@pytest.mark.parametrize("system", ["2x2", "3x3", "4x4"])
def test_step_with_a_general_polytope_w(core, system):
    P, smp, x0 = _system(system)
    nx = int(P["nx"])
    ctx, octx = core.Context(P), oracle_core.Context(P)
    assert ctx.nW == 2 * nx + 2
    rng = np.random.default_rng(11)
    B = 96
    x = np.asarray(x0)[:, None] * rng.uniform(0.0, 1.0, (1, B))
    alpha = rng.uniform(0.05, 0.9, B)
    v = np.zeros((nx, B))
    WA, Wb = np.asarray(P["W_A"]), np.ravel(P["W_b"])
    # half the disturbances well inside the current set, half on the boundary of W (outside alpha W)
    w = rng.uniform(-1, 1, (nx, B))
    w *= (rng.uniform(0.0, 1.0, B) * np.where(np.arange(B) % 2 == 0, alpha * 0.5, 1.0)) / np.max((WA @ w) / Wb[:, None], axis=0)
    g = ctx.step(x, alpha, v, w)
    o = octx.step(x, alpha, v, w)
    assert (g["alpha"] != alpha).sum() > B // 4          # the set moved for the outside half
    for k in ("status", "n_active", "qp_iters", "active_set"):
        assert np.array_equal(g[k], o[k]), k
    for k in ("alpha", "v", "Wb_hat", "hk", "s0", "u", "Sb_hat"):
        assert np.array_equal(g[k], o[k]), k
    # the returned set contains the previous one and the new sample (rows of W_hat: W_A x <= Wb_hat)
    assert np.all(np.einsum("jk,kb->bj", WA, w) <= g["Wb_hat"] + 1e-9)
    assert np.all(alpha[:, None] * Wb[None, :] <= g["Wb_hat"] + 1e-9)
    # nominal controller (no set update) on the same context
    gn, on = ctx.step(x, alpha, v, None, uq_update=False), octx.step(x, alpha, v, None, uq_update=False)
    for k in ("alpha", "v", "Wb_hat", "s0", "u", "status", "qp_iters"):
        assert np.array_equal(gn[k], on[k]), k
    ctx.close()


@pytest.mark.parametrize("system,trials,steps,card", [("2x2", 700, 40, 5), ("3x3", 300, 25, 3), ("4x4", 200, 70, 3)])
def test_mc_run_with_a_general_polytope_w(core, system, trials, steps, card):
    P, smp, x0 = _system(system)
    kw = dict(trials=trials, steps=steps, card_I0=card, seed=9, x0=x0, acc_bound=float(np.ravel(P["acc_bound"])[0]),
              distance_error=float(np.ravel(P["distance_error"])[0]), traj_trials=32)
    ctx = core.Context(P)
    g = ctx.mc_run(smp, **kw)
    ctx.close()
    n = 64
    o = oracle_core.Context(P).mc_run(smp, **{**kw, "trials": n})
    for k in ("infeasible", "status", "qp_iters"):
        assert np.array_equal(g[k][:n], o[k]), k
    for k in ("alpha0", "alpha_final", "v_final", "x_final"):
        assert np.array_equal(g[k][:n], o[k]), k
    assert np.array_equal(g["traj_alpha"][:32], o["traj_alpha"][:32])
    # the sets really moved during the run (small card(I0): the first steps fall outside)
    assert np.mean(g["alpha_final"] > g["alpha0"]) > 0.5
