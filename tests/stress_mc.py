"""GPU vs oracle closed loops over varied seeds, start states, card(I0) and proposal forms (steady-state kernel included) -- run by hand (python tests/stress_mc.py on a GPU box); it lives under tests/ because it uses the oracle."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "learning-based-mpc_amd"))
import oracle_core
from lbmpc import core, params
core.init(0)
for system, x0s in (("2x2", [[-12.0, 5.0], [3.0, -2.0], [-14.0, 1.0]]), ("3x3", [[-12.0, 5.0, 0.0], [-14.9, 14.7, 0.0], [2.0, 2.0, -1.0]]),
                    ("4x4", [[-12.0, 5.0, 0.0, 0.0], [20.0, -10.0, 5.0, 1.0], [-50.0, 30.0, -10.0, 5.0]])):
    P = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz")) if system == "2x2" else params.load(system)
    for oriented in (True, False):
        smp = params.sampler(P, oriented=oriented)
        ctx, octx = core.Context(P), oracle_core.Context(P)
        for seed, x0 in enumerate(x0s):
            kw = dict(trials=192, steps=90, card_I0=7 + 13 * seed, seed=100 + seed, x0=x0, acc_bound=float(np.ravel(P["acc_bound"])[0]),
                      distance_error=float(np.ravel(P["distance_error"])[0]), trial0=1000 * seed)
            g, o = ctx.mc_run(smp, **kw), octx.mc_run(smp, threads=8, **kw)
            for k in ("infeasible", "status", "qp_iters"):
                assert np.array_equal(g[k], o[k]), (system, oriented, seed, k)
            assert np.allclose(g["alpha_final"], o["alpha_final"], rtol=1e-9, atol=0), (system, seed)
            assert np.allclose(g["x_final"], o["x_final"], rtol=1e-6, atol=1e-6), (system, seed)
            print(system, "oriented" if oriented else "axis", seed, "ok: infeasible", int(g["infeasible"].sum()), "iters", int(g["qp_iters"].sum()), "status", int(g["status"].max()))
        ctx.close(); octx.close()
# larger batches: 8-wave blocks (two waves per SIMD: the issue-priority scheme), few lanes per set in the I0 pre-pass
for system, x0, card in (("2x2", [-12.0, 5.0], 100), ("3x3", [-12.0, 5.0, 0.0], 37), ("4x4", [-12.0, 5.0, 0.0, 0.0], 10)):
    P = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz")) if system == "2x2" else params.load(system)
    smp = params.sampler(P, oriented=True)
    ctx, octx = core.Context(P), oracle_core.Context(P)
    kw = dict(trials=16384, steps=40, card_I0=card, seed=31, x0=x0, acc_bound=float(np.ravel(P["acc_bound"])[0]),
              distance_error=float(np.ravel(P["distance_error"])[0]), trial0=77)
    g, o = ctx.mc_run(smp, **kw), octx.mc_run(smp, threads=16, **kw)
    for k in ("infeasible", "status", "qp_iters"):
        assert np.array_equal(g[k], o[k]), (system, k)
    assert np.allclose(g["alpha_final"], o["alpha_final"], rtol=1e-9, atol=0), system
    assert np.allclose(g["x_final"], o["x_final"], rtol=1e-6, atol=1e-6), system
    print(system, "16384 trials ok: plan", ctx.mc_plan(16384, 40) if hasattr(ctx, "mc_plan") else "", "iters", int(g["qp_iters"].sum()))
    ctx.close(); octx.close()
