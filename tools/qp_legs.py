#!/usr/bin/env python3
"""The QP-active legs of bench.py on their own (C1 2x2 Case_3, C3 3x3 Case_5 as written, C4 transient): launch time by
HIP events, work counters, optional parity spot-check.  Used under rocprofv3 for per-instantiation counters.
  python tools/qp_legs.py [C1] [C3] [C4t] [--reps N] [--check]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "learning-based-mpc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

from lbmpc import core, dev, params

LEGS = {"C1": ("2x2", [-12.0, 5.0], 65536, 50, 100, 1), "C3": ("3x3", None, 10000, 20, 10, 5),   # C3: x0 = Case_5 near-boundary state (api.case_5_start)
        "C4t": ("4x4", [-12.0, 5.0, 0.0, 0.0], 131072, 32, 10, 4),
        "C4": ("4x4", [-12.0, 5.0, 0.0, 0.0], 131072, 1000, 10, 4)}   # the headline launch: pre-pass + transient + steady state


def main():
    names = [a for a in sys.argv[1:] if a in LEGS] or ["C1", "C3"]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 5
    check = "--check" in sys.argv
    core.init(0)
    device = torch.device("cuda", 0)
    for name in names:
        system, x0, trials, steps, card, seed = LEGS[name]
        P = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz")) if system == "2x2" else params.load(system)
        if x0 is None:
            from lbmpc import api
            x0 = [float(t) for t in api.case_5_start(P)[0]]
        ctx = core.Context(P)
        smp = params.sampler(P, oriented=True)
        b = dev.McBuffers(ctx, trials, device)
        kw = dict(steps=steps, card_I0=card, seed=seed, x0=np.asarray(x0), acc_bound=float(np.ravel(P["acc_bound"])[0]),
                  distance_error=float(np.ravel(P["distance_error"])[0]), trial0=0)
        for _ in range(2):
            dev.mc_run(ctx, smp, b, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dev.mc_run(ctx, smp, b, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        cnt = ctx.mc_counters()
        out = {"leg": name, "x0": [float(t) for t in x0], "ms_per_launch": ms, "qp_solves_per_s": cnt["qp_solves"] / (ms * 1e-3), "counters": cnt,
               "infeasible": int(b.infeasible.sum().item()), "max_status": int(b.status.max().item()),
               "iters_sum": int(b.qp_iters.sum().item())}
        if check:
            import oracle_core
            n = 64
            o = oracle_core.Context(P).mc_run(smp, trials=n, **kw)
            for k in ("infeasible", "qp_iters", "status"):
                assert np.array_equal(o[k], getattr(b, k)[:n].cpu().numpy()), (name, k)
            assert np.allclose(o["x_final"], b.x_final[:n].cpu().numpy(), rtol=1e-6, atol=1e-6)
            out["parity"] = f"first {n} trials == oracle"
        print(json.dumps(out), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
