"""C1 / C3 launch time against the trials-per-wave packing (LBMPC_MC_TPW) -- diagnostic."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learning-based-mpc_amd"))
from lbmpc import core, dev, params
device = torch.device("cuda:0")
for name, P, x0, trials, steps, card in (
        ("C1 2x2", params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz")), [-12.0, 5.0], 65536, 50, 100),
        ("C3 3x3 as written", params.load("3x3"), [-14.9, 14.7, 0.0], 10000, 20, 10)):
    ctx = core.Context(P)
    smp = params.sampler(P, oriented=True)
    bufs = dev.McBuffers(ctx, trials, device)
    kw = dict(steps=steps, card_I0=card, seed=1, x0=np.array(x0), acc_bound=float(np.ravel(P["acc_bound"])[0]),
              distance_error=float(np.ravel(P["distance_error"])[0]), trial0=0)
    for tpw in ("auto", "4", "8", "16", "32", "64"):
        if tpw == "auto":
            os.environ.pop("LBMPC_MC_TPW", None)
        else:
            os.environ["LBMPC_MC_TPW"] = tpw
        for _ in range(2):
            dev.mc_run(ctx, smp, bufs, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dev.mc_run(ctx, smp, bufs, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"{name} tpw={tpw}: {dt*1e3:.2f} ms/launch  {trials*steps/dt:.3e} steps/s")
    ctx.close()
