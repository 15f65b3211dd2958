"""One or two trials per lane (LBMPC_MC_T) at the batch sizes where the library switches to two -- informational."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learning-based-mpc_amd"))
from lbmpc import core, dev, params
device = torch.device("cuda:0")
for name, P, x0, trials, steps, card in (
        ("2x2", params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz")), [-12.0, 5.0], 1 << 20, 50, 100),
        ("4x4", params.load("4x4"), [-12.0, 5.0, 0.0, 0.0], 1 << 19, 50, 10),
        ("4x4", params.load("4x4"), [-12.0, 5.0, 0.0, 0.0], 1 << 20, 200, 10)):
    ctx = core.Context(P)
    smp = params.sampler(P, oriented=True)
    bufs = dev.McBuffers(ctx, trials, device)
    kw = dict(steps=steps, card_I0=card, seed=1, x0=np.array(x0), acc_bound=float(np.ravel(P["acc_bound"])[0]),
              distance_error=float(np.ravel(P["distance_error"])[0]), trial0=0)
    for T in ("1", "2"):
        os.environ["LBMPC_MC_T"] = T
        for _ in range(2):
            dev.mc_run(ctx, smp, bufs, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            dev.mc_run(ctx, smp, bufs, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        print(f"{name}: {trials} trials x {steps} steps, T={T}: {dt*1e3:.2f} ms/launch, {trials*steps/dt:.3e} steps/s", flush=True)
    ctx.close()
