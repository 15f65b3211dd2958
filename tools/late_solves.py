import sys, numpy as np
sys.path.insert(0, "learning-based-mpc_amd")
from lbmpc import core, params
P = params.load("4x4"); smp = params.sampler(P, oriented=True)
ctx = core.Context(P)
n, K = 8192, 1000
o = ctx.mc_run(smp, trials=n, steps=K, card_I0=10, seed=4, x0=[-12, 5, 0, 0], acc_bound=float(P["acc_bound"][0]),
               distance_error=float(P["distance_error"][0]), trial0=0, traj_trials=n)
s0 = np.asarray(o["traj_s0"]).reshape(n, K, 4)
nz = (np.abs(s0).sum(axis=2) > 0)
last = np.where(nz.any(axis=1), K - 1 - np.argmax(nz[:, ::-1], axis=1), -1)
print("last step with s0 != 0: quantiles", np.quantile(last, [0, .5, .9, .99, 1]))
for k1 in (12, 16, 24, 32, 64):
    print("K1 =", k1, "trials with a nonzero solve at k >= K1:", int((nz[:, k1:].any(axis=1)).sum()), "of", n)
print(ctx.mc_counters())
