"""Rates of the host-pointer entry points (PCIe copies and allocations included) -- the figures DESIGN.md section 5 quotes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learning-based-mpc_amd"))
from lbmpc import core, params

P2 = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz"))
rng = np.random.default_rng(0)
N, batch = 20000, 2048
w = rng.uniform(-0.1, 0.1, size=(batch, N, 2))   # batch x N x nx = column-major nx x N per set, as the C entry takes it
lib, WA, m = core.init(), core.fcol(P2["W_A"]), np.zeros((batch, 4))
def call():
    core._chk(lib.lbmpc_halfspace_max(core.dptr(WA), 4, 2, core.dptr(w), N, batch, N * 2, core.dptr(m)), "lbmpc_halfspace_max")
call()
t0 = time.perf_counter(); call(); dt = time.perf_counter() - t0
print(f"halfspace_max host entry: {batch} sets x {N} samples ({w.nbytes/1e9:.2f} GB): {dt*1e3:.1f} ms, {batch*N/dt:.3e} samples/s, {w.nbytes/dt/1e9:.1f} GB/s", flush=True)

P4 = params.load("4x4")
ctx = core.Context(P4)
smp = params.sampler(P4, oriented=True)
kw = dict(trials=131072, steps=1000, card_I0=10, seed=4, x0=[-12.0, 5.0, 0.0, 0.0], acc_bound=float(P4["acc_bound"][0]),
          distance_error=float(P4["distance_error"][0]))
ctx.mc_run(smp, **kw)
t0 = time.perf_counter(); ctx.mc_run(smp, **kw); dt = time.perf_counter() - t0
print(f"mc_run host entry: 131072 trials x 1000 steps: {dt*1e3:.1f} ms, {131072*1000/dt:.3e} MPC steps/s", flush=True)
D = rng.normal(size=(4, 1000000)); D /= np.linalg.norm(D, axis=0)
A = 0.6 * rng.normal(size=(4, 4)) / 2; Vw = rng.uniform(-1, 1, size=(16, 4))
core.support_batch(A, Vw, 8, 0.05, D[:, :10])
t0 = time.perf_counter(); core.support_batch(A, Vw, 8, 0.05, D); dt = time.perf_counter() - t0
print(f"support_batch host entry: 1e6 directions, s=8: {dt*1e3:.1f} ms, {1e6/dt:.3e} directions/s", flush=True)
ctx.close()
