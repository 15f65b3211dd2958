"""set-learning leg timing breakdown -- diagnostic"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learning-based-mpc_amd"))
from lbmpc import core, dev, params
core.init(0)
P2 = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz"))
smp2 = params.sampler(P2)
B, N = 8192, 20000
device = torch.device("cuda:0")
samples = torch.empty((B, N, 2), dtype=torch.float64, device=device)
m = torch.empty((B, 4), dtype=torch.float64, device=device)
dev.sample_information_sets(smp2, P2["W_A"], N, B, seed=2, set0=0, samples=samples, m=m)
torch.cuda.synchronize()
alpha = torch.empty(B, dtype=torch.float64, device=device); v = torch.empty((B, 2), dtype=torch.float64, device=device)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) / n * 1e3
print("hmax_dev   ", t(lambda: dev.halfspace_max(P2["W_A"], samples, out=m)))
print("hmax_noalloc", t(lambda: dev.halfspace_max_noalloc(P2["W_A"], samples, m)))
print("lp         ", t(lambda: dev.homothety_lp(P2["W_A"], P2["W_b"], m, alpha, v)))
print("both       ", t(lambda: (dev.halfspace_max(P2["W_A"], samples, out=m), dev.homothety_lp(P2["W_A"], P2["W_b"], m, alpha, v))))
