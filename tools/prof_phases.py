"""In-kernel phase timing of mc_kernel (build with -DLBMPC_PROFILE; see tools/prof.sh).  Diagnostic only."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "learning-based-mpc_amd"))
from lbmpc import core, params, _abi

SYSTEM = os.environ.get("SYSTEM", "4x4")
P = params.load(SYSTEM)
ctx = core.Context(P)
smp = params.sampler(P, oriented=True)
trials = int(os.environ.get("TRIALS", 131072)); K = int(os.environ.get("K", 1000))
lib = _abi.load()
buf = (ctypes.c_ulonglong * 24)()
lib.lbmpc_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
for rep in range(2):
    lib.lbmpc_prof_read(buf, 1)
    t0 = time.time()
    out = ctx.mc_run(smp, trials=trials, steps=K, card_I0=int(os.environ.get("CARD", 10)), x0=[float(t) for t in os.environ.get("X0", "-12,5,0,0").split(",")], seed=4,
                     acc_bound=float(P["acc_bound"][0]), distance_error=float(P["distance_error"][0]))
    dt = time.time() - t0
    lib.lbmpc_prof_read(buf, 0)
v = np.array(list(buf), dtype=np.float64)
tot = v[6]
names = ["sample_lv+uq+tighten", "first_pass", "solve", "ev+dynamics"]
print(f"wall {dt*1e3:.1f} ms (incl. copies); waves*cycles total {tot:.3e}")
for i, n in enumerate(names):
    print(f"  {n:24s} {v[i]/tot*100:6.2f} %")
print(f"  flagged solves {v[4]:.0f}  group rounds {v[5]:.0f}  mean iters/trial {np.mean(out['qp_iters']):.2f}")
if v[5] > 0:
    print(f"  cycles per group round {v[2]/v[5]:.0f}")
g = v[8:16]
print("  solver: scan %.1f%%  step2 %.1f%%  add %.1f%%  drop %.1f%% of solve; loop trips %.0f scans %.0f drop-phases %.0f" % (
    g[0] / v[2] * 100, g[1] / v[2] * 100, g[2] / v[2] * 100, g[3] / v[2] * 100, g[4], g[5], g[6]))
if g[4] > 0:
    print("  cycles: per scan %.0f  per step2 %.0f  per add %.0f  per drop-phase %.0f" % (g[0] / max(g[5], 1), g[1] / g[4], g[2] / g[4], g[3] / max(g[6], 1)))
print("  per wave-step ticks: Bu_lv %.0f  uq_update %.0f  tighten %.0f  first_pass %.0f  ev+dyn %.0f" % tuple(
    v[i] / (trials / 64 * K) for i in (16, 17, 18, 1, 3)))
print("  wave lifetime: mean %.3e ticks, max %.3e ticks (kernel ends with the slowest wave)" % (v[6] / (trials / 64), v[7]))
wb = (ctypes.c_ulonglong * (64 * 4096))()
lib.lbmpc_prof_read_waves.argtypes = [ctypes.c_void_p]
lib.lbmpc_prof_read_waves(wb)
w = np.array(list(wb), dtype=np.float64).reshape(4096, 64)[: trials // 64]
life, solve, rounds = w[:, 6], w[:, 2], w[:, 5]
print("  lifetime quantiles (1e6 ticks) min/50/90/99/max:", np.round(np.quantile(life, [0, .5, .9, .99, 1]) / 1e6, 2))
print("  solve ticks quantiles (1e6):", np.round(np.quantile(solve, [0, .5, .9, .99, 1]) / 1e6, 2), " rounds:", np.quantile(rounds, [0, .5, 1]))
print("  corr(lifetime, rounds) %.2f  corr(lifetime, solve) %.2f" % (np.corrcoef(life, rounds)[0, 1], np.corrcoef(life, solve)[0, 1]))
blk = life.reshape(-1, 8)
print("  per-block max/mean lifetime: mean of ratio %.3f; by wave slot mean (1e6):" % (blk.max(1) / blk.mean(1)).mean(), np.round(blk.mean(0) / 1e6, 2))
per_round = solve / np.maximum(rounds, 1)
print("  ticks per round by wave: min/50/max", np.round(np.quantile(per_round, [0, .5, 1]) / 1e3, 1), "k")
sb = solve.reshape(-1, 8); ob = (life - solve).reshape(-1, 8)
print("  by wave slot: solve (1e6)", np.round(sb.mean(0) / 1e6, 2), " other (1e6)", np.round(ob.mean(0) / 1e6, 2))
