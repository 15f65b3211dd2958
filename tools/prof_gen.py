"""In-kernel phase timing of mc_kernel with the general lane solver (build: make -C learning-based-mpc_amd/csrc prof).  Diagnostic only.
   LEG=C1|C3  LBMPC_LIB=.../prof_lbmpc.so python tools/prof_gen.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learning-based-mpc_amd"))
from lbmpc import core, params, _abi
LEG = os.environ.get("LEG", "C1")
if LEG == "C1":
    P = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz")); x0 = [-12.0, 5.0]; trials, K, card, seed = 65536, 50, 100, 1
elif LEG == "C4t":   # build: make prof4
    P = params.load("4x4"); x0 = [-12.0, 5.0, 0.0, 0.0]; trials, K, card, seed = 131072, 32, 10, 4
else:
    P = params.load("3x3"); x0 = None; trials, K, card, seed = 10000, 20, 10, int(os.environ.get("SEED", 5))
trials = int(os.environ.get("TRIALS", trials))
core.init(0)
if x0 is None:
    from lbmpc import api
    x0 = [float(t) for t in api.case_5_start(P)[0]]
ctx = core.Context(P)
smp = params.sampler(P, oriented=True)
lib = _abi.load()
buf = (ctypes.c_ulonglong * 48)()
lib.lbmpc_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
for rep in range(2):
    lib.lbmpc_prof_read(buf, 1)
    out = ctx.mc_run(smp, trials=trials, steps=K, card_I0=card, x0=x0, seed=seed, acc_bound=float(np.ravel(P["acc_bound"])[0]),
                     distance_error=float(np.ravel(P["distance_error"])[0]))
    lib.lbmpc_prof_read(buf, 0)
v = np.array(list(buf), dtype=np.float64)
cnt = ctx.mc_counters()
tot = v[6]
print(LEG, "waves*ticks total %.3e; slowest wave %.3e ticks" % (tot, v[7]))
for i, n in enumerate(["uq+tighten", "first_pass(ellipsoid)", "heavy region (filter + solves)", "dynamics"]):
    print(f"  {n:32s} {v[i]/tot*100:6.2f} %")
print("  heavy region: S-lane tier %.1f %%  general lane tier %.1f %%  row tier %.1f %% of total" % (v[32] / tot * 100, v[33] / tot * 100, v[34] / tot * 100))
ls = v[8:14]
print("  S-lane tier: scans %.1f %%  step 2 %.1f %%  step 1 other %.1f %% of total; %.0f ticks per scan pass, %.0f step-2 ticks per trip, %.0f step-1 ticks per trip"
      % (ls[0] / tot * 100, ls[1] / tot * 100, ls[2] / tot * 100, ls[0] / max(ls[5], 1), ls[1] / max(ls[4], 1), ls[2] / max(ls[4], 1)))
g = v[24:32]
names = ["step1 candidates", "scans", "new pivot (row, g, cg)", "rhs + solves", "zd (Wm' r)", "ratio test + update", "append", "drop + rebuild"]
gt = g.sum()
for n, x in zip(names, g):
    print(f"    {n:28s} {x/gt*100:6.2f} % of the general solver   {x/max(cnt['gen_trips'],1):8.0f} ticks per trip")
print("  counters:", {k: cnt[k] for k in cnt if k.startswith("gen") or k in ("qp_solves", "row_solver_qps", "lane_trips")})
# per-wave lifetimes (the kernel ends with its slowest wave; a wave's lifetime includes the wait at the block's final barrier)
wb = (ctypes.c_ulonglong * (64 * 4096))()
lib.lbmpc_prof_read_waves.argtypes = [ctypes.c_void_p]
lib.lbmpc_prof_read_waves(wb)
w = np.array(list(wb), dtype=np.float64).reshape(4096, 64)
w = w[w[:, 6] > 0]
life, heavy, busy, trips, scans = w[:, 6], w[:, 2], w[:, 37], w[:, 48 + 11], w[:, 48 + 12]
q = [0, .1, .5, .9, 1]
print("  waves recorded %d; lifetime incl. final barrier (1e6 ticks) min/10/50/90/max:" % len(w), np.round(np.quantile(life, q) / 1e6, 3))
print("  busy time before the barrier (1e6):", np.round(np.quantile(busy, q) / 1e6, 3), " mean %.3f" % (busy.mean() / 1e6))
print("  heavy-region ticks per wave (1e6):", np.round(np.quantile(heavy, q) / 1e6, 3))
print("  general-solver trips per wave:", np.quantile(trips, q), " scan passes per wave:", np.quantile(scans, q))
rowq, rowt = w[:, 48 + 9], w[:, 34]
print("  waves with a row-solver QP: %d; their busy time (1e6) min/50/max:" % int((rowq > 0).sum()), np.round(np.quantile(busy[rowq > 0], [0, .5, 1]) / 1e6, 3) if (rowq > 0).any() else "-",
      " row-tier ticks per row-solver QP %.0f" % (rowt.sum() / max(rowq.sum(), 1)))
if os.environ.get("WAVE_DUMP"): np.save(os.environ["WAVE_DUMP"], w)
gap = w[:, 36]
print("  prologue %.2f %%  between-steps gap %.2f %% of lifetime total; gap per wave (1e6) min/50/90/max:" % (v[35] / tot * 100, v[36] / tot * 100), np.round(np.quantile(gap, [0, .5, .9, 1]) / 1e6, 3))
print("  corr(busy, trips) %.2f  corr(busy, scans) %.2f  corr(heavy, trips) %.2f" % (np.corrcoef(busy, trips)[0, 1], np.corrcoef(busy, scans)[0, 1], np.corrcoef(heavy, trips)[0, 1]))
print("  ticks per trip by wave (heavy / trips) min/10/50/90/max:", np.round(np.quantile(heavy / np.maximum(trips, 1), q)))
nwb = int(os.environ.get("WAVES_PER_BLOCK", 8 if LEG != "C3" else 4))
if len(w) % nwb == 0:
    bb = busy.reshape(-1, nwb)
    print("  per block: max busy / mean busy = %.3f (mean over blocks); slowest block %.3f, mean block max %.3f (1e6); by wave slot mean busy (1e6):" % (
        (bb.max(1) / bb.mean(1)).mean(), bb.max() / 1e6, bb.max(1).mean() / 1e6), np.round(bb.mean(0) / 1e6, 3))
print("  phase sum / lifetime total: %.3f;  kernel ~ slowest wave %.3f vs mean busy %.3f (1e6): balance bound %.2f" % (
    v[0:4].sum() / tot, busy.max() / 1e6, busy.mean() / 1e6, busy.max() / busy.mean()))
