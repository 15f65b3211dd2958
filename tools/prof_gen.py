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
    P = params.load("3x3"); x0 = None; trials, K, card, seed = 10000, 20, 10, 5
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
