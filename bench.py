#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

metric (BASELINE.json): "closed-loop MPC steps/sec (batched MC trials, 4x4) + samples/sec set learning".
  value            = closed-loop UQ-RMPC steps/s, 4x4 system, Case_4-style Monte-Carlo (config C4): every rank runs
                     --trials trials x --horizon steps per bench step (weak scaling), trials keyed by global id.
                     C4 converges to the origin: ~99 % of its steps only pass the ellipsoid test, so the QP-active
                     legs below are reported beside it (qp_active: C4's transient, C1, C3), each with its own counters
                     and a parity spot-check against the oracle.
  set_learning     = config C2: halfspace max-reduction + homothety LP over pre-materialised information sets resident
                     in HBM (2x2, N=20000 per set), samples/s, HBM roofline.
  support_sweep    = config C5: h_S(d) over direction batches at s in {4 (the stored 4x4 loop), 10, 30} and 1M / 10M
                     directions, fp64-vector roofline.
  cpu_baseline     = the oracle (a CPU port, not MATLAB) on a bounded sample of the C4 workload: one thread and all
                     host cores (OpenMP over trials), rank 0 at N=1 only.
  latency          = config C1 at B = 1: one UQRobustMPC.solve per call through the host C ABI and through the MEX
                     gateway (stand-in mex.h), median / p99 microseconds, beside the reference's stored 65.0 / 77.5 ms.
One JSON line on rank 0.  Launch: python bench.py [--gpus N --steps K --warmup W].  With N > 1 and no launcher in the
environment (WORLD_SIZE unset) it starts its own ranks, one process per GPU, through torch.distributed.run BEFORE
anything touches a GPU, and exits with their code; under a launcher it is one of the ranks.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "learning-based-mpc_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector (non-MFMA) peak, BASELINE.md section 2.4
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r03_pmc_summary.json")


def executed_flops(cnt, nx, nu, nc, nz, nS, nH, trial_steps):
    """fp64 arithmetic the closed-loop kernels execute, from the library's work counters (DESIGN.md "Kernel 3"):
    per MPC step the set update (exact box: 2 nx products each for b_hat and W_A w), ellipsoid test and plant update;
    per scan pass one fp64 pass over the S rows for 4 QPs; per trip of the one-QP-per-lane solver the nx x nx algebra
    of 64 lanes; the 16-lane-row solver as in round 1."""
    per_step = 2.0 * (3 * 2 * nx) + 2.0 * (nx * nx + nx) + 2.0 * nx * nu + 2.0 * nx * nx + 2.0 * nx * nu + 4.0 * nx
    lane_trip = 2.0 * (nx * nx                          # rhs = C_A g
                       + nx * (nx + 1) / 2 * nx         # M = C_A Ginv C_A'
                       + nx * (nx - 1) * (nx + 1) / 6   # Cholesky updates
                       + nx * (nx - 1)                  # two triangular solves
                       + nx + nx * nx                   # delta, primal direction
                       + 3 * nx) + 40.0                 # ratio test, updates; sqrt / divisions counted as one each
    # general lane solver (qp_lane_gen.hpp), per trip and lane with about nq = 4 active rows: rhs and primal direction
    # (2 nq nz each, + nq (nx^2 + nz - nx) for Ginv c), two triangular solves, ratio test; per drop-rebuild about as much again
    nq = 4.0
    gen_trip = 2.0 * (2 * nq * nz + nq * (nx * nx + nz - nx) + nq * nq + 2 * nz + 3 * nq) + 40.0
    return (trial_steps * per_step
            + (cnt["pivot_scans"] + cnt["lane_scans"] + cnt.get("gen_scans", 0)) * 4.0 * 2.0 * (nx + 1) * nS
            + (cnt["h_passes"] + cnt.get("gen_h_passes", 0)) * 2.0 * 2.0 * nz * nH
            + cnt["solver_trips"] * 4.0 * 6.0 * nz * nz + cnt["lane_trips"] * 64.0 * lane_trip
            + (cnt.get("gen_trips", 0) + cnt.get("gen_rebuild_trips", 0)) * 32.0 * gen_trip)


def inscribed_ellipsoid(S_A, S_b, iters=3000):
    """matrix ME of an ellipsoid {x : x' ME x <= 1} inside S = {x : S_A x <= S_b} (0 interior): minimum-volume enclosing
    ellipsoid of the points +-a_j / b_j (Khachiyan), polar, then shrunk until every row certifies."""
    Pp = np.asarray(S_A, dtype=np.float64) / np.asarray(S_b, dtype=np.float64)[:, None]
    m, n = Pp.shape
    u = np.full(m, 1.0 / m)
    for _ in range(iters):
        X = (Pp * u[:, None]).T @ Pp
        g = np.einsum("ij,jk,ik->i", Pp, np.linalg.inv(X), Pp)
        j = int(np.argmax(g))
        if g[j] - n < 1e-3 * n:
            break
        step = (g[j] - n) / (n * (g[j] - 1.0))
        u *= 1.0 - step
        u[j] += step
    ME = n * ((Pp * u[:, None]).T @ Pp)
    ME = 0.5 * (ME + ME.T)
    r2 = np.einsum("ij,jk,ik->i", Pp, np.linalg.inv(ME), Pp).max()
    return ME * r2 * (1.0 + 1e-9)


def spawn_ranks(n, script, script_args, env=None):
    """One process per GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... script args` as a child
    of this process, which has not touched a GPU (a process that has must not be replaced or forked into ranks).
    Returns the child's exit code.  tests/test_bench_launcher.py drives this with a CPU worker (gloo)."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script, *script_args]
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=e)


def trim_hist(h):
    """{bin: count} without the empty bins (bin 63 = 63 or more)"""
    return {str(i): int(n) for i, n in enumerate(h) if n}


def lib_sha256():
    from lbmpc import _abi
    with open(_abi.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def latency_leg(core, params, n=2000):
    """C1 at B = 1: one UQRobustMPC.solve (UQRobustMPC.m:68-91; timed by Case_3...:71-74) per call, on the
    stored closed-loop states of Results_3_small (QP-active first steps included), (a) through the host C ABI with
    pre-built arguments, (b) through mexFunction of the MEX gateway compiled against the stand-in mex.h."""
    import ctypes as C
    from lbmpc import _abi
    P1 = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz"))
    g3 = np.load(os.path.join(ROOT, "tests", "golden", "results_3_small.npz"))
    X = np.ascontiguousarray(np.asarray(g3["State_RM"], dtype=np.float64).T[:50])        # 50 stored states x_k
    ctx = core.Context(P1)
    nx, nu, nc, nS, nW = ctx.nx, ctx.nu, ctx.nc, ctx.nS, ctx.nW
    dp = _abi.dptr
    a_in, v_in, w_in = np.array([0.3]), np.zeros(nx), np.zeros(nx)
    o = dict(s0=np.zeros(nx), u=np.zeros(nu), alpha=np.zeros(1), v=np.zeros(nx), hk=np.zeros(nc), Sb=np.zeros(nS), Wb=np.zeros(nW))
    st = np.zeros(1, np.int32)
    lib, h = ctx.lib, ctx.handle
    iptr = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))

    xs = [dp(X[k]) for k in range(50)]      # argument pointers built once: the timed region is the call itself
    pa, pv, pw, pst = dp(a_in), dp(v_in), dp(w_in), iptr(st)
    po = {k: dp(a) for k, a in o.items()}

    def one(k):
        x = xs[k % 50]
        t0 = time.perf_counter_ns()
        rc = lib.lbmpc_uqmpc_step(h, 1, 1, x, pa, pv, pw, po["s0"], None, po["u"], po["alpha"], po["v"],
                                  po["hk"], po["Sb"], po["Wb"], pst, None, None, None)
        t1 = time.perf_counter_ns()
        assert rc == 0
        return (t1 - t0) * 1e-3
    for k in range(200):
        one(k)
    t_abi = np.array([one(k) for k in range(n)])
    out = {"workload": "C1 (2x2, parameters_2x2.mat), B = 1: one UQRobustMPC.solve per call on the 50 stored closed-loop states of "
                       "Results_3_small.mat; 7 outputs incl. S_hat.b (nS doubles); host pointers in, host pointers out",
           "c_abi_us": {"median": float(np.median(t_abi)), "p99": float(np.quantile(t_abi, 0.99)), "min": float(t_abi.min()), "calls": n},
           "reference_stored_ms_per_step": {"Results_3_small": float(np.mean(g3["Time_UQMPC"]) * 1e3),
                                            "Results_3_large": float(np.mean(np.load(os.path.join(ROOT, "tests", "golden", "results_3_large.npz"))["Time_UQMPC"]) * 1e3),
                                            "source": "Time_UQMPC_*, Case_3...:71-74 (MATLAB + CasADi/IPOPT, hardware unknown)"}}
    ctx.close()
    try:   # MEX gateway through the stand-in mex.h (tests/mex_stub): what a MATLAB caller adds on top of the C ABI
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_mex_gateway as tm
        mex = tm.Mex()
        mex.call(0, "init", 0)
        (hd,) = mex.call(1, "create", tm.params_struct(P1))
        L = mex.L
        dims = [nx, nu, nc, nS, nW]
        prhs_sets = []
        for k in range(50):
            args_k = ("step", hd, 1, dims, X[k], 0.3, np.zeros(nx), np.zeros(nx))
            prhs_sets.append((C.c_void_p * len(args_k))(*[mex.to_mx(a) for a in args_k]))
        plhs = (C.c_void_p * 8)()
        err = C.create_string_buffer(1024)

        def one_mex(k):
            pr = prhs_sets[k % 50]
            t0 = time.perf_counter_ns()
            rc = L.stub_call(7, plhs, 8, pr, err, 1024)
            t1 = time.perf_counter_ns()
            assert rc == 0, err.value
            for i in range(7):
                L.mxDestroyArray(plhs[i])
            return (t1 - t0) * 1e-3
        for k in range(200):
            one_mex(k)
        t_mex = np.array([one_mex(k) for k in range(n)])
        out["mex_gateway_us"] = {"median": float(np.median(t_mex)), "p99": float(np.quantile(t_mex, 0.99)), "min": float(t_mex.min()), "calls": n,
                                 "note": "mexFunction('step', ...) with 7 outputs allocated per call, stand-in mex.h (no MATLAB in the image)"}
        mex.call(0, "destroy", hd)
    except Exception as ex:   # the stub library is test infrastructure: its absence must not fail the benchmark
        out["mex_gateway_us"] = {"error": repr(ex)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--trials", type=int, default=131072, help="MC trials per GPU per step (config C4: >= 100k)")
    ap.add_argument("--horizon", type=int, default=1000, help="K_N closed-loop steps per trial (Case_4: 1000)")
    ap.add_argument("--card-i0", type=int, default=10, help="card(I_0^w) (Case_4...:21)")
    ap.add_argument("--system", default="4x4")
    ap.add_argument("--sl-sets", type=int, default=8192, help="information sets per GPU in the set-learning leg")
    ap.add_argument("--sl-n", type=int, default=20000, help="samples per information set")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-trials", type=int, default=1024, help="trials of the single-thread CPU leg (x32 for the all-cores leg)")
    ap.add_argument("--sweep-dirs", type=int, default=1000000)
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the qp_active (C1/C3/C4 transient) and the large sweep legs")
    ap.add_argument("--axis-aligned-proposal", action="store_true",
                    help="rejection-sample xi from the axis-aligned vertex box (the reference's proposal) instead of the principal-axes box")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # no launcher: start the ranks ourselves (nothing has touched a GPU yet)
        sys.exit(spawn_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))

    from lbmpc import core, dev, dist as ldist, params
    rank, world, local = ldist.init_process_group()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local)
    core.init()                      # the device the launcher bound (never device 0 by default)
    device = torch.device("cuda", local)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def time_launches(fn, n, warm=1):
        """n launches between HIP events on the launch stream -> mean ms"""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    # ---------------------------------------------------------------- primary leg: 4x4 closed-loop MC (C4)
    P = params.load(args.system)
    nx, nu_, nc_ = int(P["nx"]), int(P["nu"]), int(P["nc"])
    nz = nx + int(P["N"]) * nu_
    nS, nH = P["S_A"].shape[0], nc_ * (int(P["Nu"]) + 1)
    smp = params.sampler(P, oriented=not args.axis_aligned_proposal)   # same distribution on Xi, fewer rejections
    ctx = core.Context(P)
    x0 = np.zeros(nx)
    x0[0], x0[1] = -12.0, 5.0                      # Case_4...:27 x_RM_0, zero-padded
    trial0 = args.trials * rank
    bufs = dev.McBuffers(ctx, args.trials, device)
    mc_kw = dict(steps=args.horizon, card_I0=args.card_i0, seed=4, x0=x0, acc_bound=float(P["acc_bound"][0]),
                 distance_error=float(P["distance_error"][0]), trial0=trial0)

    def mc_step():
        dev.mc_run(ctx, smp, bufs, **mc_kw)

    for _ in range(args.warmup):
        mc_step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        mc_step()
        b.record()
    barrier()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))   # HIP events on the launch stream
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # gather of the per-trial statistics (the only exchange of this path)
    infeas = ldist.gather_trials(bufs.infeasible)
    iters = ldist.gather_trials(bufs.qp_iters)
    status = ldist.gather_trials(bufs.status)
    total_steps = args.trials * world * args.horizon * args.steps
    value = total_steps / dt
    iters_local = int(bufs.qp_iters.sum().item())
    cnt = ctx.mc_counters()
    it_hist, na_hist = ctx.mc_histograms()
    mc_launch = ctx.mc_plan(args.trials, args.horizon)
    exec64 = executed_flops(cnt, nx, nu_, nc_, nz, nS, nH, args.trials * args.horizon)
    exec32 = cnt["filter_wave_steps"] * 64.0 * 2.0 * (nx + 1) * nS
    achieved = exec64 / (kern_ms * 1e-3) / 1e12
    # what a solver without the pruning (one S pass per step + one full pass per active-set iteration) would
    # have to execute for the same results: reported for scale only, NOT a roofline position
    naive = args.trials * args.horizon * 2.0 * nS * nx + iters_local * 2.0 * (nS * nx + nH * nz)
    executed = {"fp64_flops_per_launch": exec64, "fp32_flops_per_launch": exec32, "counters": cnt,
                "unpruned_equivalent_flops_per_launch": naive,
                "unpruned_equivalent_tflops": naive / (kern_ms * 1e-3) / 1e12}
    # SURVEY 8(d), subsystem 3: 2 (nS nx + nH nz) flops per constraint evaluation, (iterations + 1) evaluations per MPC
    # step.  The kernels return the results of exactly that work (bit-equal to the oracle, which does evaluate every
    # row) WITHOUT executing it -- ellipsoid certificate, fp32 filter, H-row certificate, candidate pivots -- so this is
    # not a lower bound and the fraction exceeds 1: reported as the survey's figure, flagged pruned.
    survey_flops = (args.trials * args.horizon + iters_local) * 2.0 * (nS * nx + nH * nz)
    survey_tflops = survey_flops / (kern_ms * 1e-3) / 1e12
    c4_ref = {k: getattr(bufs, k)[:256].cpu().numpy() for k in ("infeasible", "qp_iters", "alpha_final", "x_final")}

    # ---------------------------------------------------------------- QP-active legs (rank-local, not aggregated)
    qp_active = None
    if not args.no_extra_legs:
        import oracle_core
        qp_active = {}

        def qp_leg(name, Pq, x0q, trials, steps, card, seed, check_trials=48):
            cq = core.Context(Pq)
            sq = params.sampler(Pq, oriented=True)
            bq = dev.McBuffers(cq, trials, device)
            kw = dict(steps=steps, card_I0=card, seed=seed, x0=np.asarray(x0q, dtype=np.float64),
                      acc_bound=float(np.ravel(Pq["acc_bound"])[0]), distance_error=float(np.ravel(Pq["distance_error"])[0]),
                      trial0=0)
            ms = time_launches(lambda: dev.mc_run(cq, sq, bq, **kw), max(5, args.steps), warm=2)
            c = cq.mc_counters()
            it_hist, na_hist = cq.mc_histograms()
            plan = cq.mc_plan(trials, steps)
            parity = None
            if rank == 0:   # spot-check of the timed launch against the oracle on its first trials
                o = oracle_core.Context(Pq).mc_run(sq, trials=check_trials, **kw)
                assert np.array_equal(o["infeasible"], bq.infeasible[:check_trials].cpu().numpy()), name
                assert np.array_equal(o["qp_iters"], bq.qp_iters[:check_trials].cpu().numpy()), name
                assert np.array_equal(o["status"], bq.status[:check_trials].cpu().numpy()), name
                assert np.allclose(o["alpha_final"], bq.alpha_final[:check_trials].cpu().numpy(), rtol=1e-9, atol=0), name
                assert np.allclose(o["x_final"], bq.x_final[:check_trials].cpu().numpy(), rtol=1e-6, atol=1e-6), name
                parity = f"first {check_trials} trials == oracle (flags, statuses, iteration counts exact; alpha 1e-9, x 1e-6)"
            nxq, nuq, ncq = int(Pq["nx"]), int(Pq["nu"]), int(Pq["nc"])
            nzq = nxq + int(Pq["N"]) * nuq
            fl = executed_flops(c, nxq, nuq, ncq, nzq, Pq["S_A"].shape[0], ncq * (int(Pq["Nu"]) + 1), trials * steps)
            out = {"workload": f"{trials} trials x {steps} steps, card(I0)={card}", "ms_per_launch": ms,
                   "mpc_steps_per_s": trials * steps / (ms * 1e-3), "qp_solves_per_s": c["qp_solves"] / (ms * 1e-3),
                   "counters": c, "executed_fp64_tflops": fl / (ms * 1e-3) / 1e12,
                   "qp_iterations_histogram": trim_hist(it_hist), "active_set_size_histogram": trim_hist(na_hist),
                   "launch": plan,
                   "infeasible_trials": int(bq.infeasible.sum().item()), "max_status": int(bq.status.max().item()),
                   "parity": parity}
            cq.close()
            return out

        P1 = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz"))
        qp_active["C1_2x2_case3"] = qp_leg("C1", P1, [-12.0, 5.0], 65536, 50, 100, 1)
        # C3 as written: Case_5...:25-47 -- learn a set, S_hat_ini, hs_UQMPC, F_N_Hat (ComputeFeasibleRegion on the
        # polytope-support LP kernel), x_RM_0 near the boundary of F_N_Hat
        from lbmpc import api
        P3 = params.load("3x3")
        t_fr = time.perf_counter()
        x03, F_N_Hat = api.case_5_start(P3, oriented=True, card_I0=10, seed=5)
        t_fr = time.perf_counter() - t_fr
        qp_active["C3_3x3_case5"] = qp_leg("C3", P3, x03, 10000, 20, 10, 5)
        qp_active["C3_3x3_case5"]["x_RM_0"] = [float(t) for t in x03]
        qp_active["C3_3x3_case5"]["feasible_region_s"] = t_fr
        qp_active["C3_3x3_case5"]["note"] = ("x_RM_0 from the coordinate extremes of F_N_Hat = F_Ns (+) S_hat_ini (Case_5...:45-47), "
                                             f"{F_N_Hat.A.shape[0]} template directions through lbmpc_polytope_support; host time incl. copies")
        qp_active["C4_4x4_transient"] = qp_leg("C4t", P, x0, args.trials, 32, args.card_i0, 4)
        qp_active["C4_4x4_transient"]["note"] = ("the first 32 steps of the C4 launch (pre-pass + transient kernel): every QP of "
                                                 "C4 is solved here; the remaining steps pass the ellipsoid test")

    # ---------------------------------------------------------------- Case_6: UQ-RMPC vs SCMPC (rank 0 only)
    case6 = None
    if not args.no_extra_legs and rank == 0:
        from lbmpc import _abi as abi
        case6 = {}
        for name, Pq in (("2x2", params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz"))), ("3x3", params.load("3x3"))):
            nxq, nuq, Nq, ncq = (int(Pq[k]) for k in ("nx", "nu", "N", "nc"))
            sq = params.sampler(Pq, oriented=True)
            x0q = np.zeros(nxq); x0q[:2] = [-12.0, 5.0]
            tr, K6, NSC = 4096, 50, 400 // Nq          # Case_6...:10,25,53: N_Samples = 400, N_SC = N_Samples / N, K_N = 50
            keep = abi.make_scmpc_params(nxq, nuq, Nq, ncq, NSC, Pq["A"], Pq["B"], np.eye(nxq), 0.1 * np.eye(nuq), Pq["F"], Pq["G"])
            kw6 = dict(trials=tr, steps=K6, seed=6, x0=x0q, acc_bound=float(np.ravel(Pq["acc_bound"])[0]),
                       distance_error=float(np.ravel(Pq["distance_error"])[0]))
            abi.scmpc_mc_run(abi.load(), "lbmpc_", keep, sq, **kw6)
            t0 = time.perf_counter()
            rc6, o6 = abi.scmpc_mc_run(abi.load(), "lbmpc_", keep, sq, **kw6)
            t_sc = time.perf_counter() - t0
            cq = core.Context(Pq)
            bq = dev.McBuffers(cq, tr, device)
            ms_uq = time_launches(lambda: dev.mc_run(cq, sq, bq, steps=K6, card_I0=400, seed=6, x0=x0q, acc_bound=kw6["acc_bound"],
                                                     distance_error=kw6["distance_error"], trial0=0), 3, warm=1)
            case6[name] = {"workload": f"{tr} trials x {K6} steps, N_SC={NSC} scenarios x N={Nq} per SCMPC step, card(I0)=400 for UQ-RMPC",
                           "scmpc_us_per_mpc_step": t_sc / (tr * K6) * 1e6, "scmpc_steps_per_s": tr * K6 / t_sc,
                           "uqrmpc_us_per_mpc_step": ms_uq * 1e-3 / (tr * K6) * 1e6, "uqrmpc_steps_per_s": tr * K6 / (ms_uq * 1e-3),
                           "scmpc_infeasible_trials": int(o6["infeasible"].sum()), "scmpc_max_status": int(o6["status"].max()),
                           "uqrmpc_infeasible_trials": int(bq.infeasible.sum().item()),
                           "note": "the time table of Case_6...:146-150 (per-step solve time of each controller), batched; SCMPC time "
                                   "includes its host<->device copies"}
            cq.close()

    # ---------------------------------------------------------------- offline parameter scripts
    offline_leg = None
    if not args.no_extra_legs and rank == 0:
        from lbmpc import pipeline
        ref_stats = np.load(os.path.join(ROOT, "tests", "golden", "stats.npz"))
        offline_leg = {}
        for nxo in (2, 3, 4):
            pipeline.offline_parameters(nxo)                      # warm-up (first launches, LRU of the support rows)
            t0 = time.perf_counter()
            Po, so = pipeline.offline_parameters(nxo)
            offline_leg[f"{nxo}x{nxo}"] = {
                "seconds": time.perf_counter() - t0, "S_rows": int(Po["S_H"].shape[0]), "Nu": int(Po["Nu"]),
                "reference_stored_seconds": float(np.ravel(ref_stats[f"{nxo}x{nxo}_elapsedTime"])[0]),
                "note": "offline_parameters_computation_nxn.m end to end (sets, tube, feasible regions, Ps) through the library; "
                        "reference: elapsedTime of its own stats_nxn.mat (MATLAB + MPT3 + IPOPT, hardware unknown)"}
    # ---------------------------------------------------------------- per-step latency (C1, B = 1; rank 0 only)
    latency = None
    if not args.no_extra_legs and rank == 0:
        latency = latency_leg(core, params)
    # ---------------------------------------------------------------- set learning (C2)
    P2 = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz"))
    smp2 = params.sampler(P2)
    B, N = args.sl_sets, args.sl_n
    samples = torch.empty((B, N, 2), dtype=torch.float64, device=device)
    m = torch.empty((B, 4), dtype=torch.float64, device=device)
    gen_ms = time_launches(lambda: dev.sample_information_sets(smp2, P2["W_A"], N, B, seed=2, set0=B * rank, samples=samples, m=m), 2, warm=1)
    torch.cuda.synchronize()
    m_ref = m.clone()
    alpha = torch.empty(B, dtype=torch.float64, device=device)
    v = torch.empty((B, 2), dtype=torch.float64, device=device)

    def sl_step():
        dev.halfspace_max(P2["W_A"], samples, out=m)
        dev.homothety_lp(P2["W_A"], P2["W_b"], m, alpha, v)

    for _ in range(max(20, args.warmup)):   # the CPU-side parity checks above leave the GPU idle: bring the clocks back
        sl_step()
    barrier()
    sl_steps = max(50, args.steps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(sl_steps):
        sl_step()
    e1.record()
    barrier()
    sl_dt = time.perf_counter() - t0
    sl_ms = e0.elapsed_time(e1) / sl_steps
    t2 = torch.tensor([sl_dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t2, op=dist.ReduceOp.MAX)
    sl_dt = float(t2.item())
    assert torch.equal(m, m_ref), "halfspace max differs from the maxima computed at generation time"
    sl_value = B * N * world * sl_steps / sl_dt
    sl_bytes = B * N * 2 * 8.0
    sl_gbs = sl_bytes / (sl_ms * 1e-3) / 1e9

    # ---------------------------------------------------------------- support sweep (C5)
    Phi4 = np.asarray(P["Phi"], dtype=np.float64)
    Wv = np.asarray(P["W_V"], dtype=np.float64)
    nV = Wv.shape[0]
    s_sw, a_sw = int(np.ravel(P["mrpi_s"])[0]), float(np.ravel(P["mrpi_alpha"])[0])

    def sweep_leg(nd, s):
        gen = torch.Generator(device=device); gen.manual_seed(5 + rank)
        Dd = torch.randn((nd, nx), dtype=torch.float64, device=device, generator=gen)
        Dd /= Dd.norm(dim=1, keepdim=True)
        hout = torch.empty(nd, dtype=torch.float64, device=device)
        ms = time_launches(lambda: dev.support_batch(Phi4, Wv, s, a_sw, Dd, transpose=True, out=hout), max(10, args.steps), warm=2)
        fl = nd * (s * nV * (2.0 * nx + 1.0) + s + 1.0)
        by = nd * 8.0 * (nx + 1)
        tf = fl / (ms * 1e-3) / 1e12
        return {"value": nd * world / (ms * 1e-3), "unit": "directions/s", "ms_per_step": ms,
                "config": {"workload": f"C5: 4x4 support sweep h_S(d), {nd} directions/GPU, s={s}, nV={nV}"},
                "roofline": {"bound": "valu_fp64", "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "hbm_GBps": by / (ms * 1e-3) / 1e9, "traffic": None,
                             "kernel": "support_kernel", "algorithmic_flops_per_launch": fl,
                             "algorithmic_bytes_per_launch": by}}

    sweep = sweep_leg(args.sweep_dirs, s_sw)
    if not args.no_extra_legs:
        sweep["other_sizes"] = {f"s{s}_nd{nd}": {k: leg[k] for k in ("value", "ms_per_step", "roofline")}
                                for s in (4, 10, 30) for nd in (1000000, 10000000) for leg in [sweep_leg(nd, s)]}
    barrier()

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_core
        octx = oracle_core.Context(P)
        ct = min(args.cpu_trials, 256 * ((args.trials + 255) // 256))
        kw0 = {**mc_kw, "trial0": 0}
        t0 = time.perf_counter()
        o = octx.mc_run(smp, trials=ct, **kw0)
        cdt = time.perf_counter() - t0
        # parity spot-check of the benchmarked launch against the oracle on the same trial ids
        n = min(ct, 256)
        assert np.array_equal(o["infeasible"][:n], c4_ref["infeasible"][:n])
        assert np.array_equal(o["qp_iters"][:n], c4_ref["qp_iters"][:n])
        assert np.allclose(o["alpha_final"][:n], c4_ref["alpha_final"][:n], rtol=1e-9, atol=0)
        assert np.allclose(o["x_final"][:n], c4_ref["x_final"][:n], rtol=1e-6, atol=1e-6)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        threads = max(1, min(avail, int(os.environ.get("OMP_NUM_THREADS", avail))))
        ct_all = ct * min(threads, 8) * 4   # ~13 s on 16 host threads: with the single-thread leg about 20 s of CPU work
        t0 = time.perf_counter()
        o2 = octx.mc_run(smp, trials=ct_all, threads=threads, **kw0)
        cdt_all = time.perf_counter() - t0
        assert np.array_equal(o2["qp_iters"][:ct], o["qp_iters"])
        # the same loop with the certificate the kernels use in front of the QP (VERDICT weak 11: the plain oracle scans
        # all rows at every step, so GPU / CPU mixes pruning with hardware): an ellipsoid inside S, computed here
        ME = inscribed_ellipsoid(P["S_A"], np.ravel(P["S_b"]))
        octx.set_certificate(ME)
        ct_cert = ct_all * 8
        t0 = time.perf_counter()
        o3 = octx.mc_run(smp, trials=ct_cert, threads=threads, **kw0)
        cdt_cert = time.perf_counter() - t0
        octx.set_certificate(None)
        for k in ("infeasible", "qp_iters", "status"):
            assert np.array_equal(o3[k][:ct_all], o2[k]), k
        assert np.array_equal(o3["x_final"][:ct_all], o2["x_final"])
        model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
        except (OSError, StopIteration):
            pass
        cpu = {"value": ct_all * args.horizon / cdt_all, "unit": "MPC steps/s", "cores": threads, "kind": "port",
               "sample": f"{ct_all} trials x {args.horizon} steps of the same {args.system} workload (trial ids 0..{ct_all - 1}), "
                         f"oracle/lbmpc_oracle.c, OpenMP over trials on {threads} threads, {cdt_all:.1f} s; the first {n} trials "
                         f"checked against the GPU run",
               "single_thread": {"value": ct * args.horizon / cdt, "cores": 1,
                                 "sample": f"{ct} trials x {args.horizon} steps, one thread, {cdt:.1f} s"},
               "with_certificate": {"value": ct_cert * args.horizon / cdt_cert, "cores": threads,
                                    "sample": f"{ct_cert} trials x {args.horizon} steps, {threads} threads, {cdt_cert:.1f} s; identical "
                                              f"results (==) to the plain oracle on the first {ct_all} trials",
                                    "note": "oracle + the inscribed-ellipsoid certificate in front of the QP (one of the "
                                            "kernels' exact shortcuts; no fp32 filter, no hinted scans): the fairer CPU figure"},
               "cpu_model": model, "host_cpus_visible": avail,
               "note": "CPU restatement (plain C, no pruning), not MATLAB; the reference's only published figure is "
                       "15.4 MPC steps/s (2x2, MATLAB+CasADi/IPOPT, hardware unknown, Results_3_small.mat)",
               "reference_stored_matlab_steps_per_s": 15.4}
    # HBM traffic per launch from the committed PMC pass of THIS library build (tools/profile_round.sh writes the
    # library's sha256 into the summary); anything else is reported as null rather than as a stale number
    traffic_mc = traffic_sl = traffic_src = None
    try:
        with open(PMC_SUMMARY) as f:
            pj = json.load(f)
        if pj.get("library_sha256") == lib_sha256():
            pk = pj["kernels"]
            traffic_src = "profiles/" + os.path.basename(PMC_SUMMARY) + " (separate rocprofv3 --pmc passes of this library build, default sizes)"
            if (B, N) == (8192, 20000):
                k2 = next(v for k, v in pk.items() if "halfspace_max_kernel" in k)
                traffic_sl = k2.get("hbm_read_bytes", 0.0) + k2.get("hbm_write_bytes", 0.0)
            if (args.trials, args.horizon, args.system) == (131072, 1000, "4x4"):
                mck = {k: v for k, v in pk.items() if "mc_" in k}
                traffic_mc = pj.get("mc_bytes_per_bench_step")
    except (OSError, KeyError, StopIteration, ValueError):
        pass
    if rank == 0:
        out = {
            "metric": "closed-loop MPC steps/sec (batched MC trials, 4x4) + samples/sec set learning",
            "value": value, "unit": "MPC steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C4: {args.system} UQ-RMPC closed-loop Monte-Carlo, {args.trials} trials/GPU x "
                                   f"{args.horizon} steps, card(I0)={args.card_i0}, build-generated params_{args.system}.npz "
                                   f"(the reference's 4x4 blob is missing: parity vs the CPU oracle only)",
                       "trials_per_gpu": args.trials, "horizon": args.horizon, "nS": int(nS), "nH": int(nH), "nz": int(nz),
                       "parallelism": f"trial-sharded x{world}, no data-path collective",
                       "xi_proposal": "axis-aligned box" if args.axis_aligned_proposal else "principal-axes box"},
            "qp_solves_per_s": cnt["qp_solves"] * world / (kern_ms * 1e-3),
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic_mc, "traffic_source": traffic_src,
                         "kernel": "mc_sample_kernel + mc_kernel (transient) + mc_steady_kernel",
                         "kernel_ms": kern_ms, "algorithmic_flops_per_launch": exec64, "executed": executed,
                         "survey_8d_flops_per_launch": survey_flops, "survey_8d_tflops": survey_tflops,
                         "survey_8d_frac": survey_tflops / FP64_VECTOR_PEAK_TFLOPS, "pruned": True,
                         "certificate_only_wave_step_share": 1.0 - cnt["filter_wave_steps"] / max(1, cnt["wave_steps"]),
                         "mpc_steps_with_a_qp_share": cnt["qp_solves"] / float(args.trials * args.horizon),
                         "qp_iterations_histogram": trim_hist(it_hist), "active_set_size_histogram": trim_hist(na_hist),
                         "occupancy": {"mc_kernel": mc_launch,
                                       "note": "waves_per_simd = min(512 / registers_per_lane, waves_per_block x blocks per CU by LDS / 4); "
                                               "lds_bytes_per_cu = dynamic LDS of the one block a CU holds; mc_steady_kernel and "
                                               "mc_sample_kernel: profiles/r03_pmc_summary.json (VGPR_Count, LDS_Block_Size, SQ_WAVES)"},
                         "qp_solves_per_s_whole_launch": cnt["qp_solves"] / (kern_ms * 1e-3),
                         "note": "MFMA deliberately unused (north_star); bound is the fp64 vector rate. achieved prices the fp64 "
                                 "arithmetic the kernels of one step (pre-pass, transient, steady state; timed together) execute, counted by "
                                 "the library's work counters; the path is instruction-issue/latency bound (DESIGN.md), most "
                                 "instructions are bookkeeping, not flops. 'unpruned_equivalent' is what a solver without the "
                                 "exact pruning (ellipsoid test, H-row certificate, candidate pivots) would execute"},
            "qp_active": qp_active,
            "latency": latency,
            "set_learning": {"value": sl_value, "unit": "samples/s", "ms_per_step": sl_ms,
                             "config": {"workload": f"C2: 2x2 set learning, {B} information sets/GPU x {N} samples resident in HBM"},
                             "roofline": {"bound": "hbm", "achieved": sl_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": sl_gbs / HBM_PEAK_GBS, "traffic": traffic_sl, "traffic_source": traffic_src,
                                          "kernel": "halfspace_max_kernel", "algorithmic_bytes_per_launch": sl_bytes},
                             "generate_mode": {"kernel": "sample_sets_kernel", "ms_per_launch": gen_ms,
                                               "samples_per_s": B * N / (gen_ms * 1e-3),
                                               "write_GBps": sl_bytes / (gen_ms * 1e-3) / 1e9,
                                               "note": "Philox + rejection sampling, VALU/int-bound (3 Philox4x32-10 blocks per sample + rejected attempts; lanes advance independently), writes each sample once"}},
            "support_sweep": sweep,
            "case_6_scmpc_comparison": case6,
            "offline_pipeline": offline_leg,
            "mc_stats": {"infeasible_trials": int(infeas.sum().item()), "max_status": int(status.max().item()),
                         "mean_qp_iters_per_trial": float(iters.double().mean().item())},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
