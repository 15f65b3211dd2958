#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

metric (BASELINE.json): "closed-loop MPC steps/sec (batched MC trials, 4x4) + samples/sec set learning".
  value            = closed-loop UQ-RMPC steps/s, 4x4 system, Case_4-style Monte-Carlo (config C4): every rank runs
                     --trials trials x --horizon steps per bench step (weak scaling), trials keyed by global id.
  set_learning     = secondary leg (config C2): halfspace max-reduction + homothety LP over pre-materialised
                     information sets resident in HBM (2x2, N=20000 per set), samples/s.
One JSON line on rank 0.  Launch: python bench.py [--gpus N --steps K --warmup W]  (N>1 under torch.distributed.run).
"""
import argparse
import json
import os
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
    ap.add_argument("--cpu-trials", type=int, default=2048)
    ap.add_argument("--sweep-dirs", type=int, default=1000000)
    ap.add_argument("--axis-aligned-proposal", action="store_true",
                    help="rejection-sample xi from the axis-aligned vertex box (the reference's proposal) instead of the principal-axes box")
    args = ap.parse_args()

    from lbmpc import core, dev, dist as ldist, params
    rank, world, local = ldist.init_process_group()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    core.init(local)
    device = torch.device("cuda", local)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- primary leg: 4x4 closed-loop MC (C4)
    P = params.load(args.system)
    nx, nz = int(P["nx"]), int(P["nx"]) + int(P["N"]) * int(P["nu"])
    nS, nH = P["S_A"].shape[0], int(P["nc"]) * (int(P["Nu"]) + 1)
    smp = params.sampler(P, oriented=not args.axis_aligned_proposal)   # same distribution on Xi, fewer rejections
    ctx = core.Context(P)
    x0 = np.zeros(nx)
    x0[0], x0[1] = -12.0, 5.0                      # Case_4...:27 x_RM_0, zero-padded
    trial0, _ = args.trials * rank, None
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
    # ALGORITHMIC flops of one launch on this GPU = the arithmetic the design executes, from the library's work counters
    # (DESIGN.md "Kernel 3"): per MPC step the set update, tightening, ellipsoid test and plant update; per filtered
    # wave-step one fp32 pass over the S rows for 64 trials; per full pivot scan one fp64 pass over the S rows for 4 QPs
    # (+ one pass over the H rows for 2 QPs where the certificate does not skip it); per solver trip the O(nz^2) algebra.
    cnt = ctx.mc_counters()
    nu_ = int(P["nu"]); nc_ = int(P["nc"])
    per_step64 = 2.0 * (nx * nx + nx) + 2.0 * nx * nx + 2.0 * nc_ * nx + 8.0 * nx * nx + 2.0 * nx * nx + 4.0 * nx * nu_
    exec64 = (args.trials * args.horizon * per_step64
              + cnt["pivot_scans"] * 4.0 * 2.0 * (nx + 1) * nS + cnt["h_passes"] * 2.0 * 2.0 * nz * nH
              + cnt["solver_trips"] * 4.0 * 6.0 * nz * nz)
    exec32 = cnt["filter_wave_steps"] * 64.0 * 2.0 * (nx + 1) * nS
    flops = exec64
    achieved = flops / (kern_ms * 1e-3) / 1e12
    # what a solver without the pruning (one S pass per step + one full pass per active-set iteration) would
    # have to execute for the same results: reported for scale only, NOT a roofline position
    naive = args.trials * args.horizon * 2.0 * nS * nx + iters_local * 2.0 * (nS * nx + nH * nz)
    executed = {"fp64_flops_per_launch": exec64, "fp32_flops_per_launch": exec32, "counters": cnt,
                "unpruned_equivalent_flops_per_launch": naive,
                "unpruned_equivalent_tflops": naive / (kern_ms * 1e-3) / 1e12}

    # ---------------------------------------------------------------- secondary leg: set learning (C2)
    P2 = params.load(os.path.join(ROOT, "tests", "golden", "params_2x2.npz"))
    smp2 = params.sampler(P2)
    B, N = args.sl_sets, args.sl_n
    samples = torch.empty((B, N, 2), dtype=torch.float64, device=device)
    m = torch.empty((B, 4), dtype=torch.float64, device=device)
    dev.sample_information_sets(smp2, P2["W_A"], N, B, seed=2, set0=B * rank, samples=samples, m=m)
    torch.cuda.synchronize()
    m_ref = m.clone()
    alpha = torch.empty(B, dtype=torch.float64, device=device)
    v = torch.empty((B, 2), dtype=torch.float64, device=device)

    def sl_step():
        dev.halfspace_max(P2["W_A"], samples, out=m)
        dev.homothety_lp(P2["W_A"], P2["W_b"], m, alpha, v)

    for _ in range(max(1, args.warmup)):
        sl_step()
    barrier()
    sl_steps = max(10, args.steps)
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

    # ---------------------------------------------------------------- third leg: support sweep (C5)
    # h_S(d) over 1M unit directions on the 4x4 system, s from the stored parameters' (s, alpha) loop at eps = 0.1
    Phi4 = np.asarray(P["Phi"], dtype=np.float64)
    Wv = np.asarray(P["W_V"], dtype=np.float64)
    nd = args.sweep_dirs
    gen = torch.Generator(device=device); gen.manual_seed(5 + rank)
    Dd = torch.randn((nd, nx), dtype=torch.float64, device=device, generator=gen)
    Dd /= Dd.norm(dim=1, keepdim=True)
    s_sw, a_sw = int(np.ravel(P["mrpi_s"])[0]), float(np.ravel(P["mrpi_alpha"])[0])
    hout = torch.empty(nd, dtype=torch.float64, device=device)
    for _ in range(max(1, args.warmup)):
        dev.support_batch(Phi4, Wv, s_sw, a_sw, Dd, transpose=True, out=hout)
    barrier()
    sw_steps = max(10, args.steps)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(sw_steps):
        dev.support_batch(Phi4, Wv, s_sw, a_sw, Dd, transpose=True, out=hout)
    e1.record()
    barrier()
    sw_ms = e0.elapsed_time(e1) / sw_steps
    nV = Wv.shape[0]
    sw_flops = nd * (s_sw * nV * (2.0 * nx + 1.0) + s_sw + 1.0)
    sw_bytes = nd * 8.0 * (nx + 1)
    sweep = {"value": nd * world / (sw_ms * 1e-3), "unit": "directions/s", "ms_per_step": sw_ms,
             "config": {"workload": f"C5: 4x4 support sweep h_S(d), {nd} directions/GPU, s={s_sw}, nV={nV}"},
             "roofline": {"bound": "valu_fp64", "achieved": sw_flops / (sw_ms * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS,
                          "unit": "TFLOP/s", "frac": sw_flops / (sw_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                          "hbm_GBps": sw_bytes / (sw_ms * 1e-3) / 1e9, "traffic": None, "kernel": "support_kernel",
                          "algorithmic_flops_per_launch": sw_flops, "algorithmic_bytes_per_launch": sw_bytes}}

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_core
        octx = oracle_core.Context(P)
        ct = args.cpu_trials
        t0 = time.perf_counter()
        o = octx.mc_run(smp, trials=ct, **{**mc_kw, "trial0": 0})
        cdt = time.perf_counter() - t0
        # parity spot-check of the benchmarked launch against the oracle on the same trial ids
        assert np.array_equal(o["infeasible"], bufs.infeasible[:ct].cpu().numpy())
        assert np.array_equal(o["qp_iters"], bufs.qp_iters[:ct].cpu().numpy())
        assert np.allclose(o["alpha_final"], bufs.alpha_final[:ct].cpu().numpy(), rtol=1e-9, atol=0)
        assert np.allclose(o["x_final"], bufs.x_final[:ct].cpu().numpy(), rtol=1e-6, atol=1e-6)
        cpu = {"value": ct * args.horizon / cdt, "unit": "MPC steps/s", "cores": 1, "kind": "port",
               "sample": f"{ct} trials x {args.horizon} steps of the same {args.system} workload (trial ids 0..{ct - 1}), "
                         f"oracle/lbmpc_oracle.c single thread, {cdt:.1f} s; parity of these trials checked against the GPU run",
               "reference_stored_matlab_steps_per_s": 15.4}
    # HBM traffic per launch from the committed PMC pass (tools/profile_round.sh), valid for the default sizes only
    traffic_mc = traffic_sl = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")) as f:
            pk = json.load(f)["kernels"]
        if (B, N) == (8192, 20000):
            k2 = next(v for k, v in pk.items() if "halfspace_max_kernel" in k)
            traffic_sl = k2.get("hbm_read_bytes", 0.0) + k2.get("hbm_write_bytes", 0.0)
        if (args.trials, args.horizon, args.system) == (131072, 1000, "4x4"):
            mck = {k: v for k, v in pk.items() if "mc_" in k}
            nl = min(v["launches"] for v in mck.values())   # profiled bench steps (mc_kernel runs twice per step)
            traffic_mc = sum((v.get("hbm_read_bytes", 0.0) + v.get("hbm_write_bytes", 0.0)) * v["launches"] / nl for v in mck.values())
    except (OSError, KeyError, StopIteration, ValueError):
        pass
    if rank == 0:
        out = {
            "metric": "closed-loop MPC steps/sec (batched MC trials, 4x4) + samples/sec set learning",
            "value": value, "unit": "MPC steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C4: {args.system} UQ-RMPC closed-loop Monte-Carlo, {args.trials} trials/GPU x "
                                   f"{args.horizon} steps, card(I0)={args.card_i0}, build-generated params_{args.system}.npz",
                       "trials_per_gpu": args.trials, "horizon": args.horizon, "nS": int(nS), "nH": int(nH), "nz": int(nz),
                       "parallelism": f"trial-sharded x{world}, no data-path collective",
                       "xi_proposal": "axis-aligned box" if args.axis_aligned_proposal else "principal-axes box"},
            "roofline": {"bound": "valu_fp64", "achieved": achieved, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic_mc, "kernel": "mc_sample_kernel + mc_kernel (transient) + mc_steady_kernel",
                         "kernel_ms": kern_ms, "algorithmic_flops_per_launch": flops, "executed": executed,
                         "note": "MFMA deliberately unused (north_star); bound is the fp64 vector rate. achieved prices the fp64 "
                                 "arithmetic the kernels of one step (pre-pass, transient, steady state; timed together) execute, counted by "
                                 "the library's work counters; the path is instruction-issue/latency bound (DESIGN.md), most "
                                 "instructions are bookkeeping, not flops. 'unpruned_equivalent' is what a solver without the "
                                 "exact pruning (ellipsoid test, H-row certificate, candidate pivots) would execute"},
            "set_learning": {"value": sl_value, "unit": "samples/s", "ms_per_step": sl_ms,
                             "config": {"workload": f"C2: 2x2 set learning, {B} information sets/GPU x {N} samples resident in HBM"},
                             "roofline": {"bound": "hbm", "achieved": sl_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": sl_gbs / HBM_PEAK_GBS, "traffic": traffic_sl, "kernel": "halfspace_max_kernel",
                                          "algorithmic_bytes_per_launch": sl_bytes}},
            "support_sweep": sweep,
            "mc_stats": {"infeasible_trials": int(infeas.sum().item()), "max_status": int(status.max().item()),
                         "mean_qp_iters_per_trial": float(iters.double().mean().item())},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
