"""Rank body for tests/test_bench_launcher.py: what bench.py's ranks do, on CPU -- join the process group the launcher
set up (gloo), run this rank's block shard of a small closed-loop Monte-Carlo with the ORACLE as compute, gather the
per-trial statistics, max-reduce the time, and let rank 0 write one JSON line.  Test infrastructure only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "learning-based-mpc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.distributed as dist

import fixtures
import oracle_core
from lbmpc import _abi, dist as ldist


def main():
    out_path, trials, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world, local = ldist.init_process_group(backend="gloo")
    P = fixtures.params_dict()
    smp = _abi.make_sampler(**fixtures.sampler_args(P))
    octx = oracle_core.Context(P)

    def run(first, count):
        o = octx.mc_run(smp, trials=count, trial0=first, steps=steps, card_I0=10, seed=13, x0=[-12.0, 5.0], acc_bound=2.0,
                        distance_error=15.0)
        return {k: torch.from_numpy(o[k]) for k in ("infeasible", "qp_iters", "alpha_final")}
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    res = ldist.sharded_mc(run, trials)
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump({"n_gpus": world, "seconds": float(dt.item()), "infeasible": res["infeasible"].tolist(),
                       "qp_iters": res["qp_iters"].tolist(), "alpha_final": res["alpha_final"].tolist()}, f)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
