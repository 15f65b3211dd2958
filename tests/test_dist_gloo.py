"""world_size-2 (and 3) gloo tests of the multi-GPU layer on CPU: trial sharding with global-id RNG keys and the
one all-reduce(MAX) of sample-sharded information sets give results identical to the unsharded run.  The compute
callables are the CPU oracle here (the sharding logic is what is under test; on GPUs bench.py injects the HIP path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fixtures
import oracle_core
from lbmpc import _abi
from lbmpc import dist as ldist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    ldist.init_process_group(backend="gloo")
    P = fixtures.params_dict()
    smp = _abi.make_sampler(**fixtures.sampler_args(P))
    ctx = oracle_core.Context(P)
    total = 37
    kw = dict(steps=6, card_I0=10, seed=9, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0)

    def run(first, count):
        o = ctx.mc_run(smp, trials=count, trial0=first, **kw)
        return {k: torch.from_numpy(o[k]) for k in ("infeasible", "qp_iters", "alpha_final", "x_final")}

    got = ldist.sharded_mc(run, total)
    # sample-sharded information sets: each rank holds a slice of every set
    rng = np.random.default_rng(0)
    S = rng.uniform(-0.2, 0.2, size=(5, 2, 1001))
    first, count = ldist.block_shard(S.shape[2], rank, world)
    m = ldist.sharded_halfspace_max(lambda x: torch.from_numpy(oracle_core.halfspace_max(P["W_A"], x.numpy())),
                                    torch.from_numpy(S[:, :, first:first + count].copy()))
    if rank == 0:
        full = ctx.mc_run(smp, trials=total, trial0=0, **kw)
        ok = all(np.array_equal(got[k].numpy(), full[k]) for k in got)
        ok = ok and np.array_equal(m.numpy(), oracle_core.halfspace_max(P["W_A"], S))
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharding_is_exact(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_block_shard_partition():
    for total in (0, 1, 7, 100000, 131072):
        for world in (1, 2, 3, 8):
            parts = [ldist.block_shard(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == total
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1
