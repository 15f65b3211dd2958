"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm).

The path shards without a data-path exchange:
  * Monte-Carlo trials: contiguous block partition by global trial id; the RNG stream is keyed by the global id,
    so results are identical for any world size.  The only collective is the gather of per-trial statistics.
  * one information set sharded by samples: each rank reduces its slice to nW maxima, then ONE all-reduce(MAX)
    of batch*nW doubles (max is associative/commutative => bit-identical to a single GPU).
The compute callables are injected, so the world_size-2 gloo tests on CPU can drive this logic with the oracle.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_process_group(backend=None):
    rank, world, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def block_shard(total, rank, world):
    """contiguous block partition -> (first, count); sizes differ by at most one (same rule as the library's
    lbmpc_block_shard, which the in-process multi-GPU entry points use; tests compare the two)"""
    q, r = divmod(int(total), int(world))
    first = rank * q + min(rank, r)
    return first, q + (1 if rank < r else 0)


def allreduce_max_(t):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def sharded_halfspace_max(local_max_fn, local_samples, world_check=True):
    """local_samples: this rank's slice [batch, N_local, nx] of every information set.  Returns the global maxima
    [batch, nW] on every rank."""
    m = local_max_fn(local_samples)
    return allreduce_max_(m)


def gather_trials(t, counts=None):
    """concatenate per-trial tensors of all ranks in rank order (ragged allowed) -> full tensor on every rank"""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return t
    world = dist.get_world_size()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    ns = [int(x.item()) for x in ns]
    mx = max(ns)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:k] for o, k in zip(outs, ns)], dim=0)


def sharded_mc(run_fn, total_trials):
    """run_fn(trial0, count) -> dict of per-trial tensors for this rank's shard; returns the gathered dict."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    first, count = block_shard(total_trials, rank, world)
    local = run_fn(first, count)
    return {k: gather_trials(v) for k, v in local.items()}
