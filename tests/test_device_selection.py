"""Device selection of the host wrappers (round-1 advisor finding: every wrapper used to call lbmpc_init(0), which
re-selected GPU 0 on ranks whose launcher had bound another GPU).  CPU part: with the library's entry points replaced
by recorders, core.init(1) followed by Context(P) and every core.* wrapper must never ask for device 0.  GPU part: a
context created on cuda:1 runs on cuda:1 whatever device is current (skipped on a one-GPU box)."""
import ctypes as C

import numpy as np
import pytest

import fixtures
from lbmpc import _abi, core


class _Recorder:
    """stands in for the CDLL: records lbmpc_init's argument, pretends every other call succeeded"""

    def __init__(self):
        self.init_args = []

    def lbmpc_init(self, d):
        self.init_args.append(int(d))
        return 0

    def lbmpc_last_error(self):
        return b""

    def __getattr__(self, name):
        if name.startswith("lbmpc_"):
            return lambda *a: 0
        raise AttributeError(name)


def test_wrappers_never_select_device_0_implicitly(monkeypatch):
    rec = _Recorder()
    monkeypatch.setattr(_abi, "_lib", rec)
    P = fixtures.params_dict()
    core.init(1)                      # what a rank bound to GPU 1 does first
    ctx = core.Context(P)             # used to call lbmpc_init(0)
    core.halfspace_max(P["W_A"], np.zeros((2, 8)))
    core.homothety_lp(P["W_A"], P["W_b"], np.zeros(4))
    core.support_batch(np.eye(2), P["W_V"], 1, 0.0, np.eye(2))
    ctx.handle = C.c_void_p()         # nothing to destroy
    assert rec.init_args[0] == 1
    assert all(d == _abi.DEVICE_CURRENT for d in rec.init_args[1:]), rec.init_args
    assert 0 not in rec.init_args
    ctx2 = core.Context(P, device=3)  # an explicit device is passed through
    ctx2.handle = C.c_void_p()
    assert rec.init_args[-1] == 3


@pytest.mark.gpu
def test_context_runs_on_its_own_device():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import oracle_core
    P = fixtures.params_dict()
    smp = _abi.make_sampler(**fixtures.sampler_args(P))
    kw = dict(trials=130, steps=6, card_I0=10, seed=9, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0)
    torch.cuda.set_device(1)
    core.init()                        # adopts cuda:1
    ctx1 = core.Context(P)
    torch.cuda.set_device(0)           # caller moves on; the context must still run on cuda:1
    core.init()
    g = ctx1.mc_run(smp, **kw)
    assert torch.cuda.current_device() == 0      # the caller's selection is restored
    o = oracle_core.Context(P).mc_run(smp, **kw)
    for k in ("infeasible", "status", "qp_iters"):
        np.testing.assert_array_equal(g[k], o[k], err_msg=k)
    ctx1.close()


def test_block_shard_rule_matches_the_process_per_gpu_driver():
    """the library's shard arithmetic (lbmpc_multi_mc_run, lbmpc_halfspace_max_multi) == lbmpc/dist.py's; shards are
    contiguous, ordered, cover [0, total) and differ in size by at most one"""
    from lbmpc import dist as ldist
    for total in (0, 1, 7, 10000, 131072, 131073):
        for world in (1, 2, 3, 8):
            nxt = 0
            sizes = []
            for r in range(world):
                first, count = core.block_shard(total, r, world)
                assert (first, count) == ldist.block_shard(total, r, world)
                assert first == nxt
                nxt += count
                sizes.append(count)
            assert nxt == total and max(sizes) - min(sizes) <= 1
    lib = _abi.load()
    assert lib.lbmpc_block_shard(10, 3, 3, None, None) == _abi.ERR_ARG


@pytest.mark.gpu
def test_multi_gpu_entry_points_on_the_devices_present():
    """lbmpc_multi_mc_run / lbmpc_halfspace_max_multi through the C ABI: with every device the box has (one on the
    test box: unmeasured on N > 1 until a multi-GPU record exists), results bitwise those of the single-device entry
    points; traj_trials and ragged shard sizes included"""
    import torch
    ndev = torch.cuda.device_count()
    devs = list(range(ndev))
    P = fixtures.params_dict()
    smp = _abi.make_sampler(**fixtures.sampler_args(P))
    core.init(0)
    kw = dict(trials=1001, steps=9, card_I0=10, seed=13, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0, traj_trials=300)
    ref = core.Context(P).mc_run(smp, **kw)
    for dl in ([0], devs):
        mc = core.MultiContext(P, dl)
        got = mc.mc_run(smp, **kw)
        for k in ref:
            np.testing.assert_array_equal(ref[k], got[k], err_msg=f"{k} devices={dl}")
        mc.close()
    rng = np.random.default_rng(3)
    S = rng.uniform(-0.3, 0.3, size=(7, 2, 5003))
    want = core.halfspace_max(P["W_A"], S)
    for dl in ([0], devs):
        np.testing.assert_array_equal(core.halfspace_max_multi(P["W_A"], S, dl), want)
    assert torch.cuda.current_device() == 0
    with pytest.raises(core.LbmpcError):
        core.MultiContext(P, [0, 0])
    with pytest.raises(core.LbmpcError):
        core.MultiContext(P, list(range(ndev + 1)))


@pytest.mark.gpu
def test_multi_gpu_threads_and_shards_with_one_device_listed_several_times():
    """LBMPC_MULTI_ALLOW_DUP=1 (test switch): the same GPU may appear several times in a device list, so that the code
    a multi-GPU node runs -- one context, one host thread and one block shard per entry, slices of the caller's output
    arrays (ragged shard sizes, traj_trials straddling a shard boundary), sample slices of every information set --
    really executes on the one-GPU test box.  Only the RCCL all-reduce itself is replaced (by a host-side max): RCCL
    needs distinct devices.  Results are bitwise those of the single-device entry points."""
    import os
    P = fixtures.params_dict()
    smp = _abi.make_sampler(**fixtures.sampler_args(P))
    core.init(0)
    kw = dict(trials=1001, steps=9, card_I0=10, seed=13, x0=[-12.0, 5.0], acc_bound=2.0, distance_error=15.0, traj_trials=400)
    ref = core.Context(P).mc_run(smp, **kw)
    rng = np.random.default_rng(4)
    S = rng.uniform(-0.3, 0.3, size=(7, 2, 5003))
    want = core.halfspace_max(P["W_A"], S)
    os.environ["LBMPC_MULTI_ALLOW_DUP"] = "1"
    try:
        for dl in ([0, 0], [0, 0, 0]):
            mc = core.MultiContext(P, dl)          # 1001 = 501 + 500 = 334 + 334 + 333: traj_trials = 400 ends inside shard 0 / 1
            got = mc.mc_run(smp, **kw)
            for k in ref:
                np.testing.assert_array_equal(ref[k], got[k], err_msg=f"{k} devices={dl}")
            mc.close()
            np.testing.assert_array_equal(core.halfspace_max_multi(P["W_A"], S, dl), want)
    finally:
        os.environ.pop("LBMPC_MULTI_ALLOW_DUP", None)
    with pytest.raises(core.LbmpcError):
        core.MultiContext(P, [0, 0])               # without the switch a repeated device stays an error
