"""bench.py starts its own ranks when asked for N > 1 GPUs without a launcher in the environment (spawn_ranks: one
process per GPU through torch.distributed.run, started before anything touches a GPU).  CPU rehearsal of exactly that
launcher with a gloo worker whose compute is the oracle: 2 and 3 ranks give the single-process results."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np

import fixtures
import oracle_core
from lbmpc import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_bench_rank_worker.py")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_spawned_ranks_reproduce_the_single_process_run(tmp_path):
    bench = _bench()
    trials, steps = 37, 6
    P = fixtures.params_dict()
    smp = _abi.make_sampler(**fixtures.sampler_args(P))
    ref = oracle_core.Context(P).mc_run(smp, trials=trials, steps=steps, card_I0=10, seed=13, x0=[-12.0, 5.0], acc_bound=2.0,
                                        distance_error=15.0)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for n in (2, 3):
        out = tmp_path / f"r{n}.json"
        rc = bench.spawn_ranks(n, WORKER, [str(out), str(trials), str(steps)], env=env)
        assert rc == 0
        got = json.loads(out.read_text())
        assert got["n_gpus"] == n
        np.testing.assert_array_equal(got["infeasible"], ref["infeasible"])
        np.testing.assert_array_equal(got["qp_iters"], ref["qp_iters"])
        np.testing.assert_array_equal(got["alpha_final"], ref["alpha_final"])


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    """under a launcher, --gpus must name the launcher's world size: a clear message instead of an assertion"""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
