"""lbmpc_halfspace_max_noalloc (the entry point for callers that cannot allocate): through the C ABI, against the
existing lbmpc_halfspace_max_dev, bit for bit (maxima are order-free, and both entry points share kernels and chunk
plan), on two calls in a row, the second with a smaller input (the static scratch keeps the first call's partials).
Collected last on purpose (file name): the round's end runs pytest -x."""
import numpy as np
import pytest

import oracle_core
from lbmpc import _abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    from lbmpc import core, dev
    core.init(0)
    return torch, core, dev


def _sets(torch, rng, batch, N, nx):
    S = rng.uniform(-0.3, 0.3, size=(batch, N, nx))
    return S, torch.from_numpy(S).to("cuda:0")


def test_noalloc_matches_dev_two_calls_in_a_row(env):
    torch, core, dev = env
    rng = np.random.default_rng(77)
    WA = np.vstack([np.eye(2), -np.eye(2)]) * np.array([[1.0, 2.5]])
    # first call: 33 sets x 20000 samples -> every set is split over many waves (partials in the static scratch)
    S1, d1 = _sets(torch, rng, 33, 20000, 2)
    want1 = dev.halfspace_max(WA, d1)
    got1 = torch.full((33, 4), float("nan"), dtype=torch.float64, device="cuda:0")
    free0 = torch.cuda.mem_get_info()[0]
    dev.halfspace_max_noalloc(WA, d1, got1)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0          # nothing was allocated on the device
    assert torch.equal(got1, want1)
    np.testing.assert_array_equal(got1.cpu().numpy(), oracle_core.halfspace_max(WA, np.transpose(S1, (0, 2, 1))))
    # second call, smaller input, same GPU: stale partials of the first call must not leak into the result
    S2, d2 = _sets(torch, rng, 5, 1000, 2)
    d2 *= 0.01                                              # maxima far below the first call's
    want2 = dev.halfspace_max(WA, d2)
    got2 = torch.full((5, 4), float("nan"), dtype=torch.float64, device="cuda:0")
    dev.halfspace_max_noalloc(WA, d2, got2)
    torch.cuda.synchronize()
    assert torch.equal(got2, want2)
    np.testing.assert_array_equal(got2.cpu().numpy(), oracle_core.halfspace_max(WA, 0.01 * np.transpose(S2, (0, 2, 1))))
    assert float(got2.max()) < float(got1.min())


@pytest.mark.parametrize("nx,nW,N,batch", [(4, 8, 4097, 11), (3, 6, 777, 9), (1, 2, 513, 4), (2, 6, 999, 3),
                                           (2, 4, 64, 16500), (2, 4, 1, 3)])
def test_noalloc_shapes(env, nx, nW, N, batch):
    """every nx, box and generic row counts, one-wave sets (no scratch needed) and split sets"""
    torch, core, dev = env
    rng = np.random.default_rng(nx * 100 + N)
    WA = rng.normal(size=(nW, nx))
    S, d = _sets(torch, rng, batch, N, nx)
    got = torch.empty((batch, nW), dtype=torch.float64, device="cuda:0")
    dev.halfspace_max_noalloc(WA, d, got)
    assert torch.equal(got, dev.halfspace_max(WA, d))
    np.testing.assert_array_equal(got.cpu().numpy()[:3], oracle_core.halfspace_max(WA, np.transpose(S[:3], (0, 2, 1))))


def test_noalloc_rejects_beyond_the_static_scratch(env):
    """9000 sets x 2 chunks x 16 rows = 288000 partial maxima > LBMPC_NOALLOC_MAX_PARTIALS: clear error, nothing
    launched; the allocating entry point takes the same input"""
    torch, core, dev = env
    rng = np.random.default_rng(5)
    WA = rng.normal(size=(16, 4))
    d = torch.from_numpy(rng.uniform(-1, 1, size=(9000, 512, 4))).to("cuda:0")
    out = torch.zeros((9000, 16), dtype=torch.float64, device="cuda:0")
    with pytest.raises(core.LbmpcError, match="exceed the static scratch") as ei:
        dev.halfspace_max_noalloc(WA, d, out)
    assert ei.value.code == _abi.ERR_DIM
    assert float(out.abs().max()) == 0.0
    ref = dev.halfspace_max(WA, d)
    # the same sets in two halves fit
    dev.halfspace_max_noalloc(WA, d[:4500], out[:4500])
    dev.halfspace_max_noalloc(WA, d[4500:], out[4500:])
    assert torch.equal(out, ref)


def test_noalloc_call_can_be_captured_in_a_graph_and_replayed(env):
    """the callers this entry point is for include code under HIP graph capture: the call is captured on a side
    stream (an allocation, a synchronisation or a copy from pageable memory inside it would abort the capture), the
    graph is replayed on new sample values in the same buffers, and every replay equals the plain call"""
    torch, core, dev = env
    rng = np.random.default_rng(5)
    WA = np.vstack([np.eye(2), -np.eye(2)]) * np.array([[1.0, 2.5]])
    S, d = _sets(torch, rng, 9, 6000, 2)
    out = torch.full((9, 4), float("nan"), dtype=torch.float64, device="cuda:0")
    dev.halfspace_max_noalloc(WA, d, out)          # first call outside the capture (one-time work of the entry point)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        dev.halfspace_max_noalloc(WA, d, out)
    for scale in (1.0, 0.25, 3.0):
        d.copy_(torch.from_numpy(S * scale).to("cuda:0"))
        out.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, dev.halfspace_max(WA, d)), scale
