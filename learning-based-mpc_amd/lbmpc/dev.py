"""Device-resident front of the C ABI: operands are torch CUDA tensors (torch is only the allocator / stream /
process-group plumbing), results stay in HBM, launches go to torch's current stream and do not synchronise."""
import ctypes as C

import numpy as np
import torch

from . import _abi, core
from ._abi import as_ptr, dptr, fcol


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f64(t):
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
    return as_ptr(t.data_ptr())


def halfspace_max(WA, samples, out=None):
    """samples: CUDA fp64 tensor [batch, N, nx] (each set column-major nx x N) -> m [batch, nW]"""
    lib = core.init()
    WA = fcol(WA)
    nW, nx = WA.shape
    batch, N, nx2 = samples.shape
    assert nx2 == nx
    if out is None:
        out = torch.empty((batch, nW), dtype=torch.float64, device=samples.device)
    core._chk(lib.lbmpc_halfspace_max_dev(dptr(WA), nW, nx, _f64(samples), N, batch, N * nx, _f64(out), _stream()),
              "lbmpc_halfspace_max_dev")
    return out


def halfspace_max_noalloc(WA, samples, out):
    """lbmpc_halfspace_max_noalloc: as halfspace_max, but `out` is mandatory and the library allocates nothing"""
    lib = core.init()
    WA = fcol(WA)
    nW, nx = WA.shape
    batch, N, nx2 = samples.shape
    assert nx2 == nx and out.shape == (batch, nW)
    core._chk(lib.lbmpc_halfspace_max_noalloc(dptr(WA), nW, nx, _f64(samples), N, batch, N * nx, _f64(out), _stream()),
              "lbmpc_halfspace_max_noalloc")
    return out


def homothety_lp(WA, Wb, m, alpha=None, v=None):
    lib = core.init()
    WA, Wb = fcol(WA), fcol(Wb).ravel()
    nW, nx = WA.shape
    batch = m.shape[0]
    alpha = torch.empty(batch, dtype=torch.float64, device=m.device) if alpha is None else alpha
    v = torch.empty((batch, nx), dtype=torch.float64, device=m.device) if v is None else v
    core._chk(lib.lbmpc_homothety_lp_dev(dptr(WA), dptr(Wb), nW, nx, _f64(m), batch, _abi.TIE_CENTER, _f64(alpha),
                                         _f64(v), None, None, _stream()), "lbmpc_homothety_lp_dev")
    return alpha, v


def sample_information_sets(sampler, WA, N, batch, seed, set0=0, max_attempts=100, samples=None, m=None):
    lib = core.init()
    WA = fcol(WA)
    nW, nx = WA.shape
    dev = torch.device("cuda", torch.cuda.current_device())
    m = torch.empty((batch, nW), dtype=torch.float64, device=dev) if m is None else m
    core._chk(lib.lbmpc_sample_information_sets_dev(sampler.ref(), dptr(WA), nW, N, batch, seed, set0, max_attempts,
                                                    None if samples is None else _f64(samples), _f64(m), None,
                                                    _stream()), "lbmpc_sample_information_sets_dev")
    return m


def support_batch(A, Vw, s, alpha, D, transpose=True, out=None):
    """D: CUDA fp64 tensor [nd, nx] (column-major nx x nd) -> h [nd]"""
    lib = core.init()
    A, Vw = fcol(A), fcol(Vw)
    nd, nx = D.shape
    out = torch.empty(nd, dtype=torch.float64, device=D.device) if out is None else out
    core._chk(lib.lbmpc_support_batch_dev(dptr(A), nx, dptr(Vw), Vw.shape[0], int(s), float(alpha), _f64(D), nd,
                                          int(bool(transpose)), _f64(out), _stream()), "lbmpc_support_batch_dev")
    return out


class McBuffers:
    """Per-trial result arrays in HBM for lbmpc_mc_run_dev."""

    def __init__(self, ctx, trials, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.trials = trials
        self.infeasible = torch.zeros(trials, dtype=torch.int32, device=dev)
        self.status = torch.zeros(trials, dtype=torch.int32, device=dev)
        self.qp_iters = torch.zeros(trials, dtype=torch.int32, device=dev)
        self.alpha_final = torch.zeros(trials, dtype=torch.float64, device=dev)
        self.alpha0 = torch.zeros(trials, dtype=torch.float64, device=dev)
        self.v_final = torch.zeros((trials, ctx.nx), dtype=torch.float64, device=dev)
        self.x_final = torch.zeros((trials, ctx.nx), dtype=torch.float64, device=dev)
        i32 = lambda t: as_ptr(t.data_ptr(), _abi.c_int32_p)
        self.struct = _abi.McOut(infeasible=i32(self.infeasible), status=i32(self.status), qp_iters=i32(self.qp_iters),
                                 alpha_final=_f64(self.alpha_final), v_final=_f64(self.v_final), alpha0=_f64(self.alpha0),
                                 x_final=_f64(self.x_final), traj_x=None, traj_u=None, traj_alpha=None, traj_s0=None)


def mc_run(ctx, sampler, bufs, steps, card_I0, seed, x0, acc_bound, distance_error, trial0=0,
           sampler_max_attempts=1000):
    x0 = fcol(x0).ravel()
    cfg = _abi.McCfg(trials=bufs.trials, trial0=trial0, steps=steps, card_I0=card_I0, seed=seed, x0=dptr(x0),
                     acc_bound=float(acc_bound), distance_error=float(distance_error), traj_trials=0,
                     sampler_max_attempts=sampler_max_attempts)
    core._chk(ctx.lib.lbmpc_mc_run_dev(ctx.handle, sampler.ref(), C.byref(cfg), C.byref(bufs.struct), _stream()),
              "lbmpc_mc_run_dev")
    return bufs
