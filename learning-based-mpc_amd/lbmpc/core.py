"""Thin numpy front of the C ABI (include/lbmpc.h).  Every function here ends in a call into
liblbmpc_hip.so; nothing is computed on the CPU.  Arrays follow MATLAB shapes (column = sample /
direction / problem)."""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import dptr, fcol, iptr


class LbmpcError(RuntimeError):
    def __init__(self, code, where):
        lib = _abi.load()
        msg = lib.lbmpc_last_error().decode(errors="replace")
        super().__init__(f"{where}: error {code}: {msg}")
        self.code = code


def init(device=None):
    """Validate a device for the library.  device=None: the calling thread's CURRENT device, left selected as it is
    (what every wrapper below uses: under one process per GPU the launcher has already bound the device, and the
    library must never fall back to device 0 behind it).  An explicit index selects that device first."""
    lib = _abi.load()
    rc = lib.lbmpc_init(_abi.DEVICE_CURRENT if device is None else int(device))
    if rc != 0:
        raise LbmpcError(rc, "lbmpc_init")
    return lib


def _chk(rc, where):
    if rc < 0:
        raise LbmpcError(rc, where)
    return rc


def support_batch(A, Vw, s, alpha, D, transpose=True):
    """h_S(d) for the columns of D (nx x nd).  Vw: nV x nx vertices of W."""
    lib = init()
    A, Vw, D = fcol(A), fcol(Vw), fcol(D)
    nx, nd = D.shape
    h = np.zeros(nd)
    _chk(lib.lbmpc_support_batch(dptr(A), nx, dptr(Vw), Vw.shape[0], int(s), float(alpha), dptr(D), nd,
                                 int(bool(transpose)), dptr(h)), "lbmpc_support_batch")
    return h


def mrpi_s_alpha(A, Vw, WA, Wb, Dirs, eps, mode=_abi.MS_MAX_LITERAL, max_s=60):
    lib = init()
    A, Vw, WA, Wb, Dirs = fcol(A), fcol(Vw), fcol(WA), fcol(Wb).ravel(), fcol(Dirs)
    nx = A.shape[0]
    s, alpha, Ms = C.c_int(0), C.c_double(0), C.c_double(0)
    rc = _chk(lib.lbmpc_mrpi_s_alpha(dptr(A), nx, dptr(Vw), Vw.shape[0], dptr(WA), dptr(Wb), WA.shape[0], dptr(Dirs),
                                     Dirs.shape[1], float(eps), int(mode), int(max_s), C.byref(s),
                                     C.cast(C.byref(alpha), _abi.c_double_p), C.cast(C.byref(Ms), _abi.c_double_p)),
              "lbmpc_mrpi_s_alpha")
    return s.value, alpha.value, Ms.value, rc


def halfspace_max(WA, samples):
    """samples: nx x N (one set) or batch x nx x N -> m: nW (or batch x nW)"""
    lib = init()
    WA = fcol(WA)
    nW, nx = WA.shape
    S = np.asarray(samples, dtype=np.float64)
    single = S.ndim == 2
    if single:
        S = S[None]
    batch, _, N = S.shape
    flat = np.ascontiguousarray(np.transpose(S, (0, 2, 1)))  # batch x N x nx == column-major nx x N per set
    m = np.zeros((batch, nW))
    _chk(lib.lbmpc_halfspace_max(dptr(WA), nW, nx, dptr(flat), N, batch, N * nx, dptr(m)), "lbmpc_halfspace_max")
    return m[0] if single else m


def halfspace_max_multi(WA, samples, devices):
    """halfspace_max with every set sharded by sample over `devices` (list of device indices): local reductions, one
    RCCL all-reduce(max) inside the library"""
    lib = _abi.load()
    WA = fcol(WA)
    nW, nx = WA.shape
    S = np.asarray(samples, dtype=np.float64)
    single = S.ndim == 2
    if single:
        S = S[None]
    batch, _, N = S.shape
    flat = np.ascontiguousarray(np.transpose(S, (0, 2, 1)))
    m = np.zeros((batch, nW))
    dv = (C.c_int * len(devices))(*devices)
    _chk(lib.lbmpc_halfspace_max_multi(dptr(WA), nW, nx, dptr(flat), N, batch, N * nx, dptr(m), dv, len(devices)),
         "lbmpc_halfspace_max_multi")
    return m[0] if single else m


def block_shard(total, rank, world):
    lib = _abi.load()
    a, b = C.c_int64(0), C.c_int64(0)
    _chk(lib.lbmpc_block_shard(int(total), int(rank), int(world), C.byref(a), C.byref(b)), "lbmpc_block_shard")
    return a.value, b.value


def sample_information_sets(sampler, WA, N, batch, seed, set0=0, max_attempts=100, want_samples=True):
    lib = init()
    WA = fcol(WA)
    nW, nx = WA.shape
    S = np.zeros((batch, N, nx)) if want_samples else None
    m = np.zeros((batch, nW))
    st = np.zeros(batch, np.int32)
    _chk(lib.lbmpc_sample_information_sets(sampler.ref(), dptr(WA), nW, N, batch, seed, set0, max_attempts,
                                           dptr(S), dptr(m), iptr(st)), "lbmpc_sample_information_sets")
    return (np.transpose(S, (0, 2, 1)) if want_samples else None), m, st


def homothety_lp(WA, Wb, m):
    """m: nW or batch x nW -> alpha, v (nx or batch x nx), active_mask, status"""
    lib = init()
    WA, Wb = fcol(WA), fcol(Wb).ravel()
    nW, nx = WA.shape
    M = np.ascontiguousarray(np.asarray(m, dtype=np.float64))
    single = M.ndim == 1
    if single:
        M = M[None]
    batch = M.shape[0]
    a, v = np.zeros(batch), np.zeros((batch, nx))
    mask, st = np.zeros(batch, np.uint32), np.zeros(batch, np.int32)
    _chk(lib.lbmpc_homothety_lp(dptr(WA), dptr(Wb), nW, nx, dptr(M), batch, _abi.TIE_CENTER, dptr(a), dptr(v),
                                mask.ctypes.data_as(_abi.c_uint32_p), iptr(st)), "lbmpc_homothety_lp")
    if single:
        return a[0], v[0], int(mask[0]), int(st[0])
    return a, v, mask, st


def polytope_support(A, b, Cd, want_z=False):
    """h[s, i] = max {Cd[:, i]' z : A z <= b[s]}: A m x n, b m or nb x m (all > 0), Cd n x nd.
    Returns h (nd or nb x nd), status (same shape) and, with want_z, the maximisers (... x n)."""
    lib = init()
    A, Cd = fcol(A), fcol(Cd)
    m, n = A.shape
    B = np.ascontiguousarray(np.atleast_2d(np.asarray(b, dtype=np.float64)))
    single = np.ndim(b) == 1
    nb, nd = B.shape[0], Cd.shape[1]
    h, st = np.zeros((nb, nd)), np.zeros((nb, nd), np.int32)
    z = np.zeros((nb, nd, n)) if want_z else None
    rc = lib.lbmpc_polytope_support(dptr(A), m, n, dptr(B), nb, dptr(Cd), nd, dptr(h), dptr(z), iptr(st))
    _chk(rc, "lbmpc_polytope_support")
    out = (h[0], st[0]) if single else (h, st)
    if want_z:
        out = out + ((z[0] if single else z),)
    return out


def polytope_redundant(A, b):
    """slack[j] = b_j/|a_j| - max {a_j' z / |a_j| : rows != j}: >= 0 where row j is implied by the others, -inf where
    only row j bounds its direction (lbmpc_polytope_redundant; needs b > 0).  Returns (slack, status)."""
    lib = init()
    A = fcol(A)
    m, n = A.shape
    bb = np.ascontiguousarray(np.asarray(b, dtype=np.float64).ravel())
    slack, st = np.zeros(m), np.zeros(m, np.int32)
    _chk(lib.lbmpc_polytope_redundant(dptr(A), m, n, dptr(bb), dptr(slack), iptr(st)), "lbmpc_polytope_redundant")
    return slack, st


class Context:
    """Device-resident controller constants (lbmpc_uqmpc_create / lbmpc_destroy)."""

    def __init__(self, params_dict, device=None):
        """device=None: the current device.  The context remembers its device; every later call runs there."""
        self.lib = init(device)
        self.d = params_dict
        self.keep = _abi.make_params(params_dict)
        self.handle = C.c_void_p()
        rc = self.lib.lbmpc_uqmpc_create(self.keep.ref(), C.byref(self.handle))
        if rc != 0:
            raise LbmpcError(rc, "lbmpc_uqmpc_create")
        p = self.keep.struct
        self.nx, self.nu, self.N, self.nc, self.nS, self.nW = p.nx, p.nu, p.N, p.nc, p.nS, p.nW
        self.nz = p.nx + p.N * p.nu

    def close(self):
        if self.handle:
            self.lib.lbmpc_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, x, alpha_prev, v_prev, w_new=None, uq_update=True):
        """Batched UQRobustMPC.solve: x nx x B (or nx), alpha_prev B, v_prev nx x B, w_new nx x B."""
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        X = np.ascontiguousarray(x.reshape(self.nx, -1).T)  # B x nx == column-major nx x B
        B = X.shape[0]
        ap = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha_prev, dtype=np.float64).ravel(), (B,)))
        V = np.ascontiguousarray(np.asarray(v_prev, dtype=np.float64).reshape(self.nx, -1).T)
        W = None if w_new is None else np.ascontiguousarray(np.asarray(w_new, dtype=np.float64).reshape(self.nx, -1).T)
        if uq_update and W is None:
            raise ValueError("w_new is required when uq_update is set")
        o = dict(s0=np.zeros((B, self.nx)), c=np.zeros((B, self.nz - self.nx)), u=np.zeros((B, self.nu)),
                 alpha=np.zeros(B), v=np.zeros((B, self.nx)), hk=np.zeros((B, self.nc)), Sb_hat=np.zeros((B, self.nS)),
                 Wb_hat=np.zeros((B, self.nW)))
        st, na, it = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        act = np.zeros((B, _abi.MAX_NZ), np.int32)
        rc = self.lib.lbmpc_uqmpc_step(self.handle, B, int(bool(uq_update)), dptr(X), dptr(ap), dptr(V), dptr(W),
                                       dptr(o["s0"]), dptr(o["c"]), dptr(o["u"]), dptr(o["alpha"]), dptr(o["v"]),
                                       dptr(o["hk"]), dptr(o["Sb_hat"]), dptr(o["Wb_hat"]), iptr(st), iptr(na),
                                       iptr(act), iptr(it))
        _chk(rc, "lbmpc_uqmpc_step")
        o.update(status=st, n_active=na, active_set=act, qp_iters=it)
        if single:
            o = {k: (v[0] if isinstance(v, np.ndarray) else v) for k, v in o.items()}
        return o

    def mc_counters(self):
        """Work counters of the last Monte-Carlo launch (lbmpc_mc_counters)."""
        buf = (C.c_uint64 * 160)()
        _chk(self.lib.lbmpc_mc_stats(self.handle, buf, 160), "lbmpc_mc_stats")
        names = ["filter_wave_steps", "wave_steps", "qp_solves", "solver_rounds", "solver_trips", "pivot_scans", "h_passes",
                 "lane_trips", "lane_scans", "row_solver_qps", "lane_scan_qps", "gen_trips", "gen_scans", "gen_h_passes",
                 "gen_scan_qps", "gen_rebuild_trips", "gen_qps", "gen_h_chunks"]
        return {n: int(buf[i]) for i, n in enumerate(names)}

    def mc_plan(self, trials, steps):
        """launch shape + kernel resources of a Monte-Carlo launch (lbmpc_mc_plan)"""
        buf = (C.c_int32 * 11)()
        _chk(self.lib.lbmpc_mc_plan(self.handle, int(trials), int(steps), buf, 11), "lbmpc_mc_plan")
        names = ["trials_per_wave", "trials_per_lane", "waves_per_block", "lds_bytes_per_cu", "gen_solver_lanes",
                 "transient_steps", "steady_kernel", "registers_per_lane", "scratch_bytes_per_lane", "waves_per_simd", "blocks"]
        return {n: int(buf[i]) for i, n in enumerate(names)}

    def mc_histograms(self):
        """(iterations per QP solve, final active-set size per QP solve) of the last Monte-Carlo launch, 64 bins each."""
        buf = (C.c_uint64 * 160)()
        _chk(self.lib.lbmpc_mc_stats(self.handle, buf, 160), "lbmpc_mc_stats")
        return [int(buf[32 + i]) for i in range(64)], [int(buf[96 + i]) for i in range(64)]

    def mc_run(self, sampler, trials, steps, card_I0, seed, x0, acc_bound, distance_error, trial0=0, traj_trials=0,
               sampler_max_attempts=1000):
        """Batched closed-loop Monte-Carlo (Case_3/4/5 loops); returns host arrays."""
        nx, nu = self.nx, self.nu
        x0 = fcol(x0).ravel()
        R = min(traj_trials, trials)
        o = dict(infeasible=np.zeros(trials, np.int32), status=np.zeros(trials, np.int32),
                 qp_iters=np.zeros(trials, np.int32), alpha_final=np.zeros(trials), v_final=np.zeros((trials, nx)),
                 alpha0=np.zeros(trials), x_final=np.zeros((trials, nx)), traj_x=np.zeros((R, steps + 1, nx)),
                 traj_u=np.zeros((R, steps, nu)), traj_alpha=np.zeros((R, steps + 1)), traj_s0=np.zeros((R, steps, nx)))
        cfg = _abi.McCfg(trials=trials, trial0=trial0, steps=steps, card_I0=card_I0, seed=seed, x0=dptr(x0),
                         acc_bound=float(acc_bound), distance_error=float(distance_error),
                         traj_trials=R, sampler_max_attempts=sampler_max_attempts)
        out = _abi.McOut(**{k: (v.ctypes.data_as(_abi.c_int32_p) if v.dtype == np.int32 else dptr(v))
                            for k, v in o.items()})
        _chk(self.lib.lbmpc_mc_run(self.handle, sampler.ref(), C.byref(cfg), C.byref(out)), "lbmpc_mc_run")
        return o


class MultiContext:
    """One controller context per device inside this process (lbmpc_multi_create); mc_run shards the trials over them."""

    def __init__(self, params_dict, devices):
        self.lib = _abi.load()
        self.keep = _abi.make_params(params_dict)
        self.handle = C.c_void_p()
        dv = (C.c_int * len(devices))(*devices)
        rc = self.lib.lbmpc_multi_create(self.keep.ref(), dv, len(devices), C.byref(self.handle))
        if rc != 0:
            raise LbmpcError(rc, "lbmpc_multi_create")
        p = self.keep.struct
        self.nx, self.nu = p.nx, p.nu

    def close(self):
        if self.handle:
            self.lib.lbmpc_multi_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def mc_run(self, sampler, trials, steps, card_I0, seed, x0, acc_bound, distance_error, trial0=0, traj_trials=0,
               sampler_max_attempts=1000):
        nx, nu = self.nx, self.nu
        x0 = fcol(x0).ravel()
        R = min(traj_trials, trials)
        o = dict(infeasible=np.zeros(trials, np.int32), status=np.zeros(trials, np.int32),
                 qp_iters=np.zeros(trials, np.int32), alpha_final=np.zeros(trials), v_final=np.zeros((trials, nx)),
                 alpha0=np.zeros(trials), x_final=np.zeros((trials, nx)), traj_x=np.zeros((R, steps + 1, nx)),
                 traj_u=np.zeros((R, steps, nu)), traj_alpha=np.zeros((R, steps + 1)), traj_s0=np.zeros((R, steps, nx)))
        cfg = _abi.McCfg(trials=trials, trial0=trial0, steps=steps, card_I0=card_I0, seed=seed, x0=dptr(x0),
                         acc_bound=float(acc_bound), distance_error=float(distance_error),
                         traj_trials=R, sampler_max_attempts=sampler_max_attempts)
        out = _abi.McOut(**{k: (v.ctypes.data_as(_abi.c_int32_p) if v.dtype == np.int32 else dptr(v))
                            for k, v in o.items()})
        _chk(self.lib.lbmpc_multi_mc_run(self.handle, sampler.ref(), C.byref(cfg), C.byref(out)), "lbmpc_multi_mc_run")
        return o
