"""numpy front of the CPU oracle with the same call shapes as lbmpc.core (tests / cpu_baseline only)."""
import ctypes as C

import numpy as np

import oracle_ffi
from lbmpc import _abi
from lbmpc._abi import dptr, fcol, iptr


def support_batch(A, Vw, s, alpha, D, transpose=True):
    lib = oracle_ffi.load()
    A, Vw, D = fcol(A), fcol(Vw), fcol(D)
    nx, nd = D.shape
    h = np.zeros(nd)
    assert lib.orc_support_batch(dptr(A), nx, dptr(Vw), Vw.shape[0], int(s), float(alpha), dptr(D), nd,
                                 int(bool(transpose)), dptr(h)) == 0
    return h


def mrpi_s_alpha(A, Vw, WA, Wb, Dirs, eps, mode=0, max_s=60):
    lib = oracle_ffi.load()
    A, Vw, WA, Wb, Dirs = fcol(A), fcol(Vw), fcol(WA), fcol(Wb).ravel(), fcol(Dirs)
    s, alpha, Ms = C.c_int(0), C.c_double(0), C.c_double(0)
    rc = lib.orc_mrpi_s_alpha(dptr(A), A.shape[0], dptr(Vw), Vw.shape[0], dptr(WA), dptr(Wb), WA.shape[0], dptr(Dirs),
                              Dirs.shape[1], float(eps), int(mode), int(max_s), C.byref(s),
                              C.cast(C.byref(alpha), _abi.c_double_p), C.cast(C.byref(Ms), _abi.c_double_p))
    return s.value, alpha.value, Ms.value, rc


def halfspace_max(WA, samples):
    lib = oracle_ffi.load()
    WA = fcol(WA)
    nW, nx = WA.shape
    S = np.asarray(samples, dtype=np.float64)
    single = S.ndim == 2
    if single:
        S = S[None]
    batch, _, N = S.shape
    flat = np.ascontiguousarray(np.transpose(S, (0, 2, 1)))
    m = np.zeros((batch, nW))
    assert lib.orc_halfspace_max(dptr(WA), nW, nx, dptr(flat), N, batch, N * nx, dptr(m)) == 0
    return m[0] if single else m


def sample_information_sets(sampler, WA, N, batch, seed, set0=0, max_attempts=100, want_samples=True):
    lib = oracle_ffi.load()
    WA = fcol(WA)
    nW, nx = WA.shape
    S = np.zeros((batch, N, nx)) if want_samples else None
    m = np.zeros((batch, nW))
    st = np.zeros(batch, np.int32)
    assert lib.orc_sample_information_sets(sampler.ref(), dptr(WA), nW, N, batch, seed, set0, max_attempts, dptr(S),
                                           dptr(m), iptr(st)) == 0
    return (np.transpose(S, (0, 2, 1)) if want_samples else None), m, st


def homothety_lp(WA, Wb, m):
    lib = oracle_ffi.load()
    WA, Wb = fcol(WA), fcol(Wb).ravel()
    nW, nx = WA.shape
    M = np.ascontiguousarray(np.asarray(m, dtype=np.float64))
    single = M.ndim == 1
    if single:
        M = M[None]
    batch = M.shape[0]
    a, v = np.zeros(batch), np.zeros((batch, nx))
    mask, st = np.zeros(batch, np.uint32), np.zeros(batch, np.int32)
    rc = lib.orc_homothety_lp(dptr(WA), dptr(Wb), nW, nx, dptr(M), batch, _abi.TIE_CENTER, dptr(a), dptr(v),
                              mask.ctypes.data_as(_abi.c_uint32_p), iptr(st))
    assert rc == 0, rc
    if single:
        return a[0], v[0], int(mask[0]), int(st[0])
    return a, v, mask, st


def polytope_support(A, b, Cd, want_z=False):
    """h[s, i] = max {Cd[:, i]' z : A z <= b[s]}: A m x n, b m or nb x m (all > 0), Cd n x nd.
    Returns h (nd or nb x nd), status (same shape) and, with want_z, the maximisers (... x n)."""
    lib = oracle_ffi.load()
    A, Cd = fcol(A), fcol(Cd)
    m, n = A.shape
    B = np.ascontiguousarray(np.atleast_2d(np.asarray(b, dtype=np.float64)))
    single = np.ndim(b) == 1
    nb, nd = B.shape[0], Cd.shape[1]
    h, st = np.zeros((nb, nd)), np.zeros((nb, nd), np.int32)
    z = np.zeros((nb, nd, n)) if want_z else None
    rc = lib.orc_polytope_support(dptr(A), m, n, dptr(B), nb, dptr(Cd), nd, dptr(h), dptr(z), iptr(st))
    assert rc == 0, rc
    out = (h[0], st[0]) if single else (h, st)
    if want_z:
        out = out + ((z[0] if single else z),)
    return out


def polytope_redundant(A, b):
    lib = oracle_ffi.load()
    A = fcol(A)
    m, n = A.shape
    bb = np.ascontiguousarray(np.asarray(b, dtype=np.float64).ravel())
    slack, st = np.zeros(m), np.zeros(m, np.int32)
    rc = lib.orc_polytope_redundant(dptr(A), m, n, dptr(bb), dptr(slack), iptr(st))
    assert rc == 0, rc
    return slack, st


class Context:
    def __init__(self, params_dict):
        self.lib = oracle_ffi.load()
        self.keep = _abi.make_params(params_dict)
        self.handle = C.c_void_p()
        rc = self.lib.orc_uqmpc_create(self.keep.ref(), C.byref(self.handle))
        assert rc == 0, rc
        p = self.keep.struct
        self.nx, self.nu, self.N, self.nc, self.nS, self.nW = p.nx, p.nu, p.N, p.nc, p.nS, p.nW
        self.nz = p.nx + p.N * p.nu

    def close(self):
        if self.handle:
            self.lib.orc_destroy(self.handle)
            self.handle = C.c_void_p()

    def step(self, x, alpha_prev, v_prev, w_new=None, uq_update=True):
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        X = np.ascontiguousarray(x.reshape(self.nx, -1).T)
        B = X.shape[0]
        ap = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha_prev, dtype=np.float64).ravel(), (B,)))
        V = np.ascontiguousarray(np.asarray(v_prev, dtype=np.float64).reshape(self.nx, -1).T)
        W = None if w_new is None else np.ascontiguousarray(np.asarray(w_new, dtype=np.float64).reshape(self.nx, -1).T)
        o = dict(s0=np.zeros((B, self.nx)), c=np.zeros((B, self.nz - self.nx)), u=np.zeros((B, self.nu)),
                 alpha=np.zeros(B), v=np.zeros((B, self.nx)), hk=np.zeros((B, self.nc)), Sb_hat=np.zeros((B, self.nS)),
                 Wb_hat=np.zeros((B, self.nW)))
        st, na, it = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
        act = np.zeros((B, _abi.MAX_NZ), np.int32)
        rc = self.lib.orc_uqmpc_step(self.handle, B, int(bool(uq_update)), dptr(X), dptr(ap), dptr(V), dptr(W),
                                     dptr(o["s0"]), dptr(o["c"]), dptr(o["u"]), dptr(o["alpha"]), dptr(o["v"]),
                                     dptr(o["hk"]), dptr(o["Sb_hat"]), dptr(o["Wb_hat"]), iptr(st), iptr(na), iptr(act),
                                     iptr(it))
        assert rc == 0, rc
        o.update(status=st, n_active=na, active_set=act, qp_iters=it)
        if single:
            o = {k: (v[0] if isinstance(v, np.ndarray) else v) for k, v in o.items()}
        return o

    def set_certificate(self, ME=None):
        """benchmark variant only (orc_set_certificate): skip the QP when the inscribed ellipsoid {x' ME x <= 1} of S
        proves z = 0 optimal; None switches it off"""
        if ME is None:
            rc = self.lib.orc_set_certificate(self.handle, None)
        else:
            m = np.ascontiguousarray(ME, dtype=np.float64)
            assert m.shape == (self.nx, self.nx)
            rc = self.lib.orc_set_certificate(self.handle, m.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc == 0, rc

    def mc_run(self, sampler, trials, steps, card_I0, seed, x0, acc_bound, distance_error, trial0=0, traj_trials=0,
               sampler_max_attempts=1000, threads=1):
        """threads > 1: the trials are spread over that many host threads (OpenMP); same results"""
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
        if threads > 1:
            rc = self.lib.orc_mc_run_omp(self.handle, sampler.ref(), C.byref(cfg), C.byref(out), int(threads))
        else:
            rc = self.lib.orc_mc_run(self.handle, sampler.ref(), C.byref(cfg), C.byref(out))
        assert rc == 0, rc
        return o
