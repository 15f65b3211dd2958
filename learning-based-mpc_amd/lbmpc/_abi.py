"""ctypes view of include/lbmpc.h: struct layouts, prototypes and the library loader.

The product library is `learning-based-mpc_amd/csrc/liblbmpc_hip.so` (built by
`__graft_entry__.build()` / `make -C learning-based-mpc_amd/csrc`).  There is no CPU fallback: if
the library is missing, loading raises.
"""
import ctypes as C
import os

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_int32_p = C.POINTER(C.c_int32)
c_uint32_p = C.POINTER(C.c_uint32)

MAX_NX, MAX_NZ, MAX_NC, MAX_NW, MAX_NXI, MAX_NV, MAX_S = 4, 16, 16, 16, 16, 1024, 64
ERR_ARG, ERR_DIM, ERR_HIP, ERR_UNSUPPORTED, ERR_NODEVICE = -1, -2, -3, -4, -5
ST_OK, ST_QP_INFEASIBLE, ST_LP_INFEASIBLE, ST_ITER_CAP, ST_SAMPLER_CAP, ST_LP_UNBOUNDED = 0, 1, 2, 3, 4, 5
MS_MAX_LITERAL, MS_SUM_LITERAL, MS_RAKOVIC = 0, 1, 2
TIE_CENTER = 1
DEVICE_CURRENT = -1
NOALLOC_MAX_PARTIALS = 262144


class Sampler(C.Structure):
    _fields_ = [("nx", C.c_int), ("nu", C.c_int), ("nEV", C.c_int), ("nLV", C.c_int),
                ("EV_A", c_double_p), ("EV_b", c_double_p), ("EV_lo", c_double_p), ("EV_hi", c_double_p),
                ("LV_A", c_double_p), ("LV_b", c_double_p), ("LV_lo", c_double_p), ("LV_hi", c_double_p),
                ("B", c_double_p), ("min_u_LV", c_double_p), ("max_u_LV", c_double_p),
                ("EV_U", c_double_p), ("LV_U", c_double_p)]


class Params(C.Structure):
    _fields_ = [("nx", C.c_int), ("nu", C.c_int), ("N", C.c_int), ("nc", C.c_int), ("Nu", C.c_int),
                ("nS", C.c_int), ("nW", C.c_int)] + [
        (k, c_double_p) for k in ["A", "B", "K", "Phi", "Px", "Pc", "F", "G", "F_bar", "Psi", "S_A", "S_b",
                                  "W_A", "W_b", "hs"]]


class McCfg(C.Structure):
    _fields_ = [("trials", C.c_int64), ("trial0", C.c_int64), ("steps", C.c_int32), ("card_I0", C.c_int32),
                ("seed", C.c_uint64), ("x0", c_double_p), ("acc_bound", C.c_double),
                ("distance_error", C.c_double), ("traj_trials", C.c_int32),
                ("sampler_max_attempts", C.c_int32)]


class McOut(C.Structure):
    _fields_ = [("infeasible", c_int32_p), ("status", c_int32_p), ("qp_iters", c_int32_p),
                ("alpha_final", c_double_p), ("v_final", c_double_p), ("alpha0", c_double_p),
                ("x_final", c_double_p), ("traj_x", c_double_p), ("traj_u", c_double_p),
                ("traj_alpha", c_double_p), ("traj_s0", c_double_p)]


class ScmpcParams(C.Structure):
    _fields_ = [("nx", C.c_int), ("nu", C.c_int), ("N", C.c_int), ("nc", C.c_int), ("N_SC", C.c_int)] + [
        (k, c_double_p) for k in ["A", "B", "Q", "R", "F", "G"]]


class ScmpcOut(C.Structure):
    _fields_ = [("infeasible", c_int32_p), ("status", c_int32_p), ("qp_iters", c_int32_p), ("x_final", c_double_p),
                ("traj_x", c_double_p), ("traj_u", c_double_p)]


def fcol(a):
    """fp64 column-major (MATLAB layout) copy of `a`."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def dptr(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def iptr(a):
    return None if a is None else a.ctypes.data_as(c_int_p)


def as_ptr(addr, typ=c_double_p):
    """integer device address (e.g. torch.Tensor.data_ptr()) -> typed pointer"""
    return C.cast(C.c_void_p(addr), typ)


class Keep:
    """Bundle a ctypes struct with the numpy arrays its pointers refer to (keeps them alive)."""

    def __init__(self, struct, arrays):
        self.struct, self.arrays = struct, arrays

    def ref(self):
        return C.byref(self.struct)


def oriented_box(V):
    """Proposal box of a polytope with vertices V (rows): the principal axes of the vertex cloud when that box is
    smaller than the axis-aligned one (a rotated cuboid is then sampled without rejections), else None.
    Returns (U, lo, hi): candidates are x = U t, t uniform in [lo, hi]."""
    V = np.asarray(V, dtype=np.float64)
    lo0, hi0 = V.min(0), V.max(0)
    c = V.mean(0)
    _, U = np.linalg.eigh((V - c).T @ (V - c))
    T = V @ U                       # coordinates t = U' x (U orthonormal)
    lo, hi = T.min(0), T.max(0)
    if np.prod(hi - lo) < 0.9 * np.prod(hi0 - lo0):
        return np.ascontiguousarray(U), lo, hi
    return None, lo0, hi0


def make_sampler(nx, nu, EV_A, EV_b, EV_V, LV_A, LV_b, LV_V, B, min_u, max_u, oriented=False):
    """oriented=False: the reference's axis-aligned bounding box of the vertices (ModelingCar.m:50-51);
    oriented=True: principal-axes box where it is tighter (same distribution, fewer rejections)."""
    boxes = {}
    for name, V in (("EV", EV_V), ("LV", LV_V)):
        boxes[name] = oriented_box(V) if oriented else (None, np.min(V, 0), np.max(V, 0))
    arr = dict(EV_A=fcol(EV_A), EV_b=fcol(EV_b).ravel(), EV_lo=fcol(boxes["EV"][1]), EV_hi=fcol(boxes["EV"][2]),
               LV_A=fcol(LV_A), LV_b=fcol(LV_b).ravel(), LV_lo=fcol(boxes["LV"][1]), LV_hi=fcol(boxes["LV"][2]),
               B=fcol(np.reshape(B, (nx, nu))), min_u_LV=fcol(np.ravel(min_u)), max_u_LV=fcol(np.ravel(max_u)))
    for name in ("EV", "LV"):
        if boxes[name][0] is not None:
            arr[name + "_U"] = fcol(boxes[name][0])
    s = Sampler(nx=nx, nu=nu, nEV=arr["EV_A"].shape[0], nLV=arr["LV_A"].shape[0],
                **{k: dptr(v) for k, v in arr.items()})
    return Keep(s, arr)


def make_params(d):
    """d: mapping with the fields of `opts_UQMPC` (UQRobustMPC.m:39-62), polytopes as S_A,S_b,W_A,W_b."""
    nx, nu, N, nc = int(d["nx"]), int(d["nu"]), int(d["N"]), int(d["nc"])
    nz = nx + N * nu
    arr = dict(A=fcol(d["A"]).reshape(nx, nx, order="F"), B=fcol(np.reshape(d["B"], (nx, nu))),
               K=fcol(np.reshape(d["K"], (nu, nx))), Phi=fcol(d["Phi"]), Px=fcol(d["Px"]), Pc=fcol(d["Pc"]),
               F=fcol(np.reshape(d["F"], (nc, nx))), G=fcol(np.reshape(d["G"], (nc, nu))),
               F_bar=fcol(np.reshape(d["F_bar"], (nc, nz))), Psi=fcol(d["Psi"]), S_A=fcol(d["S_A"]),
               S_b=fcol(d["S_b"]).ravel(), W_A=fcol(d["W_A"]), W_b=fcol(d["W_b"]).ravel(), hs=fcol(d["hs"]).ravel())
    p = Params(nx=nx, nu=nu, N=N, nc=nc, Nu=int(d["Nu"]), nS=arr["S_A"].shape[0], nW=arr["W_A"].shape[0],
               **{k: dptr(v) for k, v in arr.items()})
    return Keep(p, arr)


_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("LBMPC_LIB") or os.path.join(_PKG_DIR, "csrc", "liblbmpc_hip.so")  # LBMPC_LIB: diagnostic builds only


def declare(lib, prefix):
    """Attach argtypes/restype for the symbols of include/lbmpc.h (prefix 'lbmpc_' for the product;
    tests reuse this for the oracle's 'orc_' twins, which share the signatures of the host entry points)."""
    i64, u64, ci, cd, vp = C.c_int64, C.c_uint64, C.c_int, C.c_double, C.c_void_p
    dp, ip, up = c_double_p, c_int_p, c_uint32_p
    sig = {
        "support_batch": [dp, ci, dp, ci, ci, cd, dp, i64, ci, dp],
        "mrpi_s_alpha": [dp, ci, dp, ci, dp, dp, ci, dp, ci, cd, ci, ci, ip, dp, dp],
        "halfspace_max": [dp, ci, ci, dp, i64, i64, i64, dp],
        "sample_information_sets": [C.POINTER(Sampler), dp, ci, i64, i64, u64, u64, ci, dp, dp, ip],
        "homothety_lp": [dp, dp, ci, ci, dp, i64, ci, dp, dp, up, ip],
        "sample_polytope": [dp, dp, ci, ci, dp, dp, i64, u64, u64, C.c_uint32, ci, ci, dp, ip],
        "polytope_support": [dp, ci, ci, dp, i64, dp, i64, dp, dp, ip],
        "polytope_redundant": [dp, ci, ci, dp, dp, ip],
        "uqmpc_create": [C.POINTER(Params), C.POINTER(vp)],
        "uqmpc_step": [vp, i64, ci, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, ip, ip, ip, ip],
        "mc_run": [vp, C.POINTER(Sampler), C.POINTER(McCfg), C.POINTER(McOut)],
        "scmpc_mc_run": [C.POINTER(ScmpcParams), C.POINTER(Sampler), C.POINTER(McCfg), C.POINTER(ScmpcOut)],
    }
    for name, args in sig.items():
        f = getattr(lib, prefix + name)
        f.argtypes, f.restype = args, ci
    d = getattr(lib, prefix + "destroy")
    d.argtypes, d.restype = [vp], None
    if prefix == "lbmpc_":
        for name in ["support_batch", "halfspace_max", "sample_information_sets", "homothety_lp", "mc_run"]:
            f = getattr(lib, prefix + name + "_dev")
            f.argtypes, f.restype = sig[name] + [vp], ci
        lib.lbmpc_multi_create.argtypes, lib.lbmpc_multi_create.restype = [C.POINTER(Params), ip, ci, C.POINTER(vp)], ci
        lib.lbmpc_multi_destroy.argtypes, lib.lbmpc_multi_destroy.restype = [vp], None
        lib.lbmpc_multi_ngpu.argtypes, lib.lbmpc_multi_ngpu.restype = [vp], ci
        lib.lbmpc_multi_mc_run.argtypes, lib.lbmpc_multi_mc_run.restype = sig["mc_run"], ci
        lib.lbmpc_halfspace_max_multi.argtypes, lib.lbmpc_halfspace_max_multi.restype = sig["halfspace_max"] + [ip, ci], ci
        lib.lbmpc_block_shard.argtypes = [i64, ci, ci, C.POINTER(i64), C.POINTER(i64)]
        lib.lbmpc_block_shard.restype = ci
        lib.lbmpc_halfspace_max_noalloc.argtypes, lib.lbmpc_halfspace_max_noalloc.restype = sig["halfspace_max"] + [vp], ci
        lib.lbmpc_mc_counters.argtypes, lib.lbmpc_mc_counters.restype = [vp, C.POINTER(C.c_uint64)], ci
        lib.lbmpc_mc_stats.argtypes, lib.lbmpc_mc_stats.restype = [vp, C.POINTER(C.c_uint64), ci], ci
        lib.lbmpc_mc_plan.argtypes, lib.lbmpc_mc_plan.restype = [vp, C.c_int64, ci, C.POINTER(C.c_int32), ci], ci
        lib.lbmpc_version.restype = ci
        lib.lbmpc_init.argtypes, lib.lbmpc_init.restype = [ci], ci
        lib.lbmpc_last_error.restype = C.c_char_p
    return lib


_lib = None


def load():
    """Load liblbmpc_hip.so (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = declare(C.CDLL(LIB_PATH), "lbmpc_")
    return _lib


def make_scmpc_params(nx, nu, N, nc, N_SC, A, B, Q, R, F, G):
    arr = dict(A=fcol(np.reshape(A, (nx, nx))), B=fcol(np.reshape(B, (nx, nu))), Q=fcol(np.reshape(Q, (nx, nx))),
               R=fcol(np.reshape(R, (nu, nu))), F=fcol(np.reshape(F, (nc, nx))), G=fcol(np.reshape(G, (nc, nu))))
    return Keep(ScmpcParams(nx=nx, nu=nu, N=N, nc=nc, N_SC=N_SC, **{k: dptr(v) for k, v in arr.items()}), arr)


def scmpc_mc_run(lib, prefix, keep, sampler, trials, steps, seed, x0, acc_bound, distance_error, trial0=0, traj_trials=0,
                 sampler_max_attempts=1000):
    """lbmpc_scmpc_mc_run / its oracle twin -> dict of host arrays"""
    p = keep.struct
    nx, nu = p.nx, p.nu
    x0 = fcol(x0).ravel()
    R = min(traj_trials, trials)
    o = dict(infeasible=np.zeros(trials, np.int32), status=np.zeros(trials, np.int32), qp_iters=np.zeros(trials, np.int32),
             x_final=np.zeros((trials, nx)), traj_x=np.zeros((R, steps + 1, nx)), traj_u=np.zeros((R, steps, nu)))
    cfg = McCfg(trials=trials, trial0=trial0, steps=steps, card_I0=1, seed=seed, x0=dptr(x0), acc_bound=float(acc_bound),
                distance_error=float(distance_error), traj_trials=R, sampler_max_attempts=sampler_max_attempts)
    out = ScmpcOut(**{k: (v.ctypes.data_as(c_int32_p) if v.dtype == np.int32 else dptr(v)) for k, v in o.items()})
    rc = getattr(lib, prefix + "scmpc_mc_run")(keep.ref(), sampler.ref(), C.byref(cfg), C.byref(out))
    return rc, o
