"""Host-side mirror of the reference's MATLAB entry points for the hot path (same names, argument meaning and
outputs), every numeric step done by liblbmpc_hip.so through the C ABI:

    Fs = MRPISet(Ak, W, epsilon)                                   Functions_general/MRPISet.m:5
    IniSet = InitialSetComputation(opts); IniSet.solve(); IniSet.solve_opt(samples)   InitialSetComputation.m:28,48,61
    mpc = UQRobustMPC(opts); mpc.solve(x_k, alpha_before, v_before, w_new)            UQRobustMPC.m:39,68
    rmpc = NominalRobustMPC(opts); rmpc.solve(x_k)                                    NominalRobustMPC.m:28,47
    Car = ModelingCar(opts); Car.EVModeling(x, u); Car.LVModeling(x)                  ModelingCar.m:21,34,40
    monte_carlo(...)                                               loops of Cases_general/Case_{3,4,5}

`opts` are dicts with the reference's field names; MPT3 Polyhedra are `Polyhedron(A=, b=, V=)` objects here.
The reference is MATLAB (no toolchain in the build image): this Python layer is the stand-in for the thin `.m`
wrappers + MEX gateway of learning-based-mpc_amd/matlab/ (INTEGRATION.md).
"""
import itertools

import numpy as np

from . import _abi, core
from ._abi import dptr, fcol, iptr


class Polyhedron:
    """The subset of MPT3's Polyhedron the hot path touches: H-rep (.A, .b), V-rep (.V), scalar * P, P + vector."""

    def __init__(self, V=None, A=None, b=None):
        self.A = None if A is None else np.asarray(A, dtype=np.float64)
        self.b = None if b is None else np.asarray(b, dtype=np.float64).ravel()
        self.V = None if V is None else np.asarray(V, dtype=np.float64)
        if self.A is None and self.V is None:
            raise ValueError("Polyhedron needs V or (A, b)")

    @property
    def Dim(self):
        return (self.A if self.A is not None else self.V).shape[1]

    def __rmul__(self, s):
        s = float(s)
        return Polyhedron(V=None if self.V is None else s * self.V, A=self.A, b=None if self.b is None else s * self.b)

    def __add__(self, t):
        t = np.asarray(t, dtype=np.float64).ravel()
        return Polyhedron(V=None if self.V is None else self.V + t, A=self.A,
                          b=None if self.b is None else self.b + self.A @ t)

    __radd__ = __add__

    def box_vertices(self):
        """vertices of an axis-aligned box given in H-rep (what W is in every reference configuration)"""
        nx = self.A.shape[1]
        lo, hi = np.zeros(nx), np.zeros(nx)
        for a, b in zip(self.A, self.b):
            k = int(np.abs(a).argmax())
            if a[k] > 0:
                hi[k] = b / a[k]
            else:
                lo[k] = b / a[k]
        return np.array(list(itertools.product(*zip(lo, hi))))

    def vertices(self):
        return self.V if self.V is not None else self.box_vertices()

    def volume(self):
        """box volume (Case_2...:25 / Case_4...:63 call W_hat.volume() on homothets of the box W)"""
        V = self.vertices()
        return float(np.prod(V.max(0) - V.min(0)))


def reference_directions(nx, rng):
    """direction set of MRPISet.m:14-25: [+-I; +-1/sqrt(nx) (nx>2); 10 random unit directions]"""
    E = np.eye(nx)
    D = [E, -E]
    if nx > 2:
        D += [np.ones((1, nx)) / np.sqrt(nx), -np.ones((1, nx)) / np.sqrt(nx)]
    extra = rng.normal(size=(10, nx))
    D.append(extra / np.linalg.norm(extra, axis=1, keepdims=True))
    return np.vstack(D)


def MRPISet(Ak, W, epsilon, template_directions=None, mode=_abi.MS_RAKOVIC, seed=0, return_info=False):
    """epsilon-outer mRPI set of x+ = Ak x + w, w in W.  (s, alpha) by the loop of MRPISet.m:27-63 on the device;
    the set itself is returned as the H-rep template polytope {x : d_j' x <= h_S(d_j)} of the exact Minkowski-sum
    support (1-alpha)^-1 sum_{t<s} h_W((Ak^t)' d), evaluated by the batched support kernel, instead of the
    reference's hull of 200*nx SOCP-projected random samples (MRPISet.m:66-136; SURVEY.md hard part 4).
    mode: the default is the Rakovic form (alpha from h_W((Ak^s)' f_j), Ms from the transposed sums): it is the one
    for which (1-alpha)^-1 (+)_{t<s} Ak^t W is an epsilon-outer approximation, and the one consistent with the
    transposed sum evaluated below.  The reference's literal loops (MS_MAX_LITERAL: MRPISet.m:29,43 use A^s f_j and
    A^i d without the transpose; MS_SUM_LITERAL: Functions/MRPISet.m:8-17) remain selectable, e.g. to reproduce the
    (s, alpha) behind a stored parameter set.  Template directions are drawn from a seeded generator."""
    Ak = np.asarray(Ak, dtype=np.float64)
    nx = Ak.shape[0]
    rng = np.random.default_rng(seed)
    WV = W.vertices()
    s, alpha, Ms, rc = core.mrpi_s_alpha(Ak, WV, W.A, W.b, reference_directions(nx, rng).T, epsilon, mode=mode)
    if rc != 0 or not (0.0 <= alpha < 1.0):
        # the reference loops for ever in this case (MRPISet.m:27 has no cap); a set built from an (s, alpha) that
        # still violates the loop condition is not an outer approximation of anything
        raise RuntimeError(f"MRPISet: the (s, alpha) search stopped with status {rc} at s={s}, alpha={alpha:g} "
                           f"(Ak not contractive enough for epsilon={epsilon}?)")
    if template_directions is None:
        n = {1: 2, 2: 128, 3: 795, 4: 1644}.get(nx, 1644)
        D = rng.normal(size=(n, nx))
        template_directions = np.vstack([np.eye(nx), -np.eye(nx), D / np.linalg.norm(D, axis=1, keepdims=True)])
    D = np.asarray(template_directions, dtype=np.float64)
    h = core.support_batch(Ak, WV, s, alpha, D.T, transpose=True)
    Fs = Polyhedron(A=D, b=h)
    return (Fs, dict(s=s, alpha=alpha, Ms=Ms, status=rc)) if return_info else Fs


class InitialSetComputation:
    """fields: B, N_pre_sam, N_sam_opt, nx, W, min_u_LV, max_u_LV, Xi_true_EV, Xi_true_LV (InitialSetComputation.m:28-43)"""
    _calls = itertools.count()

    def __init__(self, parameters_NEW, seed=0):
        p = parameters_NEW
        self.B = np.atleast_2d(np.asarray(p["B"], dtype=np.float64))
        self.nx = int(p["nx"])
        if self.B.shape[0] != self.nx:
            self.B = self.B.T
        self.N_pre_sam, self.N_sam_opt = int(p["N_pre_sam"]), int(p.get("N_sam_opt", 0))
        self.W, self.W_A, self.W_b = p["W"], p["W"].A, p["W"].b
        self.min_u_LV, self.max_u_LV = p["min_u_LV"], p["max_u_LV"]
        self.Xi_true_EV, self.Xi_true_LV = p["Xi_true_EV"], p["Xi_true_LV"]
        self.seed = seed
        self.sampler = _abi.make_sampler(self.nx, self.B.shape[1], self.Xi_true_EV.A, self.Xi_true_EV.b,
                                         self.Xi_true_EV.V, self.Xi_true_LV.A, self.Xi_true_LV.b, self.Xi_true_LV.V,
                                         self.B, self.min_u_LV, self.max_u_LV)

    def solve(self):
        """[alpha_ini, v_ini, samplesinput] = obj.solve()  (:48-59); each call draws a fresh information set"""
        set_id = next(InitialSetComputation._calls)
        S, m, st = core.sample_information_sets(self.sampler, self.W_A, self.N_pre_sam, 1, self.seed, set0=set_id)
        alpha, v, _, status = core.homothety_lp(self.W_A, self.W_b, m[0])
        self.status = max(int(st[0]), status)
        return alpha, v.reshape(-1, 1), S[0]

    def solve_opt(self, samples_opt):
        """[alpha_opt, v_opt] = obj.solve_opt(samples_opt)  (:61-67), samples nx x N"""
        m = core.halfspace_max(self.W_A, np.asarray(samples_opt, dtype=np.float64))
        alpha, v, _, self.status = core.homothety_lp(self.W_A, self.W_b, m)
        return alpha, v.reshape(-1, 1)


def _params_from_opts(o):
    d = {k: o[k] for k in ("N", "nx", "nu", "nc", "Nu", "A", "B", "F", "G", "K", "Px", "Pc", "F_bar", "Psi", "Phi", "hs")
         if k in o}
    if "A" not in d:   # NominalRobustMPC's struct has no A/B/Phi: only the QP fields matter there
        nx, nu = int(o["nx"]), int(o["nu"])
        d["A"], d["B"] = np.zeros((nx, nx)), np.zeros((nx, nu))
    if "Phi" not in d:
        d["Phi"] = np.asarray(o["Psi"])[: int(o["nx"]), : int(o["nx"])]
    for k in ("F", "G"):
        if k not in d:
            d[k] = np.zeros((int(o["nc"]), int(o["nx"] if k == "F" else o["nu"])))
    d["S_A"], d["S_b"] = o["S"].A, o["S"].b
    if "W" in o:
        d["W_A"], d["W_b"] = o["W"].A, o["W"].b
    else:
        nx = int(o["nx"])
        d["W_A"], d["W_b"] = np.vstack([np.eye(nx), -np.eye(nx)]), np.ones(2 * nx)
    return d


class UQRobustMPC:
    """fields: N,nx,nu,nc,F,G,K,Px,Pc,F_bar,Psi,Phi,E,W,S,Nu,hs,Ps,num_half_space_S (UQRobustMPC.m:39-62)"""

    def __init__(self, parameters):
        self.opts = parameters
        self.W, self.S = parameters["W"], parameters["S"]
        self.nx, self.nu, self.nc = int(parameters["nx"]), int(parameters["nu"]), int(parameters["nc"])
        self.ctx = core.Context(_params_from_opts(parameters))

    def solve(self, x_k, alpha_before, v_before, w_new):
        """[s_k_opt, u_k, alpha_k, v_k, hk, S_hat_star, W_hat] = obj.solve(x_k, alpha_before, v_before, w_new)  (:68-91)"""
        o = self.ctx.step(np.ravel(x_k), alpha_before, np.ravel(v_before), np.ravel(w_new), uq_update=True)
        self.status, self.active_set = int(o["status"]), o["active_set"][: int(o["n_active"])]
        S_hat_star = Polyhedron(A=self.S.A, b=o["Sb_hat"])
        W_hat = Polyhedron(A=self.W.A, b=o["Wb_hat"])
        col = lambda a: np.asarray(a).reshape(-1, 1)
        return col(o["s0"]), col(o["u"]), float(o["alpha"]), col(o["v"]), col(o["hk"]), S_hat_star, W_hat


class NominalRobustMPC:
    """fields: N,nx,nu,nc,Nu,Px,Pc,K,F_bar,Psi,E,S,hs (NominalRobustMPC.m:28-45)"""

    def __init__(self, parameters):
        self.nx = int(parameters["nx"])
        self.ctx = core.Context(_params_from_opts(parameters))

    def solve(self, x_k):
        """[s_k_opt, u_k] = obj.solve(x_k)  (:47-52): the tube QP with the fixed S and hs"""
        o = self.ctx.step(np.ravel(x_k), 1.0, np.zeros(self.nx), None, uq_update=False)
        self.status = int(o["status"])
        return np.asarray(o["s0"]).reshape(-1, 1), np.asarray(o["u"]).reshape(-1, 1)


class ModelingCar:
    """fields: A, B, min_u_LV, max_u_LV, Xi_true_EV, Xi_true_LV (ModelingCar.m:21-32).  Every call consumes the next
    draw of this object's Philox stream (the reference consumes MATLAB's global rand)."""

    def __init__(self, parameters, seed=0, stream=0):
        p = parameters
        self.A = np.asarray(p["A"], dtype=np.float64)
        self.B = np.asarray(p["B"], dtype=np.float64).reshape(self.A.shape[0], -1)
        self.Xi_true_EV, self.Xi_true_LV = p["Xi_true_EV"], p["Xi_true_LV"]
        self.min_u_LV, self.max_u_LV = float(np.ravel(p["min_u_LV"])[0]), float(np.ravel(p["max_u_LV"])[0])
        self.seed, self.stream, self._k = seed, stream, 0

    def _sample(self, P, purpose):
        lib = core.init()
        A, b, V = fcol(P.A), fcol(P.b).ravel(), P.V
        lo, hi = fcol(V.min(0)), fcol(V.max(0))
        nx = A.shape[1]
        out, st = np.zeros(nx), np.zeros(1, np.int32)
        core._chk(lib.lbmpc_sample_polytope(dptr(A), dptr(b), A.shape[0], nx, dptr(lo), dptr(hi), 1, self.seed,
                                            self.stream, self._k, purpose, 1000, dptr(out), iptr(st)),
                  "lbmpc_sample_polytope")
        return out.reshape(-1, 1)

    def EVModeling(self, x_EV_k, u_EV_k):
        """[xi_EV_k, x_EV_k_next] = EVModeling(x, u)  (:34-38)"""
        xi = self._sample(self.Xi_true_EV, 5)
        self._k += 1
        return xi, self.A @ np.reshape(x_EV_k, (-1, 1)) + self.B @ np.reshape(u_EV_k, (-1, 1)) + xi

    def LVModeling(self, x_LV_k):
        """[xi_LV_k, x_LV_k_next] = LVModeling(x)  (:40-45)"""
        xi = self._sample(self.Xi_true_LV, 3)
        # u_LV: one uniform from the 1-row "polytope" [min_u, max_u] keeps the draw on the device stream
        u = self._sample(Polyhedron(A=np.array([[1.0], [-1.0]]), b=np.array([self.max_u_LV, -self.min_u_LV]),
                                    V=np.array([[self.min_u_LV], [self.max_u_LV]])), 4)
        return xi, self.A @ np.reshape(x_LV_k, (-1, 1)) + xi + self.B * float(u[0, 0])


def monte_carlo(opts_UQMPC, opts_Car, N_pre_sam, N_MC, K_N, x_RM_0, acc_bound, distance_error, seed=0, trial0=0,
                record_trials=0):
    """The Monte-Carlo loops of Cases_general/Case_3...:57-88, Case_4...:32-66, Case_5...:55-98 as ONE batched launch:
    per trial, InitialSetComputation.solve() with N_pre_sam samples, then K_N closed-loop UQ-RMPC steps against the
    sampled plant.  Returns dict(Infeasible_Index, alpha_K, v_K, W_Hat_Vol_final, State_RM, Control_EV, ...)."""
    mpc = opts_UQMPC if isinstance(opts_UQMPC, UQRobustMPC) else UQRobustMPC(opts_UQMPC)
    nx, nu = mpc.nx, mpc.nu
    B = np.asarray(opts_Car["B"], dtype=np.float64).reshape(nx, nu)
    smp = _abi.make_sampler(nx, nu, opts_Car["Xi_true_EV"].A, opts_Car["Xi_true_EV"].b, opts_Car["Xi_true_EV"].V,
                            opts_Car["Xi_true_LV"].A, opts_Car["Xi_true_LV"].b, opts_Car["Xi_true_LV"].V, B,
                            opts_Car["min_u_LV"], opts_Car["max_u_LV"])
    o = mpc.ctx.mc_run(smp, trials=N_MC, steps=K_N, card_I0=N_pre_sam, seed=seed, x0=np.ravel(x_RM_0),
                       acc_bound=acc_bound, distance_error=distance_error, trial0=trial0, traj_trials=record_trials)
    volW = mpc.W.volume()
    return dict(Infeasible_Index=o["infeasible"], status=o["status"], qp_iters=o["qp_iters"], alpha_ini=o["alpha0"],
                alpha_K=o["alpha_final"], v_K=o["v_final"], W_Hat_Vol_final=o["alpha_final"] ** nx * volW,
                State_RM=np.transpose(o["traj_x"], (0, 2, 1)), Control_EV=np.transpose(o["traj_u"], (0, 2, 1)),
                Nominal_RM_State=np.transpose(o["traj_s0"], (0, 2, 1)), Alpha=o["traj_alpha"],
                W_Hat_Vol=o["traj_alpha"] ** nx * volW)


def template_directions(nx, n=None, seed=0):
    """unit directions of an H-rep template in R^nx: +-I first (so that coordinate extremes are exact), then evenly
    spaced angles (nx = 2) or seeded random unit vectors"""
    E = np.vstack([np.eye(nx), -np.eye(nx)])
    if nx == 1:
        return E
    if nx == 2:
        n = 360 if n is None else n
        th = 2 * np.pi * (np.arange(n) + 0.5) / n
        return np.vstack([E, np.column_stack([np.cos(th), np.sin(th)])])
    n = {3: 795, 4: 1644}.get(nx, 1644) if n is None else n
    D = np.random.default_rng(seed).normal(size=(n, nx))
    return np.vstack([E, D / np.linalg.norm(D, axis=1, keepdims=True)])


class ComputeFeasibleRegion:
    """Functions_general/ComputeFeasibleRegion.m with the same constructor fields (S, S_true, S_hat_opt, F, G, K, F_bar,
    Psi, nc, nx, nu, N, Ps, num_half_space_S) and methods.  Every LP the reference hands to IPOPT (Com_hs, Com_z,
    Com_hs_ini) is one call of lbmpc_polytope_support; the feasible region F_N = F_Ns (+) S is returned as the H-rep
    template polytope {x : d_j'x <= h_{F_Ns}(d_j) + h_S(d_j)} of its EXACT support instead of the hull of 5000 sampled
    YALMIP projections (:153-203) -- `.V` holds one boundary point per template direction (maximiser of F_Ns plus
    maximiser of S), so callers that take min / max over `.V` columns (Case_5...:45-47) get the exact coordinate
    extremes."""

    def __init__(self, parameters, directions=None, backend=None):
        p = parameters
        self.S, self.S_true, self.S_hat_opt = p["S"], p.get("S_true"), p.get("S_hat_opt")
        self.nc, self.nx, self.nu, self.N = (int(np.ravel(p[k])[0]) for k in ("nc", "nx", "nu", "N"))
        nz = self.nx + self.N * self.nu
        self.F = np.asarray(p["F"], dtype=np.float64).reshape(self.nc, self.nx)
        self.G = np.asarray(p["G"], dtype=np.float64).reshape(self.nc, self.nu)
        self.K = np.asarray(p["K"], dtype=np.float64).reshape(self.nu, self.nx)
        self.F_bar = np.asarray(p["F_bar"], dtype=np.float64).reshape(self.nc, nz)
        self.Psi = np.asarray(p["Psi"], dtype=np.float64)
        self.Ps = p.get("Ps")
        self.F_Com = self.F + self.G @ self.K
        self.D = template_directions(self.nx) if directions is None else np.asarray(directions, dtype=np.float64)
        self._lp = core.polytope_support if backend is None else backend   # tests pass the oracle's twin

    # ---- the LPs (ComputeFeasibleRegion.m:447-519) ----
    def _hs(self, S, rows):
        h, st = self._lp(S.A, S.b, np.asarray(rows, dtype=np.float64).T)
        if st.max() != 0:
            raise RuntimeError(f"support LP over the tube set failed with status {int(st.max())}")
        return h

    def _H(self, Nu):
        M, rows = self.F_bar.copy(), []
        for _ in range(Nu + 1):
            rows.append(M)
            M = M @ self.Psi
        return np.vstack(rows)

    def ComNu_RMPC(self, hs, literal=False, max_Nu=100):
        """smallest Nu >= N with max {F_bar Psi^(Nu+1) z : z in Omega} <= 1 - hs  (:66-116).
        literal=False: Omega = {F_bar Psi^i z <= 1 - hs, i = 0..Nu}, the set the condition is about.
        literal=True: the reference's LP as written (Com_z gets M_before = F_bar Psi^Nu only, :76,:514).  In exact
        arithmetic that LP does not depend on Nu at all (substitute eta = Phi^(Nu-N) xi), so its exit test either
        holds at Nu = N or never; the Nu = 19 stored in parameters_2x2.mat comes from IPOPT's absolute tolerances on
        rows that shrink like rho(Phi)^Nu and cannot be reproduced by an exact solver.  Since every row beyond the
        exact nu_s is redundant, Omega(1 - hs, 19) = Omega(1 - hs, nu_s): the stored value is valid, only larger."""
        hs = np.ravel(hs)
        Nu = self.N
        while Nu < max_Nu:
            M = self.F_bar @ np.linalg.matrix_power(self.Psi, Nu + 1)
            A = self.F_bar @ np.linalg.matrix_power(self.Psi, Nu) if literal else self._H(Nu)
            b = (1 - hs) if literal else np.tile(1 - hs, Nu + 1)
            goal, st = self._lp(A, b, M.T)
            if st.max() == 0 and np.all(goal <= 1 - hs):
                break
            Nu += 1
        return Nu

    def Com_Nuk(self, hk, hs, Nu):
        """:521-535 (Alg. 1) with the intended exponent in the loop (the reference's `obj.Nu_k` is a typo that would
        raise as soon as the loop body runs)"""
        hk, hs = np.ravel(hk), np.ravel(hs)
        zeta = np.max((1 - hk) / (1 - hs))
        Ps = np.ones((self.Psi.shape[0],) * 2) if self.Ps is None else np.asarray(self.Ps, dtype=np.float64)
        Pi = np.linalg.pinv(Ps)
        Nu_k = Nu
        for _ in range(200):
            Pk = np.linalg.matrix_power(self.Psi, Nu_k + 1)
            vec = [self.F_bar[i] @ Pk @ Pi @ Pk.T @ self.F_bar[i] - (1 - hk[i]) / zeta ** 2 for i in range(self.nc)]
            if max(vec) <= 0:
                break
            Nu_k += 1
        return Nu_k

    def _region(self, S, h, Nu):
        """F_Ns = proj_s Omega(1 - h, Nu) and F_N = F_Ns (+) S as template polytopes over self.D"""
        nz = self.Psi.shape[0]
        Cd = np.zeros((nz, self.D.shape[0]))
        Cd[: self.nx] = self.D.T
        hF, st, z = self._lp(self._H(Nu), np.tile(1 - np.ravel(h), Nu + 1), Cd, want_z=True)
        hS, st2, e = self._lp(S.A, S.b, self.D.T, want_z=True)
        if st.max() != 0 or st2.max() != 0:
            raise RuntimeError(f"feasible-region LP failed with status {int(max(st.max(), st2.max()))}")
        F_Ns = Polyhedron(A=self.D, b=hF, V=z[:, : self.nx])
        F_N = Polyhedron(A=self.D, b=hF + hS, V=z[:, : self.nx] + e)
        return F_Ns, F_N

    def ComFeasibleRegion_RMPC(self):
        """[F_N, hs, Nu] = ComFeasibleRegion_RMPC()  (:118-203)"""
        hs = self._hs(self.S, self.F_Com)
        Nu = self.ComNu_RMPC(hs)
        return self._region(self.S, hs, Nu)[1], hs.reshape(-1, 1), Nu

    def ComFeasibleRegion_True(self):
        """[F_N, hs_true] = ComFeasibleRegion_True()  (:257-327)"""
        hs = self._hs(self.S_true, self.F_Com)
        return self._region(self.S_true, hs, self.ComNu_RMPC(hs))[1], hs.reshape(-1, 1)

    def ComFeasibleRegion_UQOPT(self):
        """[F_N_hat_opt, hs_hat_opt] = ComFeasibleRegion_UQOPT()  (:329-378)"""
        hs = self._hs(self.S_hat_opt, self.F_Com)
        return self._region(self.S_hat_opt, hs, self.ComNu_RMPC(hs))[1], hs.reshape(-1, 1)

    def ComFeasibleRegion_UQMPC(self, S_hat, h_k):
        """[F_Ns, F_N] = ComFeasibleRegion_UQMPC(S_hat, h_k)  (:380-435): rows i = 0..Nu_k with 1 - h_k"""
        hs = self._hs(self.S, self.F_Com)
        Nu_k = self.Com_Nuk(h_k, hs, self.ComNu_RMPC(hs))
        return self._region(S_hat, h_k, Nu_k)

    def ComFeasibleRegion_UQMPC_Hs_ini(self, S_hat_ini):
        """hs_ini = ComFeasibleRegion_UQMPC_Hs_ini(S_hat_ini)  (:437-445)"""
        return self._hs(S_hat_ini, self.F_Com).reshape(-1, 1)


def minHRep(P, tol=1e-9, return_keep=False):
    """MPT3's minHRep for a full-dimensional polytope in H-rep: drop the rows the others imply.
    (offline_parameters_computation_nxn.m:221-222, :399; ComputeFeasibleRegion.m:112.)  The m redundancy LPs
    (maximise a_j over the rows != j) run on the device, one per lane (lbmpc_polytope_redundant); host logic:
    shift an interior point to 0, merge parallel duplicates (keep the tightest), drop the rows whose LP value stays
    within tol of b_j, then CHECK the result with one more batched LP -- every dropped normal's support over the
    kept rows must stay <= b_j + 10 tol, else the row is put back.  Returns Polyhedron(A, b[, V]) (+ kept row ids)."""
    A, b = np.asarray(P.A, dtype=np.float64), np.asarray(P.b, dtype=np.float64).ravel()
    x0 = np.zeros(A.shape[1])
    if not np.all(b > 0):
        if P.V is None:
            raise ValueError("minHRep: 0 is not strictly inside and no vertices are known to find an interior point")
        x0 = P.V.mean(0)
    bs = b - A @ x0
    if not np.all(bs > 0):
        raise ValueError("minHRep: the polytope has no interior at the vertex mean (lower-dimensional set?)")
    nr = np.linalg.norm(A, axis=1)
    An, bn = A / nr[:, None], bs / nr
    order = np.lexsort(np.vstack([bn, np.round(An, 9).T[::-1]]))   # by direction, tightest first within a direction
    uniq = []
    for j in order:
        if uniq and np.abs(An[j] - An[uniq[-1]]).max() <= 1e-9:
            continue
        uniq.append(int(j))
    uniq = np.array(sorted(uniq))
    slack, st = core.polytope_redundant(An[uniq], bn[uniq])
    if np.any((st != 0) & (st != _abi.ST_LP_UNBOUNDED)):
        raise RuntimeError(f"minHRep: redundancy LP failed with status {int(st.max())}")
    red = slack >= -tol
    keep, drop = uniq[~red], uniq[red]
    if len(drop) and len(keep):
        h, st2 = core.polytope_support(An[keep], bn[keep], An[drop].T)
        back = (st2 != 0) | (h > bn[drop] + 10 * tol)
        keep = np.sort(np.concatenate([keep, drop[back]]))
    elif not len(keep):
        keep = uniq
    out = Polyhedron(A=A[keep], b=b[keep], V=P.V)
    return (out, keep) if return_keep else out


def hull(V):
    """Polyhedron('V', V) with both representations: extreme points and facets of conv(V) (qhull on the host, as
    MPT3 calls CDD; triangulated facets of one plane are merged)"""
    from scipy.spatial import ConvexHull
    V = np.asarray(V, dtype=np.float64)
    ch = ConvexHull(V)
    eq = ch.equations
    H = np.hstack([eq[:, :-1], -eq[:, -1:]])
    H = H[np.lexsort(np.round(H, 9).T[::-1])]
    keep = [0]
    for i in range(1, len(H)):
        if np.abs(H[i] - H[keep[-1]]).max() > 1e-9:
            keep.append(i)
    H = H[keep]
    return Polyhedron(V=V[np.sort(ch.vertices)], A=H[:, :-1], b=H[:, -1])


def minkowski_sum(*vertex_sets):
    """P1 (+) P2 (+) ... from vertex lists (MPT3's `+` on Polyhedra, offline_parameters_computation_nxn.m:166):
    hull of all vertex sums, facets pruned by minHRep on the device"""
    acc = np.asarray(vertex_sets[0], dtype=np.float64)
    for V in vertex_sets[1:]:
        V = np.asarray(V, dtype=np.float64)
        acc = (acc[:, None, :] + V[None, :, :]).reshape(-1, acc.shape[1])
        acc = hull(acc).V
    return minHRep(hull(acc))


def case_5_initial_state(V):
    """x_RM_0 of Case_5...:45-47 from the vertices of F_N_Hat: first coordinate 0.1 inside the lower extreme, second
    0.3 inside the upper extreme, the rest 0"""
    V = np.asarray(V, dtype=np.float64)
    x = np.zeros(V.shape[1])
    x[0] = V[:, 0].min() + 0.1
    if V.shape[1] >= 2:
        x[1] = V[:, 1].max() - 0.3
    return x


def case_5_start(P, oriented=True, card_I0=10, seed=5):
    """Case_5...:25-47 as written, for a flat parameter dict P: learn one information set of card_I0 samples, build
    S_hat_ini, its tightening hs_ini and the feasible region F_N_Hat = F_Ns (+) S_hat_ini (ComputeFeasibleRegion on the
    polytope-support LP kernel), and return (x_RM_0, F_N_Hat): the near-boundary start state of the feasibility study."""
    from . import params
    nx = int(P["nx"])
    S = Polyhedron(A=P["S_A"], b=P["S_b"])
    fr = ComputeFeasibleRegion(dict(S=S, S_true=S, S_hat_opt=S, F=P["F"], G=P["G"], K=P["K"], F_bar=P["F_bar"], Psi=P["Psi"],
                                    nc=int(P["nc"]), nx=nx, nu=int(P["nu"]), N=int(P["N"]), Ps=None))
    _, m, _ = core.sample_information_sets(params.sampler(P, oriented=oriented), P["W_A"], card_I0, 1, seed=seed, set0=0, want_samples=False)
    a, v, _, _ = core.homothety_lp(P["W_A"], P["W_b"], m[0])
    S_hat = a * S + (1 - a) * (np.linalg.inv(np.eye(nx) - np.asarray(P["Phi"], dtype=np.float64)) @ v)
    _, F_N_Hat = fr.ComFeasibleRegion_UQMPC(S_hat, fr.ComFeasibleRegion_UQMPC_Hs_ini(S_hat))
    return case_5_initial_state(F_N_Hat.V), F_N_Hat


class ScenarioMPC:
    """fields: A, B, N_SC, Q, R, N, F, G, nu, nx, min_u_LV, max_u_LV, Xi_true_EV, Xi_true_LV (ScenarioMPC.m:30-48).
    solve(x_k) draws N_SC x N fresh scenarios on the device and returns u_k (:52-58); every call consumes the next
    stream of this object.  scmpc_monte_carlo() below is the batched closed loop."""
    _calls = itertools.count()

    def __init__(self, parameters, seed=0):
        p = parameters
        self.nx, self.nu, self.N = int(p["nx"]), int(p["nu"]), int(p["N"])
        self.F = np.asarray(p["F"], dtype=np.float64).reshape(-1, self.nx)
        self.keep = _abi.make_scmpc_params(self.nx, self.nu, self.N, self.F.shape[0], int(p["N_SC"]), p["A"], p["B"], p["Q"],
                                           np.reshape(p["R"], (self.nu, self.nu)), self.F, p["G"])
        self.sampler = _abi.make_sampler(self.nx, self.nu, p["Xi_true_EV"].A, p["Xi_true_EV"].b, p["Xi_true_EV"].V,
                                         p["Xi_true_LV"].A, p["Xi_true_LV"].b, p["Xi_true_LV"].V, p["B"], p["min_u_LV"],
                                         p["max_u_LV"])
        self.seed = seed

    def solve(self, x_k):
        """u_k = obj.solve(x_k)"""
        lib = core.init()
        rc, o = _abi.scmpc_mc_run(lib, "lbmpc_", self.keep, self.sampler, trials=1, steps=1, seed=self.seed, x0=np.ravel(x_k),
                                  acc_bound=np.inf, distance_error=np.inf, trial0=next(ScenarioMPC._calls), traj_trials=1)
        core._chk(rc, "lbmpc_scmpc_mc_run")
        self.status = int(o["status"][0])
        return o["traj_u"][0, 0].reshape(-1, 1)


def scmpc_monte_carlo(opts_SCMPC, N_MC, K_N, x_RM_0, acc_bound, distance_error, seed=0, trial0=0, record_trials=0):
    """Case_6's SCMPC loop (Case_6...:96-150) for N_MC trials at once; trial ids and seed select the same plant
    disturbances as monte_carlo() with the same arguments."""
    sc = opts_SCMPC if isinstance(opts_SCMPC, ScenarioMPC) else ScenarioMPC(opts_SCMPC, seed=seed)
    lib = core.init()
    rc, o = _abi.scmpc_mc_run(lib, "lbmpc_", sc.keep, sc.sampler, trials=N_MC, steps=K_N, seed=seed, x0=np.ravel(x_RM_0),
                              acc_bound=acc_bound, distance_error=distance_error, trial0=trial0, traj_trials=record_trials)
    core._chk(rc, "lbmpc_scmpc_mc_run")
    return dict(Infeasible_Index=o["infeasible"], status=o["status"], qp_iters=o["qp_iters"], x_K=o["x_final"],
                State_RM=np.transpose(o["traj_x"], (0, 2, 1)), Control_EV=np.transpose(o["traj_u"], (0, 2, 1)))
