"""offline_parameters_computation_nxn.m (and its 2x2 predecessor offline_parameters_computation.m) on the HIP path
(SURVEY.md 8f-2): the scripts that build `parameters_nxn.mat`.

Every set operation of the script runs through liblbmpc_hip.so:
    W_true = Xi_EV (+) (-Xi_LV) (+) (-B U_LV)      :166      api.minkowski_sum  (hull on the host, facets pruned by the
                                                            device redundancy LPs of lbmpc_polytope_redundant)
    minHRep(W_true), minHRep(W), minHRep(S...)     :221-222, :399   api.minHRep (same LPs)
    IniSet.solve_opt(samples_opt)                  :351      lbmpc_halfspace_max + lbmpc_homothety_lp
    MRPISet(Phi, W_hat_opt | W_true | W, epsilon)  :397-540  lbmpc_mrpi_s_alpha + lbmpc_support_batch (H-rep template)
    ComputeFeasibleRegion: RMPC / True / UQOPT     :653-656  lbmpc_polytope_support
Host numpy/scipy is left with what is O(nx^3) once: the Riccati solution behind dlqr (:30), the facet enumeration
of <= 64 summed vertices (qhull), and compute_Ps (:827-848).

compute_Ps: `max logdet(Ps) s.t. |Ps p_i| <= 1` is the centred minimum-volume ellipsoid of the points p_i: with
X = Ps^2 the constraint reads p_i' X p_i <= 1, and the maximiser is X = (n M(u))^-1 for the D-optimal weights u of
M(u) = sum u_i p_i p_i'.  Solved by the multiplicative D-optimal iteration (u_i <- u_i kappa_i / n) instead of an SDP
solver.  The two scripts differ in the points: offline_parameters_computation.m:137-154 takes the VERTICES of
Omega = {F_bar Psi^i z <= 1 - hs, i <= Nu} (that is the Ps stored in parameters_2x2.mat: all 1024 vertices of the
stored Omega sit on its unit level set -- the golden test for this function), the nxn script's compute_Ps (:827-848)
the ROWS of that H-rep.  Ps only feeds Com_Nuk, which the Case scripts never reach.

script="2x2" is the system behind the stored parameters_2x2.mat: the double integrator A = [1 T; 0 1], B = [0; T],
R = 0.1, epsilon = 1e-2 of offline_parameters_computation.m:9-16,76-78 with the prior W = +-(0.5, 0.2) the stored
file holds (the nxn script's 2-D branch, :182).

U_true_LV: the script's live line `Polyhedron([1/max_u_LV; -1/min_u_LV], [1; -1])` describes u <= -1/20 (unbounded);
the commented-out line below it, [min_u_LV, max_u_LV], is what the stored W_true (6 vertices) was built from, and is
what is used here (pinned by tests/golden/params_2x2.npz: W_true_V).
"""
import time

import numpy as np
from scipy.linalg import solve_discrete_are

from . import _abi, api

# 4x4 Xi rotations: QR of (I + pi/18 * G) with these fixed matrices stands in for MATLAB's randn under rng(42)
# (offline_parameters_computation_nxn.m:90-96 -- not reproducible outside MATLAB; SURVEY.md hard part 5)
G4_EV = np.array([[0.30, -0.14, 0.65, 1.52], [-0.23, -0.23, 1.58, 0.77], [-0.47, 0.54, -0.46, -0.47],
                  [0.24, -1.91, -1.72, -0.56]])
G4_LV = np.array([[-1.01, 0.31, -0.91, -1.41], [1.47, -0.23, 0.07, -1.42], [-0.54, 0.11, -1.15, 0.38],
                  [-0.60, -0.29, -0.60, 1.85]])


def definitions(nx, script="nxn"):
    """system, disturbance sets, prior W and constraint scales of offline_parameters_computation_nxn.m:17-99, :181-204,
    :226-281 (4x4: with the stand-in above for the rng(42) rotations); script="2x2": see the
    module docstring"""
    T, N, nu = 0.5, 8, 1
    A = np.eye(nx) * 0.5
    for i in range(nx - 1):
        A[i, i + 1] = T * 0.5
    B = np.zeros((nx, nu))
    B[-1, 0] = T * 0.5
    R = np.array([[0.01]])
    if script == "2x2":
        if nx != 2:
            raise ValueError('script="2x2" is the 2-state system')
        A, B, R = np.array([[1.0, T], [0.0, 1.0]]), np.array([[0.0], [T]]), np.array([[0.1]])
    if nx == 2:
        th = np.pi / 40
        Rm = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        V_EV = np.array([[-0.06, -0.015], [0.06, -0.015], [0.01, 0.025], [-0.01, 0.025]]) @ Rm.T
        V_LV = np.array([[-0.06, 0.015], [0.06, 0.015], [0.01, -0.025], [-0.01, -0.025]]) @ Rm.T
        half_W, dist, acc = np.array([0.5, 0.2]), 15.0, 2.0
        nc = 4
        F, G = np.zeros((nc, nx)), np.zeros((nc, nu))
        F[0, 0], F[1, 0], G[2, 0], G[3, 0] = 1 / dist, -1 / dist, 1 / acc, -1 / acc
    elif nx in (3, 4):
        if nx == 3:
            th = np.pi / 12
            Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
            ang = np.arange(6) * np.pi / 3

            def prism(rx, ry):
                ring = np.stack([rx * np.cos(ang), ry * np.sin(ang)], 1)
                return np.vstack([np.hstack([ring, np.full((6, 1), -0.005)]), np.hstack([ring, np.full((6, 1), 0.005)])])
            V_EV, V_LV = prism(0.018, 0.012) @ Rz.T, prism(0.022, 0.010) @ Rz.T
            half_W, dist, acc = 1.5 * np.array([0.12, 0.08, 0.06]), 15.0, 4.0
        else:
            bits = np.array([[(i >> (3 - k)) & 1 for k in range(4)] for i in range(16)]) * 2.0 - 1.0
            Q_EV, _ = np.linalg.qr(np.eye(4) + np.pi / 18 * G4_EV)
            Q_LV, _ = np.linalg.qr(np.eye(4) + np.pi / 18 * G4_LV)
            V_EV = (bits * np.array([0.015, 0.01, 0.008, 0.012])) @ Q_EV.T
            V_LV = (bits * np.array([0.018, 0.009, 0.007, 0.011])) @ Q_LV.T
            half_W, dist, acc = np.array([0.12, 0.08, 0.06, 0.09]), 60.0, 60.0
        nc = 2 * nx + 2
        F, G = np.zeros((nc, nx)), np.zeros((nc, nu))
        for i in range(nx):
            F[2 * i, i], F[2 * i + 1, i] = 1 / dist, -1 / dist
        G[2 * nx, 0], G[2 * nx + 1, 0] = 1 / acc, -1 / acc
    else:
        raise ValueError("nx must be 2, 3 or 4")
    W_V = np.array([[(1 if (i >> k) & 1 else -1) * half_W[k] for k in range(nx)] for i in range(2 ** nx)], dtype=float)
    return dict(nx=nx, nu=nu, N=N, nc=nc, T=T, A=A, B=B, Q=np.eye(nx), R=R, V_EV=V_EV, V_LV=V_LV,
                W_V=W_V, F=F, G=G, dist=dist, acc=acc, min_u_LV=-1 / 20, max_u_LV=1 / 16)


def compute_Ps(F_bar, Psi, Nu, hs=None, over="rows", tol=1e-9, max_iter=200000):
    """Ps of compute_Ps (nxn script :827-848, over="rows") or of offline_parameters_computation.m:137-154
    (over="vertices", needs hs); see the module docstring"""
    rows, M = [], np.asarray(F_bar, dtype=np.float64).copy()
    for _ in range(Nu + 1):
        rows.append(M)
        M = M @ Psi
    H = np.vstack(rows)
    if over == "vertices":
        from scipy.spatial import HalfspaceIntersection
        h = np.tile(1 - np.ravel(hs), Nu + 1)
        H = HalfspaceIntersection(np.hstack([H, -h[:, None]]), np.zeros(H.shape[1])).intersections
    H = H[np.linalg.norm(H, axis=1) > 1e-14]
    m, n = H.shape
    u = np.ones(m) / m
    for _ in range(max_iter):
        Mu = (H * u[:, None]).T @ H
        kappa = np.einsum("ij,ij->i", H @ np.linalg.inv(Mu), H)
        if kappa.max() <= n * (1 + tol):
            break
        u = u * kappa / n
    w, Q = np.linalg.eigh(np.linalg.inv(Mu) / kappa.max())   # / kappa.max(): feasible even if stopped early
    return Q @ np.diag(np.sqrt(w)) @ Q.T


def _hrep(P):
    H = np.hstack([P.A, P.b.reshape(-1, 1)])
    return H / np.linalg.norm(H, axis=1, keepdims=True)      # MPT3 stores rows with |[a b]| = 1


def offline_parameters(nx, script="nxn", epsilon=None, mrpi_mode=_abi.MS_RAKOVIC, seed=0, n_template=None, n_region=None,
                       log=None, n_template_S=None, rows_S=None, nu_floor=None):
    """Run the script for nx in {2, 3, 4}; returns (P, stats): P is the flat parameter dict of
    learning-based-mpc_amd/params/*.npz / tests/golden/params_2x2.npz (matio.save_parameters writes it as the
    reference's `parameters` struct), stats the timing / size fields of the script's `stats` struct.
    n_template_S: template size of the tube set S alone (the other two sets keep n_template); rows_S: keep exactly
    this many of the facets minHRep leaves (the axis and F + GK directions first; dropping facets of an irredundant
    H-rep leaves an irredundant H-rep of a slightly larger outer approximation) -- the committed 3x3 / 4x4 sets take
    the reference's facet counts this way (stats_*.mat: S_num_constraints = 795 / 1644, every row a facet as after its
    minHRep, nxn.m:399,449,564); nu_floor: lower bound on Nu (the reference ran 3x3 with Nu = 47, Case_5...:5)."""
    say = log or (lambda *a: None)
    d = definitions(nx, script)
    epsilon = (1e-2 if script == "2x2" else 0.1) if epsilon is None else epsilon
    A, B, N, nu, nc, F, G = d["A"], d["B"], d["N"], d["nu"], d["nc"], d["F"], d["G"]
    stats = dict(nx=nx, nu=nu, N=N)
    t_all = time.perf_counter()
    Px = solve_discrete_are(A, B, d["Q"], d["R"])
    K = -np.linalg.solve(d["R"] + B.T @ Px @ B, B.T @ Px @ A)
    Phi = A + B @ K
    if np.abs(np.linalg.eigvals(Phi)).max() >= 1:
        raise RuntimeError("closed loop is not stable: the mRPI set is unbounded")
    Xi_EV, Xi_LV = api.hull(d["V_EV"]), api.hull(d["V_LV"])
    U_V = np.array([[d["min_u_LV"]], [d["max_u_LV"]]])
    W_true = api.minkowski_sum(Xi_EV.V, -Xi_LV.V, -(U_V @ B.T))
    if np.linalg.matrix_rank(W_true.V - W_true.V[0]) < nx:
        raise RuntimeError("W_true is lower-dimensional")
    W = api.minHRep(api.hull(d["W_V"]))
    samples_opt = W_true.V
    E = np.zeros((1, N))
    E[0, 0] = 1
    Pc = np.eye(N) * float((B.T @ Px @ B + d["R"])[0, 0])
    Psi = np.block([[Phi, B @ E], [np.zeros((N, nx)), np.diag(np.ones(N - 1), 1)]])
    F_bar = np.hstack([F + G @ K, G @ E])

    t = time.perf_counter()
    ini = api.InitialSetComputation(dict(B=B, nx=nx, N_pre_sam=5, N_sam_opt=len(samples_opt), W=W, min_u_LV=d["min_u_LV"],
                                         max_u_LV=d["max_u_LV"], Xi_true_EV=Xi_EV, Xi_true_LV=Xi_LV), seed=seed)
    alpha_opt, v_opt = ini.solve_opt(samples_opt.T)
    stats["t_initialset"], stats["alpha_opt"], stats["v_opt"] = time.perf_counter() - t, alpha_opt, v_opt.ravel()
    W_hat_opt = api.Polyhedron(A=W.A, b=alpha_opt * W.b + (1 - alpha_opt) * (W.A @ v_opt.ravel()),
                               V=alpha_opt * W.vertices() + (1 - alpha_opt) * v_opt.ravel())
    say(f"alpha_opt {alpha_opt:.6f} v_opt {v_opt.ravel()}")

    Fc = F + G @ K                                            # tightening directions: h_S(F_Com) exact in the template
    Fc = Fc[np.linalg.norm(Fc, axis=1) > 0]
    D = np.vstack([api.template_directions(nx, n_template, seed=seed), Fc / np.linalg.norm(Fc, axis=1, keepdims=True)])
    sets = {}
    n_must = 2 * nx + len(Fc)
    for name, Wset in (("S_hat_opt", W_hat_opt), ("S_true", W_true), ("S", W)):
        t = time.perf_counter()
        Dn = D
        if name == "S" and n_template_S is not None:
            Dn = np.vstack([api.template_directions(nx, n_template_S, seed=seed), Fc / np.linalg.norm(Fc, axis=1, keepdims=True)])
            Dn = np.vstack([Dn[:2 * nx], Dn[-len(Fc):], Dn[2 * nx:-len(Fc)]])      # axis and F + GK directions first
        Sraw, info = api.MRPISet(Phi, Wset, epsilon, template_directions=Dn, mode=mrpi_mode, seed=seed, return_info=True)
        Smin, keep = api.minHRep(Sraw, return_keep=True)
        if name == "S" and rows_S is not None:
            if len(keep) < rows_S:
                raise RuntimeError(f"tube set: {len(keep)} facets from {len(Dn)} template rows, {rows_S} asked for: raise n_template_S")
            if n_template_S is None:
                raise ValueError("rows_S needs n_template_S (the directions to keep come first in that template)")
            kept = keep[:rows_S]       # keep is ascending: the must-have directions (rows < n_must) stay
            Smin = api.Polyhedron(A=Sraw.A[kept], b=Sraw.b[kept])
        key = {"S_hat_opt": "t_mrpi_shat", "S_true": "t_mrpi_strue", "S": "t_mrpi_s"}[name]
        stats[key] = time.perf_counter() - t
        stats[name + "_num_constraints"], stats[name + "_template_rows"] = len(Smin.b), len(Sraw.b)
        stats[name + "_s"], stats[name + "_alpha"] = info["s"], info["alpha"]
        sets[name] = Smin
        say(f"{name}: s {info['s']} alpha {info['alpha']:.3e} rows {len(Sraw.b)} -> {len(Smin.b)} in {stats[key]:.2f} s")

    t = time.perf_counter()
    Ps = None
    fea = dict(S=sets["S"], S_true=sets["S_true"], S_hat_opt=sets["S_hat_opt"], F=F, G=G, K=K, F_bar=F_bar, Psi=Psi,
               nc=nc, nx=nx, nu=nu, N=N, num_half_space_S=len(sets["S"].b))
    reg = api.ComputeFeasibleRegion(fea, directions=None if n_region is None else api.template_directions(nx, n_region, seed))
    F_N_RMPC, hs_RMPC, Nu_RMPC = reg.ComFeasibleRegion_RMPC()
    stats["Nu_exact"] = int(Nu_RMPC)
    if nu_floor is not None and Nu_RMPC < nu_floor:
        Nu_RMPC = int(nu_floor)      # rows F_bar Psi^i beyond nu_s are implied by the earlier ones: the same set Omega, more QP rows
    F_N_True, hs_True = reg.ComFeasibleRegion_True()
    F_N_Hat_Opt, hs_Hat_Opt = reg.ComFeasibleRegion_UQOPT()
    stats["t_feareg"] = time.perf_counter() - t
    for k, P_ in (("F_N_RMPC", F_N_RMPC), ("F_N_True", F_N_True), ("F_N_Hat_Opt", F_N_Hat_Opt)):
        stats[k + "_size"] = np.array(P_.V.shape)
        stats[k + "_bbox_vol"] = float(np.prod(P_.V.max(0) - P_.V.min(0)))
    say(f"Nu {Nu_RMPC} hs {hs_RMPC.ravel()} in {stats['t_feareg']:.2f} s")
    t = time.perf_counter()
    Ps = compute_Ps(F_bar, Psi, Nu_RMPC, hs_RMPC, over="vertices" if script == "2x2" else "rows")
    stats["t_sdp"] = time.perf_counter() - t
    stats["elapsedTime"] = time.perf_counter() - t_all

    x_des = np.zeros(nx)
    x_des[0] = -35
    P = dict(nx=np.int64(nx), nu=np.int64(nu), N=np.int64(N), nc=np.int64(nc), Nu=np.int64(Nu_RMPC), A=A, B=B, K=K, Phi=Phi,
             Px=Px, Pc=Pc, F=F, G=G, F_bar=F_bar, Psi=Psi, E=E, hs=hs_RMPC, Ps=Ps,
             num_half_space_S=np.int64(len(sets["S"].b)), W_H=_hrep(W), W_V=W.V, S_H=_hrep(sets["S"]),
             S_true_H=_hrep(sets["S_true"]), S_hat_opt_H=_hrep(sets["S_hat_opt"]), Xi_EV_H=_hrep(Xi_EV), Xi_EV_V=Xi_EV.V,
             Xi_LV_H=_hrep(Xi_LV), Xi_LV_V=Xi_LV.V, min_u_LV=np.array([d["min_u_LV"]]), max_u_LV=np.array([d["max_u_LV"]]),
             W_true_H=_hrep(W_true), W_true_V=W_true.V, W_hat_opt_H=_hrep(W_hat_opt), W_hat_opt_V=W_hat_opt.V,
             hs_feasible_region=hs_RMPC, hs_True=hs_True, hs_Hat_Opt=hs_Hat_Opt, F_N_RMPC_V=F_N_RMPC.V,
             F_N_True_V=F_N_True.V, F_N_Hat_Opt_V=F_N_Hat_Opt.V, F_N_RMPC_H=np.hstack([F_N_RMPC.A, F_N_RMPC.b[:, None]]),
             F_N_Hat_Opt_H=np.hstack([F_N_Hat_Opt.A, F_N_Hat_Opt.b[:, None]]), x_des=x_des,
             distance_error=np.array([d["dist"]]), acc_bound=np.array([d["acc"]]), T=np.array([d["T"]]),
             N_sam_opt=np.int64(len(samples_opt)), vol_W=np.array([np.prod(W.V.max(0) - W.V.min(0))]),
             mrpi_s=np.int64(stats["S_s"]), mrpi_alpha=np.array([stats["S_alpha"]]), mrpi_eps=np.array([epsilon]),
             Nu_exact=np.int64(stats["Nu_exact"]))
    return P, stats


def main(argv=None):
    import argparse

    from . import matio
    ap = argparse.ArgumentParser(description="build parameters_nxn.mat on the GPU (offline_parameters_computation_nxn.m)")
    ap.add_argument("--nx", type=int, default=2)
    ap.add_argument("--script", default="nxn", choices=["nxn", "2x2"])
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    P, stats = offline_parameters(a.nx, script=a.script, log=print)
    out = a.out or f"parameters_{a.nx}x{a.nx}.mat"
    matio.save_parameters(out, P)
    print("wrote", out, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in stats.items() if k.startswith("t_")})


if __name__ == "__main__":
    main()
