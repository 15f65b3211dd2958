"""Build-generated 3x3 / 4x4 parameter sets (SURVEY.md hard part 5: the reference's parameters_3x3.mat and
parameters_4x4.mat are missing from the checkout, and its 4x4 Xi sets come from MATLAB's rng(42)).

This is host-side set-up code run ONCE in the build container (numpy/scipy); its output
(learning-based-mpc_amd/params/params_{3x3,4x4}.npz) is data consumed by the HIP path.  It follows the
definitions of offline_parameters_computation_nxn.m: system :18-32, Xi sets :44-99, W :181-204, F/G :238-281,
Psi/F_bar/Pc :309-340, epsilon :385, with these documented substitutions:
  * dlqr -> scipy.linalg.solve_discrete_are (same Riccati solution);
  * MRPISet's sampled hull (stochastic inner approximation, MRPISet.m:66-136) -> the H-rep template polytope
    {x : D_j' x <= h_S(D_j)} of the exact Minkowski-sum support, with as many rows as the reference's S had
    (795 for 3x3, 1644 for 4x4, stats_*.mat: S_num_constraints); (s, alpha) from the Rakovic loop at epsilon = 0.1;
  * 4x4 Xi rotation: QR of (I + pi/18 * G) with G a fixed matrix written below instead of randn under rng(42);
  * nu_s: smallest nu >= N such that max F_bar Psi^(nu+1) z over {F_bar Psi^i z <= 1-hs, i<=nu} stays <= 1-hs
    (the textbook condition; ComNu_RMPC's LP only bounds the last block, ComputeFeasibleRegion.m:66-116),
    floored at the reference's 47 (= 5.875 * 8, Case_5...:5) so the QP has at least the reference's row count.
"""
import os

import numpy as np
from scipy.linalg import solve_discrete_are
from scipy.optimize import linprog
from scipy.spatial import ConvexHull

PARAMS_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "params")

G4_EV = np.array([[0.30, -0.14, 0.65, 1.52], [-0.23, -0.23, 1.58, 0.77], [-0.47, 0.54, -0.46, -0.47],
                  [0.24, -1.91, -1.72, -0.56]])
G4_LV = np.array([[-1.01, 0.31, -0.91, -1.41], [1.47, -0.23, 0.07, -1.42], [-0.54, 0.11, -1.15, 0.38],
                  [-0.60, -0.29, -0.60, 1.85]])


def hrep_from_vertices(V):
    """facets of conv(V) as rows [a b] with ||a|| = 1, duplicates merged"""
    hull = ConvexHull(V)
    eq = hull.equations  # a.x + c <= 0
    H = np.hstack([eq[:, :-1], -eq[:, -1:]])
    H = H[np.lexsort(np.round(H, 9).T[::-1])]
    keep = [0]
    for i in range(1, len(H)):
        if np.abs(H[i] - H[keep[-1]]).max() > 1e-9:
            keep.append(i)
    return H[keep]


def sphere_directions(n, nx, seed):
    """deterministic, well-spread unit directions: normalised Sobol-free construction (golden-angle lattice in 3-D,
    fixed-seed normalised Gaussians repelled by a few Lloyd-like steps in 4-D)"""
    if nx == 2:
        th = 2 * np.pi * (np.arange(n) + 0.5) / n
        return np.stack([np.cos(th), np.sin(th)], 1)
    if nx == 3:
        i = np.arange(n) + 0.5
        phi = np.arccos(1 - 2 * i / n)
        th = np.pi * (1 + 5 ** 0.5) * i
        return np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)
    rng = np.random.default_rng(seed)
    D = rng.normal(size=(n, nx))
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    for _ in range(30):  # push nearest neighbours apart
        Gm = D @ D.T
        np.fill_diagonal(Gm, -1)
        nn = Gm.argmax(1)
        D = D - 0.15 * (D[nn] - (D[nn] * D).sum(1, keepdims=True) * D)
        D /= np.linalg.norm(D, axis=1, keepdims=True)
    return D


def support_S(Phi, WV, s, alpha, D):
    """h_S(d) = (1-alpha)^-1 sum_{t<s} max_v WV (Phi^t)' d   (numpy restatement, set-up only)"""
    P = np.eye(Phi.shape[0])
    tot = np.zeros(D.shape[0])
    for _ in range(s):
        tot += (WV @ P.T @ D.T).max(0)
        P = P @ Phi
    return tot / (1 - alpha)


def s_alpha_rakovic(Phi, WV, WA, Wb, eps, max_s=60):
    nx = Phi.shape[0]
    E = np.vstack([np.eye(nx), -np.eye(nx)])
    sums = np.zeros(2 * nx)
    P = np.eye(nx)
    alpha, Ms, s = 1000.0, 1000.0, 0
    while alpha > eps / (eps + Ms) and s < max_s:
        s += 1
        sums += (WV @ P.T @ E.T).max(0)       # t = s-1
        P = P @ Phi                           # Phi^s
        alpha = ((WV @ P.T @ WA.T).max(0) / Wb).max()
        Ms = sums.max()
    return s, alpha


def compute_nu(F_bar, Psi, hs, N, nu_min, nu_max=100):
    nc, nz = F_bar.shape
    rows = [F_bar @ np.linalg.matrix_power(Psi, i) for i in range(nu_max + 2)]
    for nu in range(N, nu_max + 1):
        A_ub = np.vstack(rows[:nu + 1])
        b_ub = np.tile(1 - hs, nu + 1)
        ok = True
        for r in range(nc):
            res = linprog(-rows[nu + 1][r], A_ub=A_ub, b_ub=b_ub, bounds=[(None, None)] * nz, method="highs")
            if res.status != 0 or -res.fun > 1 - hs[r] + 1e-9:
                ok = False
                break
        if ok:
            return max(nu, nu_min), nu
    raise RuntimeError("nu_s not found")


def build(nx, n_rows_S, nu_min=47, eps=0.1):
    nu, N, T = 1, 8, 0.5
    A = np.eye(nx) * 0.5
    for i in range(nx - 1):
        A[i, i + 1] = T * 0.5
    B = np.zeros((nx, nu))
    B[-1, 0] = T * 0.5
    Q, R = np.eye(nx), np.array([[0.01]])
    Px = solve_discrete_are(A, B, Q, R)
    K = -np.linalg.solve(R + B.T @ Px @ B, B.T @ Px @ A)
    Phi = A + B @ K
    if nx == 3:
        th = np.pi / 12
        Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
        ang = np.arange(6) * np.pi / 3

        def prism(rx, ry):
            ring = np.stack([rx * np.cos(ang), ry * np.sin(ang)], 1)
            return np.vstack([np.hstack([ring, np.full((6, 1), -0.005)]), np.hstack([ring, np.full((6, 1), 0.005)])])
        V_EV, V_LV = prism(0.018, 0.012) @ Rz.T, prism(0.022, 0.010) @ Rz.T
        half_W = 1.5 * np.array([0.12, 0.08, 0.06])
        dist, acc = 15.0, 4.0
    elif nx == 4:
        bits = np.array([[(i >> (3 - k)) & 1 for k in range(4)] for i in range(16)]) * 2.0 - 1.0
        Q_EV, _ = np.linalg.qr(np.eye(4) + np.pi / 18 * G4_EV)
        Q_LV, _ = np.linalg.qr(np.eye(4) + np.pi / 18 * G4_LV)
        V_EV = (bits * np.array([0.015, 0.01, 0.008, 0.012])) @ Q_EV.T
        V_LV = (bits * np.array([0.018, 0.009, 0.007, 0.011])) @ Q_LV.T
        half_W = np.array([0.12, 0.08, 0.06, 0.09])
        dist, acc = 60.0, 60.0
    else:
        raise ValueError("nx must be 3 or 4 (2x2 comes from the reference's parameters_2x2.mat)")
    Xi_EV_H, Xi_LV_H = hrep_from_vertices(V_EV), hrep_from_vertices(V_LV)
    # prior W: axis-aligned cuboid, rows in MPT3's joint normalisation ||[a b]|| = 1
    W_rows = []
    for k in range(nx):
        for sg in (1.0, -1.0):
            a = np.zeros(nx)
            a[k] = sg
            W_rows.append(np.concatenate([a, [half_W[k]]]))
    W_H = np.array(W_rows)
    W_H /= np.linalg.norm(W_H, axis=1, keepdims=True)
    W_V = np.array([[(1 if (i >> k) & 1 else -1) * half_W[k] for k in range(nx)] for i in range(2 ** nx)], dtype=float)
    nc = 2 * nx + 2
    F, G = np.zeros((nc, nx)), np.zeros((nc, nu))
    for i in range(nx):
        F[2 * i, i], F[2 * i + 1, i] = 1 / dist, -1 / dist
    G[2 * nx, 0], G[2 * nx + 1, 0] = 1 / acc, -1 / acc
    E = np.zeros((1, N))
    E[0, 0] = 1
    Pc = np.eye(N) * float(B.T @ Px @ B + R)
    M = np.diag(np.ones(N - 1), 1)
    Psi = np.block([[Phi, B @ E], [np.zeros((N, nx)), M]])
    F_bar = np.hstack([F + G @ K, G @ E])
    # tube cross-section S: template polytope of the exact support
    s, alpha = s_alpha_rakovic(Phi, W_V, W_H[:, :-1], W_H[:, -1], eps)
    Fc = F + G @ K
    must = np.vstack([np.eye(nx), -np.eye(nx), Fc / np.linalg.norm(Fc, axis=1, keepdims=True)])
    must = must[np.unique(np.round(must, 12), axis=0, return_index=True)[1]]
    D = np.vstack([must, sphere_directions(n_rows_S - len(must), nx, seed=nx)])
    S_b = support_S(Phi, W_V, s, alpha, D)
    S_H = np.hstack([D, S_b[:, None]])
    S_H /= np.linalg.norm(S_H, axis=1, keepdims=True)
    hs = support_S(Phi, W_V, s, alpha, Fc)
    Nu, nu_exact = compute_nu(F_bar, Psi, hs, N, nu_min)
    x_des = np.zeros(nx)
    x_des[0] = -35
    return dict(nx=np.int64(nx), nu=np.int64(nu), N=np.int64(N), nc=np.int64(nc), Nu=np.int64(Nu),
                Nu_exact=np.int64(nu_exact), num_half_space_S=np.int64(len(S_H)), A=A, B=B, K=K, Phi=Phi, Px=Px, Pc=Pc,
                F=F, G=G, F_bar=F_bar, Psi=Psi, E=E, hs=hs.reshape(-1, 1), W_H=W_H, W_V=W_V, S_H=S_H,
                Xi_EV_H=Xi_EV_H, Xi_EV_V=V_EV, Xi_LV_H=Xi_LV_H, Xi_LV_V=V_LV, min_u_LV=np.array([-1 / 20]),
                max_u_LV=np.array([1 / 16]), x_des=x_des, distance_error=np.array([dist]), acc_bound=np.array([acc]),
                T=np.array([T]), mrpi_s=np.int64(s), mrpi_alpha=np.array([alpha]), mrpi_eps=np.array([eps]),
                vol_W=np.array([np.prod(2 * half_W)]))


def main():
    os.makedirs(PARAMS_DIR, exist_ok=True)
    for nx, rows in ((3, 795), (4, 1644)):
        d = build(nx, rows)
        np.savez_compressed(os.path.join(PARAMS_DIR, f"params_{nx}x{nx}.npz"), **d)
        print(nx, "s", int(d["mrpi_s"]), "alpha", float(d["mrpi_alpha"][0]), "Nu", int(d["Nu"]), "exact", int(d["Nu_exact"]),
              "hs", d["hs"].ravel().round(5), "nEV", d["Xi_EV_H"].shape, "nLV", d["Xi_LV_H"].shape, "K", d["K"].round(4))


if __name__ == "__main__":
    main()
