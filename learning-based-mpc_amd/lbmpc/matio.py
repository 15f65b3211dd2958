""".mat I/O in the layouts of the reference's scripts, scipy only.

Reading: `load_parameters(path)` turns a `parameters_*.mat` written by offline_parameters_computation_nxn.m:746-773
into the plain dict the rest of the package takes (MPT3 Polyhedra decoded to H = [A b] and V arrays by lbmpc/mcos.py;
nothing in the file is executed).

Writing: `save_results_2 / _3 / _4 / _5` write the `Results_*` structs with the field names and shapes of
Cases_general/Case_2...:31-40, Case_3...:93-115, Case_4...:76-82, Case_5...:101-153, so that Results/Fig_Case_*.m find
what they index.  One deviation, forced by the format: an MPT3 `Polyhedron` is a MATLAB class object (MCOS) that only
MATLAB can serialise, so polytopes are written as structs with fields A, b (and V when known);
learning-based-mpc_amd/matlab/lbmpc_poly.m turns such a struct (or a cell array of them) back into Polyhedron objects.
"""
import numpy as np
import scipy.io as sio

from .mcos import McosFile


def _f64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def _poly_struct(A=None, b=None, V=None):
    d = {}
    if A is not None:
        d["A"], d["b"] = _f64(A), _f64(b).reshape(-1, 1)
    if V is not None:
        d["V"] = _f64(V)
    return d


def _cell(items):
    c = np.empty((len(items), 1), dtype=object)
    for i, it in enumerate(items):
        c[i, 0] = it
    return c


def load_parameters(path):
    """parameters_*.mat -> dict with the keys of lbmpc.params.load (A, B, K, Phi, Px, Pc, F, G, F_bar, Psi, hs, Nu, ...,
    S_H/S_V, W_H/W_V, Xi_EV_H/V, Xi_LV_H/V, W_true_H/V, x_des, distance_error, acc_bound, T)"""
    f = McosFile(path)
    P = f.mat["parameters"][0, 0]
    U = P["opts_UQMPC"][0, 0]
    out = {}
    for k in ["A", "B", "F", "G", "K", "Px", "Pc", "F_bar", "Psi", "Phi", "E", "hs", "Ps"]:
        if k in U.dtype.names:
            out[k] = _f64(U[k])
    for k in ["N", "nx", "nu", "nc", "Nu", "num_half_space_S"]:
        if k in U.dtype.names:
            out[k] = np.int64(int(np.asarray(U[k]).ravel()[0]))

    def poly(opaque, name):
        p = f.polyhedron(opaque)
        if p["H"] is not None:
            out[name + "_H"] = _f64(p["H"])
        if p["V"] is not None:
            out[name + "_V"] = _f64(p["V"])

    poly(U["W"], "W")
    poly(U["S"], "S")
    C = P["opts_Car"][0, 0]
    poly(C["Xi_true_EV"], "Xi_EV")
    poly(C["Xi_true_LV"], "Xi_LV")
    out["min_u_LV"], out["max_u_LV"] = _f64(C["min_u_LV"]).ravel(), _f64(C["max_u_LV"]).ravel()
    for k in ("W_true", "W_hat_opt", "F_N_RMPC", "F_N_True", "F_N_Hat_Opt"):
        if k in P.dtype.names:
            poly(P[k], k)
    for k in ("x_des", "distance_error", "acc_bound", "T"):
        if k in P.dtype.names:
            out[k] = _f64(P[k]).ravel()
    out["S_A"], out["S_b"] = out["S_H"][:, :-1], out["S_H"][:, -1]
    out["W_A"], out["W_b"] = out["W_H"][:, :-1], out["W_H"][:, -1]
    return out


def save_results_5(path, N_pre_sam, x_RM_0, mc):
    """Case_5...:101-153.  mc: the dict of lbmpc.api.monte_carlo(..., record_trials=N_MC)"""
    n = len(mc["Infeasible_Index"])
    name = f"Results_5_{int(N_pre_sam)}"
    S = {"x_RM_0": _f64(x_RM_0).reshape(-1, 1), "N_pre_sam": float(N_pre_sam),
         "Control_EV": _cell([_f64(mc["Control_EV"][i]) for i in range(n)]),        # nu x K_N each
         "State_RM": _cell([_f64(mc["State_RM"][i]) for i in range(n)]),            # nx x (K_N+1) each
         "Infeasible_Index": _f64(mc["Infeasible_Index"]).reshape(-1, 1)}
    sio.savemat(path, {name: S})
    return name


def save_results_3(path, tag, trial, hs, S_A, S_hat_b, W_A, W_hat_b, T, time_per_step=None):
    """Case_3...:93-115 for one recorded trial.  trial: dict with State_RM (nx x K+1), Control_EV (nu x K),
    Nominal_RM_State (nx x K); hs: nc x K; S_hat_b / W_hat_b: K x nS / K x nW right-hand sides of S_hat_k / W_hat_k"""
    K = np.asarray(trial["Control_EV"]).shape[1]
    suf = "_" + tag
    nom = np.asarray(trial["Nominal_RM_State"], dtype=np.float64)
    if nom.shape[1] == K:   # the reference allocates K_N + 1 columns (Case_3...:47) and fills the first K_N
        nom = np.hstack([nom, np.ones((nom.shape[0], 1))])
    S = {"Control_EV_UQMPC" + suf: _f64(trial["Control_EV"]), "State_RM_UQMPC" + suf: _f64(trial["State_RM"]),
         "Nominal_RM_State_UQMPC" + suf: nom, "Hs_UQMPC" + suf: _f64(hs),
         "S_Hat" + suf: _cell([_poly_struct(S_A, S_hat_b[k]) for k in range(K)]),
         ("W_Hat_Small" if tag == "small" else "W_Hat" + suf): _cell([_poly_struct(W_A, W_hat_b[k]) for k in range(K)]),
         "t" + suf: (np.arange(K + 1) * float(T)).reshape(1, -1),
         "Time_UQMPC" + suf: _f64(np.zeros(K - 1) if time_per_step is None else np.ravel(time_per_step)[1:]).reshape(-1, 1)}
    name = "Results_3" + suf
    sio.savemat(path, {name: S})
    return name


def save_results_2(path, N_sam_ini, W_Hat_Vol, W=None, extra=None):
    """Case_2...:31-40.  W_Hat_Vol: len(N_sam_ini) x N_MC; W: dict(A, b, V) of the prior set; extra: more polytopes
    (W_hat_opt, W_true, F_N_*) as dicts"""
    S = {"N_sam_ini": _f64(N_sam_ini).reshape(1, -1), "W_Hat_Vol": _f64(W_Hat_Vol)}
    if W is not None:
        S["W"] = _poly_struct(**W)
    for k, v in (extra or {}).items():
        S[k] = _poly_struct(**v)
    sio.savemat(path, {"Results_2": S})
    return "Results_2"


def save_results_4(path, T, K_N, W_Hat_Vol, Samples=None, W_hat_opt=None):
    """Case_4...:76-82.  W_Hat_Vol: N_MC x K_N"""
    S = {"t": (np.arange(int(K_N) + 1) * float(T)).reshape(1, -1), "K_N": float(K_N), "W_Hat_Vol": _f64(W_Hat_Vol)}
    if Samples is not None:
        S["Samples"] = _cell([_f64(s) for s in Samples])
    if W_hat_opt is not None:
        S["W_hat_opt"] = _poly_struct(**W_hat_opt)
    sio.savemat(path, {"Results_4": S})
    return "Results_4"


def save_parameters(path, P):
    """a `parameters` struct with the nested opts_* structs the Case scripts read (Case_3...:5-20), polytopes as
    {A, b, V} structs (see the module docstring)"""
    def poly(name):
        H, V = P.get(name + "_H"), P.get(name + "_V")
        return _poly_struct(None if H is None else H[:, :-1], None if H is None else H[:, -1], V)

    U = {k: _f64(P[k]) for k in ("A", "B", "F", "G", "K", "Px", "Pc", "F_bar", "Psi", "Phi", "hs") if k in P}
    for k in ("N", "nx", "nu", "nc", "Nu"):
        U[k] = float(np.ravel(P[k])[0])
    U["W"], U["S"] = poly("W"), poly("S")
    U["num_half_space_S"] = float(P["S_H"].shape[0])
    car = {"A": U["A"], "B": U["B"], "min_u_LV": _f64(P["min_u_LV"]), "max_u_LV": _f64(P["max_u_LV"]),
           "Xi_true_EV": poly("Xi_EV"), "Xi_true_LV": poly("Xi_LV")}
    ini = {"B": U["B"], "nx": U["nx"], "W": U["W"], "min_u_LV": car["min_u_LV"], "max_u_LV": car["max_u_LV"],
           "Xi_true_EV": car["Xi_true_EV"], "Xi_true_LV": car["Xi_true_LV"], "N_pre_sam": 100.0,
           "N_sam_opt": float(np.ravel(P.get("N_sam_opt", 0))[0])}
    top = {"opts_UQMPC": U, "opts_RMPC": U, "opts_Car": car, "opts_ini_set": ini}
    if "Ps" in P and P["Ps"] is not None:
        U["Ps"] = _f64(P["Ps"])
    if "S_true_H" in P and "S_hat_opt_H" in P:   # what ComputeFeasibleRegion's constructor reads (:31-46)
        fea = {k: U[k] for k in ("F", "G", "K", "F_bar", "Psi", "nc", "nx", "nu", "N", "num_half_space_S", "hs") if k in U}
        fea.update(S=U["S"], S_true=poly("S_true"), S_hat_opt=poly("S_hat_opt"))
        if "Ps" in U:
            fea["Ps"] = U["Ps"]
        top["opts_feasible_region"] = fea
    for k in ("F_N_RMPC", "F_N_True", "F_N_Hat_Opt"):   # the reference stores these as vertex arrays
        if k + "_V" in P:
            top[k] = _f64(P[k + "_V"])
    for k in ("x_des", "distance_error", "acc_bound", "T"):
        if k in P:
            top[k] = _f64(P[k]).reshape(-1, 1)
    for k in ("W_true", "W_hat_opt"):
        if k + "_H" in P:
            top[k] = poly(k)
    sio.savemat(path, {"parameters": top})
