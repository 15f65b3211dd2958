#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's stored .mat DATA files.

Runs ONLY in the build container (needs /root/reference).  It reads data files with scipy
(`loadmat`, no code execution, no pickle) and decodes the MPT3 Polyhedron objects with
lbmpc/mcos.py.  No reference source text is copied: the outputs
are numeric arrays (inputs + expected outputs of the hot path) stored as .npz (allow_pickle=False).

Fixtures written (all fp64 unless noted):
  params_2x2.npz        parameters_2x2.mat : opts_UQMPC matrices, W/S/Xi/W_true polytopes (H=[A b], V)
  params_upstream.npz   parameters.mat     : same layout (S has 41 rows)
  results_1.npz         Results/Results_1.mat : 5 sample sets -> Alpha_ini, V_ini, W_Hat_ini H-reps
  results_3_small.npz / results_3_large.npz : 50-step closed-loop KATs (x_k, u_k, s_k, h_k, S_hat, W_hat)
  results_5.npz         Results/Results_5_*.mat : Infeasible_Index + trajectories (statistical)
  results_2.npz         Results/Results_2.mat : N_sam_ini, W_Hat_Vol (statistical)
  stats.npz             stats_{2x2,3x3,4x4}.mat : alpha_opt, v_opt, set sizes, timings
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "learning-based-mpc_amd"))
from lbmpc.mcos import McosFile  # noqa: E402

REF = os.environ.get("LBMPC_REFERENCE", "/root/reference")


def f64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def poly(f, opaque, name, out):
    p = f.polyhedron(opaque)
    if p["H"] is not None:
        out[name + "_H"] = f64(p["H"])
    if p["V"] is not None:
        out[name + "_V"] = f64(p["V"])


def params(path, dst):
    f = McosFile(path)
    P = f.mat["parameters"][0, 0]
    U = P["opts_UQMPC"][0, 0]
    out = {}
    for k in ["A", "B", "F", "G", "K", "Px", "Pc", "F_bar", "Psi", "Phi", "E", "hs", "Ps"]:
        out[k] = f64(U[k])
    for k in ["N", "nx", "nu", "nc", "Nu", "num_half_space_S"]:
        out[k] = np.int64(int(np.asarray(U[k]).ravel()[0]))
    poly(f, U["W"], "W", out)
    poly(f, U["S"], "S", out)
    C = P["opts_Car"][0, 0]
    poly(f, C["Xi_true_EV"], "Xi_EV", out)
    poly(f, C["Xi_true_LV"], "Xi_LV", out)
    out["min_u_LV"] = f64(C["min_u_LV"]).ravel()
    out["max_u_LV"] = f64(C["max_u_LV"]).ravel()
    FR = P["opts_feasible_region"][0, 0]
    poly(f, FR["S_true"], "S_true", out)
    poly(f, FR["S_hat_opt"], "S_hat_opt", out)
    out["hs_feasible_region"] = f64(FR["hs"])
    poly(f, P["W_true"], "W_true", out)
    poly(f, P["W_hat_opt"], "W_hat_opt", out)
    for k in ["F_N_RMPC", "F_N_True", "F_N_Hat_Opt"]:
        poly(f, P[k], k, out)
    out["x_des"] = f64(P["x_des"]).ravel()
    out["distance_error"] = f64(P["distance_error"]).ravel()
    out["acc_bound"] = f64(P["acc_bound"]).ravel()
    out["T"] = f64(P["T"]).ravel()
    I = P["opts_ini_set"][0, 0]
    out["N_sam_opt"] = np.int64(int(np.asarray(I["N_sam_opt"]).ravel()[0]))
    np.savez_compressed(dst, **out)
    return out


def results_1(dst):
    f = McosFile(os.path.join(REF, "Results/Results_1.mat"))
    R = f.mat["Results_1"][0, 0]
    out = {
        "N_sam_ini": np.asarray(R["N_sam_ini"]).ravel().astype(np.int64),
        "Alpha_ini": f64(R["Alpha_ini"]).ravel(),
        "V_ini": f64(R["V_ini"]),
    }
    for i in range(5):
        out[f"Samples_{i}"] = f64(R["Samples_ini"][i, 0])
        poly(f, R["W_Hat_ini"][i, 0], f"W_Hat_{i}", out)
    poly(f, R["W"], "W", out)
    poly(f, R["W_true"], "W_true", out)
    poly(f, R["W_hat_opt"], "W_hat_opt", out)
    np.savez_compressed(dst, **out)


def results_3(tag, dst):
    f = McosFile(os.path.join(REF, f"Results/Results_3_{tag}.mat"))
    R = f.mat[f"Results_3_{tag}"][0, 0]
    wkey = "W_Hat_Small" if tag == "small" else "W_Hat_large"
    out = {
        "Control_EV": f64(R[f"Control_EV_UQMPC_{tag}"]),
        "State_RM": f64(R[f"State_RM_UQMPC_{tag}"]),
        "Nominal_RM_State": f64(R[f"Nominal_RM_State_UQMPC_{tag}"]),
        "Hs": f64(R[f"Hs_UQMPC_{tag}"]),
        "Time_UQMPC": f64(R[f"Time_UQMPC_{tag}"]).ravel(),
    }
    SH, WH = [], []
    for k in range(50):
        SH.append(f.polyhedron(R[f"S_Hat_{tag}"][k, 0])["H"])
        WH.append(f.polyhedron(R[wkey][k, 0])["H"])
    out["S_Hat_H"] = f64(np.stack(SH))
    out["W_Hat_H"] = f64(np.stack(WH))
    np.savez_compressed(dst, **out)


def results_5(dst):
    out = {}
    for n in [10, 100, 200, 500, 2000, 5000]:
        f = McosFile(os.path.join(REF, f"Results/Results_5_{n}.mat"))
        R = f.mat[f"Results_5_{n}"][0, 0]
        out[f"x_RM_0_{n}"] = f64(R["x_RM_0"]).ravel()
        out[f"Infeasible_Index_{n}"] = np.asarray(R["Infeasible_Index"]).ravel().astype(np.int64)
        out[f"Control_EV_{n}"] = f64(np.stack([np.asarray(c).ravel() for c in R["Control_EV"].ravel()]))
        out[f"State_RM_{n}"] = f64(np.stack([np.asarray(c) for c in R["State_RM"].ravel()]))
    np.savez_compressed(dst, **out)


def results_2(dst):
    f = McosFile(os.path.join(REF, "Results/Results_2.mat"))
    R = f.mat["Results_2"][0, 0]
    np.savez_compressed(dst, N_sam_ini=np.asarray(R["N_sam_ini"]).ravel().astype(np.int64),
                        W_Hat_Vol=f64(R["W_Hat_Vol"]))


def stats(dst):
    out = {}
    for tag in ["2x2", "3x3", "4x4"]:
        f = McosFile(os.path.join(REF, f"stats_{tag}.mat"))
        S = f.mat["stats"][0, 0]
        for k in S.dtype.names:
            if k == "start_time":
                continue
            out[f"{tag}_{k}"] = f64(S[k]).ravel()
    np.savez_compressed(dst, **out)


def main():
    params(os.path.join(REF, "parameters_2x2.mat"), os.path.join(HERE, "params_2x2.npz"))
    params(os.path.join(REF, "parameters.mat"), os.path.join(HERE, "params_upstream.npz"))
    results_1(os.path.join(HERE, "results_1.npz"))
    results_3("small", os.path.join(HERE, "results_3_small.npz"))
    results_3("large", os.path.join(HERE, "results_3_large.npz"))
    results_5(os.path.join(HERE, "results_5.npz"))
    results_2(os.path.join(HERE, "results_2.npz"))
    stats(os.path.join(HERE, "stats.npz"))
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
