""".mat files in the layouts of the reference's Case scripts.  The writers are checked against the
field names and shapes of the reference's stored files (decoded once in the build container: Results_5_10.mat has
x_RM_0 nx x 1, N_pre_sam 1 x 1, Control_EV / State_RM N_MC x 1 cells of nu x K_N / nx x (K_N+1), Infeasible_Index
N_MC x 1; Results_3_small.mat the eight *_small fields) by reading the written files back with scipy."""
import os

import numpy as np
import scipy.io as sio

import fixtures
from lbmpc import matio


def test_results_5_layout(tmp_path):
    rng = np.random.default_rng(0)
    n, K = 7, 20
    mc = dict(Infeasible_Index=rng.integers(0, 2, n), Control_EV=rng.normal(size=(n, 1, K)), State_RM=rng.normal(size=(n, 2, K + 1)))
    p = os.path.join(tmp_path, "Results_5_10.mat")
    assert matio.save_results_5(p, 10, [-4.9, 1.9], mc) == "Results_5_10"
    R = sio.loadmat(p)["Results_5_10"]
    assert set(R.dtype.names) == {"x_RM_0", "N_pre_sam", "Control_EV", "State_RM", "Infeasible_Index"}
    r = R[0, 0]
    assert r["x_RM_0"].shape == (2, 1) and r["N_pre_sam"].shape == (1, 1) and r["Infeasible_Index"].shape == (n, 1)
    assert r["Control_EV"].shape == (n, 1) and r["State_RM"].shape == (n, 1)
    assert r["State_RM"][3, 0].shape == (2, K + 1) and r["Control_EV"][3, 0].shape == (1, K)
    np.testing.assert_array_equal(r["State_RM"][3, 0], mc["State_RM"][3])


def test_results_3_2_4_layouts(tmp_path):
    rng = np.random.default_rng(1)
    K, nS = 50, 80
    trial = dict(State_RM=rng.normal(size=(2, K + 1)), Control_EV=rng.normal(size=(1, K)), Nominal_RM_State=rng.normal(size=(2, K)))
    SA, WA = rng.normal(size=(nS, 2)), np.vstack([np.eye(2), -np.eye(2)])
    p = os.path.join(tmp_path, "r3.mat")
    name = matio.save_results_3(p, "small", trial, rng.normal(size=(4, K)), SA, rng.uniform(1, 2, (K, nS)), WA, rng.uniform(0.1, 0.5, (K, 4)), 0.5)
    r = sio.loadmat(p)[name][0, 0]
    want = {"Control_EV_UQMPC_small": (1, 50), "State_RM_UQMPC_small": (2, 51), "Nominal_RM_State_UQMPC_small": (2, 51),
            "Hs_UQMPC_small": (4, 50), "S_Hat_small": (50, 1), "W_Hat_Small": (50, 1), "t_small": (1, 51), "Time_UQMPC_small": (49, 1)}
    assert {k: r[k].shape for k in r.dtype.names} == want
    assert r["S_Hat_small"][0, 0]["A"][0, 0].shape == (nS, 2) and r["S_Hat_small"][0, 0]["b"][0, 0].shape == (nS, 1)
    p2 = os.path.join(tmp_path, "r2.mat")
    matio.save_results_2(p2, np.round(2 ** np.arange(2, 14.5, 0.5)), rng.uniform(size=(25, 30)), W=dict(A=WA, b=np.ones(4)))
    r2 = sio.loadmat(p2)["Results_2"][0, 0]
    assert r2["N_sam_ini"].shape == (1, 25) and r2["W_Hat_Vol"].shape == (25, 30)
    p4 = os.path.join(tmp_path, "r4.mat")
    matio.save_results_4(p4, 0.5, 1000, rng.uniform(size=(10, 1000)))
    r4 = sio.loadmat(p4)["Results_4"][0, 0]
    assert r4["t"].shape == (1, 1001) and r4["W_Hat_Vol"].shape == (10, 1000) and float(r4["K_N"][0, 0]) == 1000


def test_parameters_round_trip(tmp_path):
    """save_parameters writes the nested opts_* structs; plain loadmat reads the matrices back"""
    P = fixtures.params_dict()
    p = os.path.join(tmp_path, "parameters_out.mat")
    matio.save_parameters(p, P)
    m = sio.loadmat(p)["parameters"][0, 0]
    U = m["opts_UQMPC"][0, 0]
    np.testing.assert_array_equal(U["Psi"], P["Psi"])
    np.testing.assert_array_equal(U["S"][0, 0]["A"], P["S_H"][:, :-1])
    assert float(U["Nu"][0, 0]) == 19 and m["opts_Car"][0, 0]["Xi_true_EV"][0, 0]["V"].shape == (4, 2)


def test_load_parameters_reads_the_reference_file_when_present():
    """in the build container the reference's parameters_2x2.mat decodes to the committed fixture"""
    path = "/root/reference/parameters_2x2.mat"
    if not os.path.exists(path):
        import pytest
        pytest.skip("reference data not present (GPU box)")
    L, P = matio.load_parameters(path), fixtures.params_dict()
    for k in ("A", "K", "Psi", "S_H", "S_V", "W_H", "Xi_EV_V", "W_true_H", "hs"):
        np.testing.assert_array_equal(L[k], P[k], err_msg=k)
    assert int(L["Nu"]) == 19
