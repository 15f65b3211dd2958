"""offline_parameters_computation_nxn.m on the HIP path -- device redundancy LPs (minHRep), the
Minkowski sum behind W_true, and the whole script for the 2x2 system against what parameters_2x2.mat / stats_2x2.mat
store (tests/golden/params_2x2.npz, stats.npz)."""
import os

import numpy as np
import pytest

import oracle_core
from lbmpc import _abi, api, core, matio, pipeline

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _random_polytope(rng, n, m_facets, m_extra):
    """unit-ball-ish polytope with m_facets tangent rows, m_extra rows pushed outwards (strictly redundant) and a few
    duplicated rows"""
    D = rng.normal(size=(m_facets, n))
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    b = 1.0 + 0.2 * rng.random(m_facets)
    X = rng.normal(size=(m_extra, n))
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    A = np.vstack([D, X, D[:5] * 3.0])
    bb = np.concatenate([b, 4.0 + rng.random(m_extra), b[:5] * 3.0])
    return A, bb


@pytest.mark.parametrize("n,m", [(2, 40), (3, 120), (4, 300), (10, 160)])
def test_redundancy_lps_equal_the_oracle(n, m):
    rng = np.random.default_rng(100 + n)
    A, b = _random_polytope(rng, n, m, m // 2)
    s_gpu, st_gpu = core.polytope_redundant(A, b)
    s_ref, st_ref = oracle_core.polytope_redundant(A, b)
    assert np.array_equal(st_gpu, st_ref)
    assert np.array_equal(s_gpu, s_ref)                  # bit-exact, -inf included
    assert np.all(s_gpu[m:m + m // 2] > 0)               # the pushed-out rows are implied by the rest
    assert np.all(s_gpu[-5:] >= -1e-12) and np.all(s_gpu[:5] >= -1e-12)   # a duplicated pair implies each other (slack 0)


def test_minHRep_recovers_the_stored_W_true():
    g = np.load(os.path.join(GOLD, "params_2x2.npz"))
    H = g["W_true_H"]
    rng = np.random.default_rng(5)
    D = rng.normal(size=(300, 2))
    D /= np.linalg.norm(D, axis=1, keepdims=True)
    hV = (g["W_true_V"] @ D.T).max(0)
    extra = np.hstack([D, (hV * np.where(np.arange(300) % 3 == 0, 1.0, 1.3))[:, None]])   # every third one touches a vertex
    allrows = np.vstack([extra[:150], H, 2.5 * H[2:4], extra[150:]])
    P, keep = api.minHRep(api.Polyhedron(A=allrows[:, :-1], b=allrows[:, -1]), return_keep=True)
    assert list(keep) == list(range(150, 156))
    # and a set that does not contain the origin, through the vertex mean
    t = np.array([0.7, -0.3])
    Q = api.minHRep(api.Polyhedron(A=allrows[:, :-1], b=allrows[:, -1] + allrows[:, :-1] @ t, V=g["W_true_V"] + t))
    assert len(Q.b) == 6


def _same_point_set(P, Q, tol):
    P, Q = np.asarray(P), np.asarray(Q)
    if P.shape != Q.shape:
        return False
    d = np.abs(P[:, None, :] - Q[None, :, :]).max(2)
    return bool(np.all(d.min(1) <= tol) and np.all(d.min(0) <= tol))


def test_pipeline_2x2_reproduces_the_stored_parameters(tmp_path):
    g = np.load(os.path.join(GOLD, "params_2x2.npz"))
    stats_g = np.load(os.path.join(GOLD, "stats.npz"))
    P, stats = pipeline.offline_parameters(2, script="2x2", mrpi_mode=_abi.MS_MAX_LITERAL)   # the stored sets' (s, alpha)
    for k in ("A", "B", "F", "G", "K", "Px", "Pc", "F_bar", "Psi", "Phi", "E"):
        assert np.allclose(P[k], g[k].reshape(P[k].shape), rtol=0, atol=1e-9), k      # dlqr etc.
    assert _same_point_set(P["W_true_V"], g["W_true_V"], 1e-12)                        # the Minkowski sum (:166)
    assert P["W_true_H"].shape == g["W_true_H"].shape
    assert _same_point_set(P["W_true_H"], g["W_true_H"], 1e-9)                         # minHRep + MPT3's row scaling
    assert _same_point_set(P["Xi_EV_V"], g["Xi_EV_V"], 1e-12) and _same_point_set(P["Xi_LV_V"], g["Xi_LV_V"], 1e-12)
    assert _same_point_set(P["W_H"], g["W_H"], 1e-12)
    assert abs(stats["alpha_opt"] - stats_g["2x2_alpha_opt"][0]) < 1e-6                # IPOPT accuracy of the stored value
    # the LP's optimum is a face: only the second coordinate of v binds (as tests/test_gpu_api.py notes for a4); the
    # stored v and ours (centre of the face) both keep W_true inside (1 - alpha) v + alpha W
    assert abs(stats["v_opt"][1] - stats_g["2x2_v_opt"][1]) < 1e-6
    Wn = g["W_H"][:, :2] / g["W_H"][:, -1:]                      # rows a'x <= 1
    for v in (stats["v_opt"], stats_g["2x2_v_opt"]):
        assert ((g["W_true_V"] - (1 - stats["alpha_opt"]) * v) @ Wn.T).max() <= stats["alpha_opt"] + 1e-6
    assert np.abs(np.sort(P["W_hat_opt_V"][:, 1]) - np.sort(g["W_hat_opt_V"][:, 1])).max() < 1e-5
    assert abs(np.ptp(P["W_hat_opt_V"][:, 0]) - np.ptp(g["W_hat_opt_V"][:, 0])) < 1e-5
    # tube sets: the stored ones are hulls of sampled points (inner approximations of an epsilon-outer set); ours are
    # H-rep templates of the exact support.  Stored vertices must be inside up to the (s, alpha) convention, and
    # the tightening within 1.5 % (as tests/test_gpu_api.py does for S).
    for name in ("S", "S_true", "S_hat_opt"):
        H = P[name + "_H"]
        viol = (g[name + "_V"] @ H[:, :-1].T - H[:, -1]).max()
        scale = np.abs(g[name + "_V"]).max()
        assert viol <= 0.015 * scale, (name, viol, scale)
        hV = (g[name + "_V"] @ H[:, :-1].T).max(0)
        assert np.all(hV >= 0.97 * H[:, -1]), name                                     # and not much larger than them
    assert np.abs(P["hs"].ravel() - g["hs"].ravel()).max() <= 0.015 * np.abs(g["hs"]).max()
    assert int(P["Nu"]) == 9        # the exact nu_s; the stored 19 is valid but redundant (api.ComputeFeasibleRegion.ComNu_RMPC)
    # feasible regions: stored hull vertices inside the template (1 % slack for the sampled-hull / (s, alpha) difference)
    for name in ("F_N_RMPC", "F_N_Hat_Opt"):
        H = P[name + "_H"]
        viol = (g[name + "_V"] @ H[:, :-1].T - H[:, -1]).max()
        assert viol <= 0.01 * np.abs(g[name + "_V"]).max(), (name, viol)
    # Ps: centred minimum-volume ellipsoid of Omega's vertices; Omega(1 - hs, 9) = Omega(1 - hs, 19) as sets and hs
    # agrees within 1.5 %, so the stored Ps is reproduced to that level
    assert np.allclose(P["Ps"], P["Ps"].T) and np.linalg.eigvalsh(P["Ps"]).min() > 0
    assert np.abs(P["Ps"] - g["Ps"]).max() <= 0.03 * np.abs(g["Ps"]).max()
    # and the .mat the Case scripts would load
    out = str(tmp_path / "parameters_2x2.mat")
    matio.save_parameters(out, P)
    import scipy.io as sio
    M = sio.loadmat(out, squeeze_me=True, struct_as_record=False)["parameters"]
    assert np.allclose(M.opts_UQMPC.K, P["K"].ravel()) and int(M.opts_UQMPC.Nu) == 9
    assert M.opts_feasible_region.S_true.A.shape[1] == 2 and M.F_N_RMPC.shape[1] == 2
    assert M.W_true.V.shape == (6, 2)


def test_pipeline_3x3_runs_and_prunes():
    P, stats = pipeline.offline_parameters(3, n_template=300, n_region=200)
    assert P["W_true_V"].shape[1] == 3 and len(P["W_true_H"]) >= 8
    # every vertex sum lies in W_true and every facet is touched
    d = pipeline.definitions(3)
    U = np.array([[d["min_u_LV"]], [d["max_u_LV"]]]) @ d["B"].T
    S = (d["V_EV"][:, None, None, :] - d["V_LV"][None, :, None, :] - U[None, None, :, :]).reshape(-1, 3)
    H = P["W_true_H"]
    sl = H[:, -1][None, :] - S @ H[:, :-1].T
    assert sl.min() >= -1e-12 and np.all(sl.min(0) <= 1e-12)
    # the committed 3x3 parameter set was built by lbmpc.offline (host numpy): same K, and the tightening of the
    # pruned 300-row template is exact (its rows include F_Com)
    c = np.load(os.path.join(os.path.dirname(core.__file__), "..", "params", "params_3x3.npz"))
    assert np.allclose(P["K"], c["K"], atol=1e-10)
    assert np.allclose(P["hs"].ravel(), c["hs"].ravel(), rtol=0, atol=1e-9)   # F_Com rows are template directions
    assert stats["S_num_constraints"] <= stats["S_template_rows"]


def test_pipeline_4x4_runs_with_a_many_faced_w_true():
    """4x4: W_true (sum of two rotated 4-D cuboids and a segment) has far more than LBMPC_MAX_NW facets; the (s, alpha)
    search takes them as directions.  K and the exact tightening equal the committed parameter set's."""
    P, stats = pipeline.offline_parameters(4, n_template=400, n_region=100)
    assert len(P["W_true_H"]) > 16 and P["W_true_V"].shape[1] == 4
    c = np.load(os.path.join(os.path.dirname(core.__file__), "..", "params", "params_4x4.npz"))
    assert np.allclose(P["K"], c["K"], atol=1e-10)
    assert np.allclose(P["hs"].ravel(), c["hs"].ravel(), rtol=0, atol=1e-9)
    assert int(P["Nu"]) >= 8 and stats["S_num_constraints"] <= stats["S_template_rows"]
