"""Parameter sets as plain dicts (the fields of the reference's `parameters.opts_UQMPC` / `opts_Car`,
Polyhedra unpacked to H = [A b] and V arrays)."""
import os

import numpy as np

from . import _abi

PARAMS_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "params")


def load(name_or_path):
    """'3x3' / '4x4' (build-generated, lbmpc/offline.py) or a path to an .npz with the same keys
    (e.g. tests/golden/params_2x2.npz decoded from the reference's parameters_2x2.mat)."""
    path = name_or_path
    if not os.path.exists(path):
        path = os.path.join(PARAMS_DIR, f"params_{name_or_path}.npz")
    p = np.load(path, allow_pickle=False)
    d = {k: p[k] for k in p.files}
    d["S_A"], d["S_b"] = d["S_H"][:, :-1], d["S_H"][:, -1]
    d["W_A"], d["W_b"] = d["W_H"][:, :-1], d["W_H"][:, -1]
    return d


def sampler(d, oriented=False):
    """oriented=True: principal-axes proposal boxes where they are tighter than the reference's axis-aligned ones
    (lbmpc_sampler.EV_U / LV_U): same uniform distribution on Xi, fewer rejections."""
    nx, nu = int(d["nx"]), int(d["nu"])
    return _abi.make_sampler(nx=nx, nu=nu, EV_A=d["Xi_EV_H"][:, :-1], EV_b=d["Xi_EV_H"][:, -1], EV_V=d["Xi_EV_V"],
                             LV_A=d["Xi_LV_H"][:, :-1], LV_b=d["Xi_LV_H"][:, -1], LV_V=d["Xi_LV_V"], B=d["B"],
                             min_u=d["min_u_LV"], max_u=d["max_u_LV"], oriented=oriented)
