"""Load the committed golden fixtures (tests/golden/*.npz) into the shapes the ABI wants."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def params_dict(name="params_2x2.npz"):
    p = npz(name)
    d = {k: p[k] for k in p.files}
    d["S_A"], d["S_b"] = p["S_H"][:, :-1], p["S_H"][:, -1]
    d["W_A"], d["W_b"] = p["W_H"][:, :-1], p["W_H"][:, -1]
    return d


def sampler_args(d):
    nx, nu = int(d["nx"]), int(d["nu"])
    return dict(nx=nx, nu=nu, EV_A=d["Xi_EV_H"][:, :-1], EV_b=d["Xi_EV_H"][:, -1], EV_V=d["Xi_EV_V"],
                LV_A=d["Xi_LV_H"][:, :-1], LV_b=d["Xi_LV_H"][:, -1], LV_V=d["Xi_LV_V"], B=d["B"],
                min_u=d["min_u_LV"], max_u=d["max_u_LV"])
