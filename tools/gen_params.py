#!/usr/bin/env python3
"""Generates the committed 3x3 / 4x4 parameter sets (learning-based-mpc_amd/params/params_{3x3,4x4}.npz) -- the
reference's parameters_3x3.mat / parameters_4x4.mat are missing from its checkout -- through lbmpc/pipeline.py, i.e.
offline_parameters_computation_nxn.m on the HIP path (needs the GPU: run under gpurun, then copy
gpurun_out/params/*.npz into learning-based-mpc_amd/params/).  The tube set S keeps the reference's facet counts
(stats_3x3.mat / stats_4x4.mat: S_num_constraints = 795 / 1644) with EVERY row a facet: the template is made large
enough that at least that many rows survive minHRep, then that many are kept.  Nu is floored at 47 (= 5.875 x 8, the
value the reference ran 3x3 with, Case_5...:5), so the QP has the reference's row count.
  python tools/gen_params.py [3] [4]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "learning-based-mpc_amd"))
import numpy as np

from lbmpc import api, core, pipeline

PLAN = {3: dict(rows=795, templates=(130000, 180000, 250000)), 4: dict(rows=1644, templates=(6000, 8000, 12000))}


def main():
    core.init(0)
    out_dir = os.path.join(ROOT, "gpurun_out", "params")
    os.makedirs(out_dir, exist_ok=True)
    for nx in [int(a) for a in sys.argv[1:]] or [3, 4]:
        for n in PLAN[nx]["templates"]:
            t0 = time.perf_counter()
            try:
                P, st = pipeline.offline_parameters(nx, n_template_S=n, rows_S=PLAN[nx]["rows"], nu_floor=47, log=print)
            except RuntimeError as e:
                print(f"nx={nx}: template {n}: {e}", flush=True)
                continue
            S = api.Polyhedron(A=P["S_H"][:, :-1], b=P["S_H"][:, -1])
            _, keep = api.minHRep(S, return_keep=True)
            print(f"nx={nx}: template {n} -> S rows {len(S.b)}, irredundant {len(keep)}, Nu {int(P['Nu'])} (exact {int(P['Nu_exact'])}), "
                  f"s {int(P['mrpi_s'])}, hs {np.ravel(P['hs']).round(5)}, {time.perf_counter() - t0:.1f} s", flush=True)
            assert len(keep) == len(S.b) == PLAN[nx]["rows"]
            P["template_rows_S"] = np.int64(n)
            np.savez_compressed(os.path.join(out_dir, f"params_{nx}x{nx}.npz"), **{k: v for k, v in P.items() if v is not None})
            break
        else:
            sys.exit(f"nx={nx}: no template size gave {PLAN[nx]['rows']} facets")


if __name__ == "__main__":
    main()
