"""Round 6: the four soak steps outside the stated tolerance (refused landings -> the interior point at su_tol, in a nearly singular direction) once more with a
TIGHTER su_tol on both sides (rda_opts::su_tol / orc_set_su_tol): does the difference go away, and does any solve fail at the tighter stop?

    python tools/experiments/soak_outliers_su_tol.py [--scale 1e-2]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from soak_lib import run_soak                                  # noqa: E402
from oracle.oracle_backend import api as orc_api              # noqa: E402
from rda_planner_amd.rda_solver import hip_options             # noqa: E402

CASES = [(23000, 52, 46), (31000, 19, 43), (31000, 185, 81), (31000, 435, 54)]
BASE = (1e-9, 1e-10, 1e-11)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, nargs="+", default=[1.0, 1e-2])
    a = ap.parse_args()
    lib = orc_api().lib
    lib.orc_set_su_tol.argtypes = [C.c_double] * 3
    for sc_ in a.scale:
        tol = tuple(t * sc_ for t in BASE)
        lib.orc_set_su_tol(*tol)
        for seed, scene, step in CASES:
            out = run_soak(scenes=scene + 1, steps=100, seed=seed, cold_oracle=True, only=scene, exotic=True, robots=True, dump_tol=1e-6,
                           hip_kw={"hip_opts": hip_options(su_tol=tol)}, log=lambda *_: None)
            worst = max([r["du_raw"] for r in out["outliers"]] + [0.0])
            print(f"su_tol x {sc_:g}: seed {seed} scene {scene}: {out['steps']} steps, max raw {out['worst_raw']:.2e} body {out['worst_body']:.2e}, steps > 1e-6: {out['over_raw']} "
                  f"(worst {worst:.2e}), failed su-solves {out['failed']}, iteration-count mismatches {out['iter_mismatch']}, ipm gpu {out['ipm_gpu']} / oracle {out['ipm_cpu']}", flush=True)
