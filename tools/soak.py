"""Random closed-loop soak (CLI of tests/soak_lib.py): the HIP path against the CPU oracle, step by step from the SAME solver state.
Test infrastructure, like everything that touches oracle/.  tests/test_gpu_soak.py is the short form the driver runs.
    gpurun -- 'python tools/soak.py --scenes 24 --steps 120'
    gpurun -- 'python tools/soak.py --scenes 96 --steps 100 --seed 2 --cold --dump-dir gpurun_out/soak_outliers'
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from soak_lib import run_soak                                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=12)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tol", type=float, default=1e-5, help="steps whose raw |du| exceeds this are listed (and dumped with --dump-dir)")
    ap.add_argument("--only", type=int, default=-1, help="run this scene only (the random draws of the others are still made)")
    ap.add_argument("--cold", action="store_true", help="cold oracle (orc_set_su_warm(0,0,0)): the independent checker; default = the warm oracle that mirrors the kernel's first start rules")
    ap.add_argument("--lmz-central", type=float, default=0.0, help="interior-point LamMuZ mode on both sides (central-path point at this barrier parameter, e.g. 1e-3)")
    ap.add_argument("--dump", default="", help="record ALL the oracle's su-problems (same state as the GPU's: re-synchronised every step) for tools/su_replay.py")
    ap.add_argument("--dump-dir", default="", help="record the su-problems and both answers of the listed steps only (one .bin + .npz per step)")
    ap.add_argument("--so", default="", help="another build of librda_hip.so (A/B against an older commit)")
    ap.add_argument("--su-tol-early", action="store_true", help="GPU side with the opt-in rda_opts::su_tol_early = (1e-6, 1e-7, 1e-8); the oracle keeps su_tol: how far the loose class moves the controls")
    ap.add_argument("--large", action="store_true", help="scenes of the BASELINE regime: T in {20, 25, 30}, 100 - 420 obstacles (the oracle needs ~0.1 s per step there)")
    ap.add_argument("--exotic", action="store_true", help="what the examples do not use: max_edge_num 5 - 8, circle robot, accelerated=False, T in {5, 12, 40}, obstacle_order=False, other weights")
    ap.add_argument("--robots", action="store_true", help="with --exotic: convex robot bodies with 3 / 5 / 6 / 8 edges (R up to 8; E + R + 1 > 16 runs the one-row-per-wave LamMuZ kernel)")
    ap.add_argument("--circles", action="store_true", help="two of three obstacles are circles (norm2 cone: the reference's dynamic_obs example)")
    ap.add_argument("--tight", action="store_true", help="half the clearance between path and obstacles: blocked lanes, collisions, su-problems that are hard or fail")
    ap.add_argument("--hard-off", action="store_true", help="GPU side without rda_opts::su_hard_warm (the start rule of round 4)")
    a = ap.parse_args()
    t0 = time.time()
    hip_kw = None
    if a.su_tol_early or a.hard_off:
        from rda_planner_amd.rda_solver import hip_options
        ch = {}
        if a.su_tol_early:
            ch["su_tol_early"] = (1e-6, 1e-7, 1e-8)
        if a.hard_off:
            ch["su_hard_warm"] = (0.0, 0.0)
        hip_kw = {"hip_opts": hip_options(**ch)}
    out = run_soak(scenes=a.scenes, steps=a.steps, seed=a.seed, lmz_central=a.lmz_central, cold_oracle=a.cold, only=a.only, dump_dir=a.dump_dir,
                   dump_tol=a.tol, su_dump=a.dump, so=a.so, hip_kw=hip_kw, large=a.large, exotic=a.exotic, tight=a.tight, robots=a.robots, circles=a.circles)
    print(f"soak: {out['steps']} steps over {a.scenes} scenes in {time.time() - t0:.0f} s; max |du| raw {out['worst_raw']:.2e} body {out['worst_body']:.2e} "
          f"(whole horizon, body {out['worst_hor_body']:.2e}); control mismatches > {a.tol:g}: {out['over_raw']}; "
          f"iteration-count mismatches: {out['iter_mismatch']}; steps with a failed su-solve: {out['failed']}; "
          f"interior-point iterations gpu {out['ipm_gpu']} / oracle {out['ipm_cpu']}")


if __name__ == "__main__":
    main()
