"""Round 6: the C5 fleet closed loop as G fleets of 64 / G members, every fleet ticked by its own host thread (benchlib.legs.fleet_closed_loop(groups=G)).

    python tools/experiments/fleet_groups.py [--groups 1 2 4 8]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from benchlib import legs                      # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--egos", type=int, default=64)
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--n-obs", "100", "--horizon", "25", "--steps", "30", "--warmup", "8", "--no-cpu-baseline", "--size-leg"]
    ctx = bench.setup(bench.parse_args())
    for g in a.groups:
        for rep in range(2):
            r = legs.fleet_closed_loop(ctx, a.egos, groups=g, check=(0, a.egos - 1) if rep == 0 else ())
            print(f"groups {g}: {r['ego_steps_per_s']:9.1f} ego-steps/s, {r['ms_per_fleet_tick']:.3f} ms per tick of {a.egos}, ADMM {r['mean_admm_iters']}, "
                  f"ipm {r['su_interior_point_iters_per_ego_step']}, vs solo {r['max_du_vs_solo_closed_loop']:.2e}", flush=True)
