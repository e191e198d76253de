"""-m gpu : the randomised soak as a TEST (`tools/soak.py` found both solver cycles of round 3 and the steps beyond the
stated tolerance, but the driver never ran it).  24 seeded random scenes x 50 closed-loop steps in the default mode, 12 x 50 in the
interior-point LamMuZ mode (`lmz_central = 1e-3`): random kinematics (acker / diff / omni), horizons 10-25, 8-60 polygon obstacles (half
of the scenes moving) + circles, iter_num 2-4, padded / truncated slot counts - tests/soak_lib.py.  The HIP path runs with its defaults;
the checker is the COLD oracle (every su-problem from a cold interior-point start, like ECOS in the reference: no start heuristic is
shared with the kernel); the oracle is re-synchronised to the GPU's state after every step, so each step is an independent sample.

ONE stated tolerance (tests/helpers.py TOL_U = 1e-6 since round 6 - both sides land their su solve on its vertex; rounds 3-5: 5e-4) - asserted
here, in tests/test_gpu_baseline_sizes.py and quoted by bench.py:
    |u_gpu - u_oracle| <= TOL_U on the applied control in the solver's own coordinates, on every step whose ADMM iteration counts agree;
    zero failed su-solves; at most MAX_FLIPS_PER_1000 steps per 1000 on which the two sides stop one ADMM iteration apart (a residual
    within the solver tolerance of `iter_threshold`), each bounded by TOL_U_FLIP.
The same difference expressed as body rates (soak_lib.body_rates) is printed, not asserted.  What the long soaks of round 6 found beyond these seeds
(tools/soak.py, 290 k steps): largest raw difference 3.3e-7 - and FOUR steps (of 64 k in the --exotic flavour) at 4.8e-6 .. 1.2e-4: the steering angle
of an Ackermann robot at |v| <= 0.13 m/s on a solve whose landings were all refused (the fallback is the interior point; the su-problem is nearly singular in that direction).
"""
import os

import pytest

from helpers import TOL_U, TOL_U_FLIP, MAX_FLIPS_PER_1000

pytestmark = pytest.mark.gpu


def _check(out, what):
    print(f"\n{what}: {out['steps']} steps, max |du| body {out['worst_body']:.2e} (raw {out['worst_raw']:.2e}, whole horizon body "
          f"{out['worst_hor_body']:.2e}); steps with raw |du| > 1e-5: {out['over_raw']}; iteration-count mismatches {out['iter_mismatch']}; "
          f"failed su-solves {out['failed']}; interior-point iterations gpu {out['ipm_gpu']} / cold oracle {out['ipm_cpu']}")
    assert out["steps"] >= 500
    assert out["failed"] == 0, "a su-solve failed"
    assert out["worst_raw"] <= TOL_U, out["worst_raw"]
    assert out["iter_mismatch"] * 1000 <= MAX_FLIPS_PER_1000 * out["steps"], (out["iter_mismatch"], out["steps"])
    for r in out["flips"]:
        assert r["du_raw"] <= TOL_U_FLIP, r


def test_mini_soak_default_mode():
    from soak_lib import run_soak
    dump = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "soak_outliers")
    out = run_soak(scenes=24, steps=50, seed=11, cold_oracle=True, dump_dir=dump if os.environ.get("RDA_SOAK_DUMP") else "")
    _check(out, "mini-soak, default mode vs cold oracle")


def test_mini_soak_interior_point_lammuz_mode():
    from soak_lib import run_soak
    out = run_soak(scenes=12, steps=50, seed=12, cold_oracle=True, lmz_central=1e-3)
    _check(out, "mini-soak, lmz_central = 1e-3 vs cold oracle")


def test_mini_soak_circle_obstacles():
    """two of three obstacles are CIRCLES (norm2 obstacle cone: what the reference's dynamic_obs example is made of); the remembered supports of circle rows
    (`lmz::warm_circle`, round 5) and the enumeration behind them against the cold oracle.  VERDICT r05 #6c: the 45 040-step circle soak was a tool run only."""
    from soak_lib import run_soak
    out = run_soak(scenes=12, steps=50, seed=101, cold_oracle=True, circles=True)
    _check(out, "mini-soak, circle obstacles vs cold oracle")


def test_a_refused_landing_stays_within_the_interior_point_tolerance():
    """The exception to TOL_U, pinned (tests/helpers.py): the `--exotic --robots` soak scene on which the long soaks of round 6 found the first step outside
    1e-6 - seed 23000, scene 52: Ackermann robot, T = 40, 56 moving obstacles, `accelerated=False`; at step 46 the robot creeps at v = -0.004 m/s, every landing of one
    su-solve is refused on the GPU side, the solve returns its fallback (the interior point at su_tol) and the steering angle - a direction the su-problem is nearly
    singular in there - is 7.9e-5 from the cold oracle's landed answer.  What holds for such a solve: TOL_U_IP on the raw control, TOL_U on what the robot does with it
    (linear velocity, yaw rate), no failed solve, the same ADMM iteration counts.  (100 steps: the scene's draws depend on the soak's step count.)"""
    from helpers import TOL_U_IP
    from soak_lib import run_soak
    out = run_soak(scenes=53, steps=100, seed=23000, cold_oracle=True, only=52, exotic=True, robots=True, dump_tol=TOL_U, log=lambda *_: None)
    print(f"\nscene 52 of seed 23000: {out['steps']} steps, max |du| raw {out['worst_raw']:.2e} body {out['worst_body']:.2e}, steps with raw |du| > {TOL_U:g}: {out['over_raw']}")
    assert out["steps"] == 100 and out["failed"] == 0 and out["iter_mismatch"] == 0
    assert out["worst_raw"] <= TOL_U_IP, out["worst_raw"]
    assert out["worst_body"] <= TOL_U, out["worst_body"]
    assert out["over_raw"] <= 2, out["over_raw"]              # (7.9e-5 on one step before the fallback ran tighter than su_tol - SU_LAND_FALLBACK -, 9.1e-8 since; the bound is on the class)
