"""One-off (round 5): phase cycles of the LamMuZ enumeration (lmz::solve_wave) on one sub-problem - needs a -DRDA_LMZ_PROF build:
RDA_HIP_SO=tools/_bin/librda_hip_lmzprof.so python tools/experiments/lmz_enum_prof.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as hp  # noqa: E402
from rda_planner_amd._lib import hip_api  # noqa: E402

hip = hip_api()
for seed in range(4):
    inp = hp.lammuz_batch_inputs(np.random.default_rng(seed), 64, E=4, circles=0.0)
    hp.hip_lammuz_batch(hip, inp)
