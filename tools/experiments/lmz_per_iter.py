"""One-off (round 5): hipEvent time of the LamMuZ / su launches of the headline loop by ADMM iteration index (4 per step)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "--only-headline", "--steps", "100", "--warmup", "10"]
import bench  # noqa: E402
from benchlib import closed_loop  # noqa: E402

ctx = bench.setup(bench.parse_args())
tm = closed_loop.run(ctx, per_tick_scene=False, ordered=True, timing=True, compare=False)
for name in ("k_lammuz", "k_su"):
    v = np.asarray(tm.kernel_ms[name]) * 1e3
    n = (len(v) // 4) * 4
    m = v[:n].reshape(-1, 4)
    print(name, "launches", len(v), "mean by iteration index", np.round(m.mean(axis=0), 1), "median", np.round(np.median(m, axis=0), 1), "min", np.round(m.min(axis=0), 1), "max", np.round(m.max(axis=0), 1))
    if name == "k_lammuz":
        print("   histogram (us):", np.histogram(v, bins=[0, 18, 20, 22, 25, 28, 31, 35, 40, 60])[0])
