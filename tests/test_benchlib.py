"""-m "not gpu": the pure pieces of bench.py (benchlib/): roofline arithmetic, the residual summary of the driver line, the CLI surface the driver
and tools/profile_round.sh depend on.  (The legs themselves need a GPU: every `gpurun` bench call exercises them.)"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import closed_loop, roofline  # noqa: E402
from benchlib.context import stats  # noqa: E402


def test_roofline_separates_executed_from_skipped_launches():
    # 6 executed launches of 100 us, 4 launches behind the early-stop flag of 3 us: the n_exec longest are the executed ones
    ms = np.array([0.1] * 6 + [0.003] * 4)
    r = roofline.roof("k", ms, 1_152_000, 6)
    assert r["launches"] == 6 and r["skipped_launches"] == 4 and abs(r["avg_launch_us"] - 100.0) < 1e-9
    assert abs(r["achieved"] - 1_152_000 / 100e-6 / 1e9) < 1e-3 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-6
    assert abs(r["skipped_avg_us"] - 3.0) < 1e-9
    assert roofline.unit_bytes(4, 4) == 288
    assert roofline.su_bytes(20, 200) == 20384                   # 32 T ceil(N/8) + trajectory / multipliers / pose


def test_issue_figures_come_from_the_committed_profile_of_the_same_workload(tmp_path):
    prof = tmp_path / "profiles"; prof.mkdir()
    (prof / "issue.json").write_text(json.dumps({"source": "x", "workloads": {"ns": {"n_obs": 200, "horizon": 20, "moving": False, "per": "executed", "kernels": {
        "k_su<20>": {"SQ_INSTS_VALU": 60000, "SQ_INSTS_SALU": 20000, "SQ_INSTS_LDS": 15000, "SQ_INSTS_VMEM_RD": 1000, "SQ_WAVE_CYCLES": 250000, "SQ_WAVES": 4,
                     "SQ_ACTIVE_INST_LDS": 15000, "SQ_LDS_BANK_CONFLICT": 12600, "SQ_INSTS_VALU_FMA_F64": 10000}}}}}))
    r_su = roofline.roof("k_su<20>", np.full(8, 0.1), 20384, 8); r_lm = roofline.roof("k_lammuz_rows", np.full(8, 0.03), 1152000, 8)
    roofline.attach_issue(str(tmp_path), r_su, r_lm, 200, 20, False)
    assert abs(r_su["ipc_per_wave"] - 96000 / 1e6) < 1e-4 and r_su["lds_conflict_ratio"] == 0.84 and r_su["issue_per"].startswith("executed")
    assert "ipc_per_wave" not in r_lm
    roofline.attach_issue(str(tmp_path), r_lm, r_lm, 2000, 20, False)       # another workload: nothing attached
    assert "ipc_per_wave" not in r_lm


def test_residual_summary_says_where_the_admm_ends():
    info = np.array([[0.15, 0.01, 5], [0.29, 0.0, 20], [np.inf, 0.0, 18], [1.1, 0.3, 24]], float)
    s = closed_loop.residual_summary(info, 0.2)
    assert s["steps"] == 4 and s["steps_below_threshold"] == 1 and s["max_resi_dual"] == 1.1 and abs(s["su_interior_point_iters_per_step"] - 16.75) < 1e-9
    assert closed_loop.residual_summary(None, 0.2) is None
    assert stats(20, 0.01, [0.0005] * 20, [4] * 20, what="x") == {"steps_per_s": 2000.0, "median_ms_per_step": 0.5, "mean_admm_iters": 4.0, "what": "x"}


def test_bench_cli_keeps_the_flags_the_driver_and_the_profile_script_use():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--only-headline", "--n-obs", "--horizon", "--moving", "--no-sizes", "--no-cpu-baseline", "--mode"):
        assert flag in out.stdout, flag
    text = open(os.path.join(ROOT, "tools", "profile_round.sh")).read()
    assert "bench.py --only-headline" in text
