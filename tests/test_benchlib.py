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


def _worst_case_full_line():
    """the round-5 driver-window dictionary (21.5 KB, the one the driver could not parse) with every optional block present and the strings
    a failing leg would add"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_ns_driver_window.json")))
    full["n_gpus"] = 8
    full["per_rank_steps_per_s"] = [1919.845123456] * 8
    full["duals_follow_obstacles"] = {"error": "x" * 200}
    full["obstacle_shard_leg"] = {"amdahl": {"bound_speedup_without_exchange": 1.46, "bound_speedup_with_measured_gather": 1.41, "what": "y" * 300},
                                  "workload": "T=20, N_obs=2000 static seeded polygons, obstacles sharded 8-way (250 slots per rank)", "steps_per_s": 1000.123456,
                                  "ms_per_step": 1.0, "unsharded_one_gpu_steps_per_s": 800.0, "speedup_vs_one_gpu": 1.25, "gather_us_per_iteration": 21.5,
                                  "gathers": 160, "chunk_bytes_per_rank": 300000, "what": "z" * 400}
    for e in full["sizes"].values():
        e["multi_ego_fleet"] = {"egos": 64, "aggregate_steps_per_s": 71982.7, "c_abi_closed_loop": {"ego_steps_per_s": 41234.5, "what": "w" * 300}}
    full["sizes"]["extra_leg_that_failed"] = {"error": "TimeoutExpired(" + "q" * 400 + ")"}
    full["residuals"]["max_resi_dual"] = float("inf")                 # a failed LamMuZ problem makes a residual infinite: must not become `Infinity`
    full["config"]["env_switches"] = {f"RDA_SWITCH_{i}": "1" * 30 for i in range(12)}
    return full


def test_driver_line_is_compact_and_parseable():
    """BENCH_r05.json had parsed = null because the one line was 21.5 KB; the driver's tail holds 8 018 characters"""
    from benchlib import compact
    full = _worst_case_full_line()
    assert len(json.dumps(full)) > 20000
    s = compact.line(full)
    assert len(s) <= 6000 and "\n" not in s and s.startswith("{")
    assert "Infinity" not in s and "NaN" not in s
    j = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "roofline_secondary", "cpu_baseline", "residuals", "second_window", "sizes"):
        assert k in j, k
    assert "dropped_for_length" not in j
    assert j["value"] == 1919.85 and j["roofline"]["kernel"] == "k_su<20>" and j["roofline"]["bound"] == "hbm" and j["roofline"]["peak"] == 8000.0
    assert set(("achieved", "frac", "traffic", "unit")) <= set(j["roofline"]) and set(("value", "unit", "cores", "kind", "sample")) <= set(j["cpu_baseline"])
    assert len(j["config"]["protocol"]) <= 200 and "workload" in j["config"] and "model" not in j["config"]
    for name in ("n20_T20", "n2000_T20", "c4_dynamic_obs_n200_T30_moving", "c5_shape_n100_T25_fleet64"):
        e = j["sizes"][name]
        assert e["value"] > 0 and e["roofline"]["frac"] > 0 and e["cpu_baseline"]["value"] > 0 and e["fleet"]["closed_loop_ego_steps_per_s"] > 0
    assert j["obstacle_shard_leg"]["amdahl_bound"] == 1.41 and len(j["per_rank_steps_per_s"]) == 8
    # a pathological input still ends in a line the driver can parse: optional blocks are shed, the contract keys never
    full["sizes"] = {f"leg_{i}": dict(full["sizes"]["n20_T20"]) for i in range(40)}
    s = compact.line(full)
    j = json.loads(s)
    assert len(s) <= 6000 and "sizes" in j["dropped_for_length"] and "roofline" in j and "cpu_baseline" in j and "value" in j


def test_bench_prints_exactly_one_stdout_line(capsys, tmp_path, monkeypatch):
    from benchlib import compact
    monkeypatch.setattr(compact, "ROOT", str(tmp_path))
    compact.emit(_worst_case_full_line())
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and json.loads(out)["value"] == 1919.85
    detail = json.load(open(tmp_path / "gpurun_out" / "bench_detail.json"))
    assert "what" in detail["fixed_slot_binding"]
    compact.emit(_worst_case_full_line(), None, detail_to_stdout=True)            # a `sizes` sub-run: the parent reads the DETAIL line
    lines = capsys.readouterr().out.splitlines()
    assert len(lines) == 2 and lines[0].startswith("DETAIL {") and json.loads(lines[0][7:])["value"] == 1919.845 and lines[1].startswith("{")
