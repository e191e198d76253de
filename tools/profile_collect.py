#!/usr/bin/env python
"""gpurun_out/prof_<tag>/ (written on the GPU box by tools/profile_round.sh) -> the committed summaries under profiles/:
<tag>_<cfg>_kernel_stats.csv, <tag>_pmc_fetch_write.txt, <tag>_<cfg>_issue_counters.txt, <tag>_bench_<cfg>.json, <tag>_suprof_<cfg>.txt and
profiles/traffic.json (HBM bytes per EXECUTED launch of the solver kernels PER CONFIGURATION, read by bench.py for `roofline.traffic`)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
for name in os.listdir(src):
    f = os.path.join(src, name)
    if os.path.isfile(f) and name.endswith((".csv", ".txt", ".json")) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(dst, f"{tag}_{name}"))

# HBM traffic per EXECUTED launch: rocprofv3 lists the counter per dispatch; tools/profile_round.sh separates the dispatches that
# ran (moved more than 0.4 of the largest one of their kernel) from those queued behind the early-stop flag, which move nothing
pm = open(os.path.join(src, "pmc_fetch_write.txt")).read()
workloads = {}
for cfg in ("ns", "n20", "n2000", "c4", "ip"):
    try:
        nj = json.loads(open(os.path.join(src, f"{cfg}_bench_pmc_FETCH_SIZE.json")).read())
    except Exception:
        continue
    m = re.search(r"T=(\d+), N_obs=(\d+)", nj["metric"])
    detail = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        found = {}
        for mm in re.finditer(rf"^{cfg} {counter} (?:void )?([\w<>]+): dispatches (\d+) total ([\d.]+) per-dispatch [\d.]+ executed (\d+) per-executed ([\d.]+)(?: steady-total ([\d.]+))?", pm, re.M):
            k = mm.group(1)
            if not (k.startswith("k_su") or k.startswith("k_lammuz") or k.startswith("k_lmz")):
                continue
            found[k] = mm
            d = detail.setdefault(k, {})
            d["dispatches"] = int(mm.group(2))
            d[counter.lower() + "_bytes_per_executed_launch"] = round(float(mm.group(5)) * 1024)
        # the work-list kernel: steady-state bytes per executed ITERATION (= per executed launch of the common-path kernel), see profile_round.sh
        if "k_lammuz_enum" in found and "k_lammuz_rows_fast" in found and found["k_lammuz_enum"].group(6):
            detail["k_lammuz_enum"][counter.lower() + "_bytes_per_executed_launch"] = round(float(found["k_lammuz_enum"].group(6)) * 1024 / max(1, int(found["k_lammuz_rows_fast"].group(4))))
    # the LamMuZ step of an iteration: one kernel on small grids, the common-path + work-list + finalize kernels on dense ones (summed)
    lm = sum(sum(v for kk, v in d.items() if kk.endswith("_launch")) for k, d in detail.items() if k.startswith(("k_lammuz", "k_lmz")) and k != "k_lmz_finalize_all")
    su = sum(sum(v for kk, v in d.items() if kk.endswith("_launch")) for k, d in detail.items() if re.fullmatch(r"k_su<\d+>", k))
    workloads[cfg] = {"n_obs": int(m.group(2)), "horizon": int(m.group(1)), "moving": "moving" in nj["config"]["workload"],
                      "lmz_mode": 1 if cfg == "ip" else 0, "k_lammuz": lm or None, "k_su": su or None, "detail": detail}
traffic = {
    "_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each with --kernel-trace only (tools/profile_round.sh), per BASELINE configuration. "
                "rocprofv3 reports KB; bytes per EXECUTED launch = mean over the dispatches that moved more than 0.4 of the largest dispatch of that kernel "
                "(launches queued behind the early-stop flag move no data). k_lammuz = all kernels of the LamMuZ step of one ADMM iteration summed. The first "
                "su-problem of a tracked tick is the kernel k_su_tracked<T> (listed in `detail`, same solve). The gfx950 x2 FETCH_SIZE correction of "
                "MI355X_MICROARCH.md applies to wide (16 B/lane) streaming reads only; these kernels read 8 B/lane, so the raw value is kept (uncalibrated "
                "for this width). Working set << L2, Infinity-Cache hits are counted by these counters.",
    "source": f"profiles/{tag}_pmc_fetch_write.txt",
    "workloads": workloads,
}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)

# SQ issue counters per dispatch (averaged over executed and skipped launches alike) -> profiles/issue.json: bench.py derives
# roofline.ipc_per_wave / fp64_frac / serial_cycles from RATIOS of these (the figures that move with kernel quality)
issue = {}
for cfg in ("ns", "n20", "n2000", "c4", "ip"):
    f = os.path.join(src, f"{cfg}_issue_counters.txt")
    if not os.path.exists(f) or cfg not in workloads:
        continue
    kern = {}
    # per EXECUTED dispatch since round 5 (tools/profile_round.sh separates them like the FETCH / WRITE passes do); older files: per dispatch
    text = open(f).read()
    executed = " per-executed " in text
    # a loop whose every step ran all iter_num ADMM iterations has NO skipped launch: every dispatch is an executed one (the headline protocol: 4.0 -
    # there the 0.4-of-a-large-one rule of profile_round.sh would throw the cheap su-solves out, an su launch takes 25 ... 260 us)
    all_executed = False
    try:
        bj = json.loads(open(os.path.join(src, f"{cfg}_bench_under_rocprof.json")).read())
        all_executed = abs(float(bj["mean_admm_iters"]) - 4.0) < 1e-9
    except Exception:
        pass
    executed = executed or all_executed
    for mm in re.finditer(r"^(?:void )?([\w<>]+)\s+(SQ_\w+)\s+dispatches\s+(\d+) per-dispatch\s+([\d.]+)(?: executed\s+(\d+) per-executed\s+([\d.]+))?", text, re.M):
        d = kern.setdefault(mm.group(1), {"dispatches": int(mm.group(3))})
        d[mm.group(2)] = float(mm.group(6)) if (mm.group(6) and not all_executed) else float(mm.group(4))
        if mm.group(5):
            d["executed"] = int(mm.group(3)) if all_executed else int(mm.group(5))
    if kern:
        issue[cfg] = {k: workloads[cfg][k] for k in ("n_obs", "horizon", "moving", "lmz_mode")}
        issue[cfg]["kernels"] = kern
        issue[cfg]["per"] = "executed" if executed else "dispatch"
if issue:
    json.dump({"_comment": "rocprofv3 --pmc SQ_* (four passes per configuration, tools/profile_round.sh; every process is bench.py --only-headline: the headline loop of "
                           "that configuration and nothing else), wave-instructions / cycles per EXECUTED dispatch (`per`: executed - a dispatch that counted more than "
                           "0.4 of a large one of its kernel; launches queued behind the early-stop flag are left out).",
               "source": f"profiles/{tag}_<cfg>_issue_counters.txt", "workloads": issue}, open(os.path.join(dst, "issue.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "detail"} for k, v in workloads.items()}, indent=1))
