#!/usr/bin/env python
"""gpurun_out/prof_<tag>/ (written on the GPU box by tools/profile_round.sh) -> the committed summaries under profiles/:
r<NN>_kernel_stats.csv, r<NN>_pmc_fetch_write.txt, r<NN>_issue_counters.txt, r<NN>_bench*.json and profiles/traffic.json
(HBM bytes per EXECUTED launch of the two solver kernels, read by bench.py for `roofline.traffic`)."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
for name in os.listdir(src):
    if name.endswith((".csv", ".txt", ".json")) and os.path.getsize(os.path.join(src, name)) > 0:
        shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))

# HBM traffic per EXECUTED launch: rocprofv3 lists the counter per dispatch; tools/profile_round.sh separates the dispatches that
# ran (moved more than 0.4 of the largest one) from those queued behind the early-stop flag, which move nothing
pm = open(os.path.join(src, "pmc_fetch_write.txt")).read()
detail, out = {}, {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for key, pat in (("k_su", r"k_su<\d+>"), ("k_lammuz", r"k_lammuz\w*")):
        m = re.search(rf"{counter} (?:void )?({pat}): dispatches (\d+) total ([\d.]+) per-dispatch [\d.]+ executed (\d+) per-executed ([\d.]+)", pm)
        if not m:
            continue
        d = detail.setdefault(m.group(1), {})
        d["dispatches"], d["executed_" + counter.lower()] = int(m.group(2)), int(m.group(4))
        d[counter.lower() + "_kb_total"] = float(m.group(3))
        d[counter.lower() + "_bytes_per_executed_launch"] = round(float(m.group(5)) * 1024)
        out[key] = out.get(key, 0) + d[counter.lower() + "_bytes_per_executed_launch"]
nj = json.loads(open(os.path.join(src, "bench_pmc_FETCH_SIZE.json")).read())
m = re.search(r"T=(\d+), N_obs=(\d+)", nj["metric"])
traffic = {
    "_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each with --kernel-trace only (tools/profile_round.sh). rocprofv3 reports KB; "
                "bytes per EXECUTED launch = mean over the dispatches that moved more than 0.4 of the largest dispatch of that kernel (launches queued "
                "behind the early-stop flag move no data). The first su-problem of a tracked tick is the kernel k_su_tracked<T> (listed in the pmc file, same solve). The gfx950 x2 FETCH_SIZE correction of MI355X_MICROARCH.md applies to wide (16 B/lane) "
                "streaming reads only; these kernels read 8 B/lane, so the raw value is kept (uncalibrated for this width). Working set << L2, "
                "Infinity-Cache hits are counted by these counters.",
    "source": f"profiles/{tag}_pmc_fetch_write.txt",
    "workload": {"n_obs": int(m.group(2)), "horizon": int(m.group(1))},
    "k_lammuz": out.get("k_lammuz"), "k_su": out.get("k_su"), "detail": detail,
}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
