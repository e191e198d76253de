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

# HBM traffic per executed launch: PMC totals over ALL dispatches of the pass / the executed dispatches of that pass
pm = open(os.path.join(src, "pmc_fetch_write.txt")).read()
detail, out = {}, {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    bj = json.loads(open(os.path.join(src, f"bench_pmc_{counter}.json")).read())
    for key, roof in (("k_su", bj["roofline"]), ("k_lammuz", bj["roofline_secondary"])):
        if not roof["kernel"].startswith(key):
            roof = bj["roofline_secondary"] if roof is bj["roofline"] else bj["roofline"]
        frac = roof["launches"] / max(roof["launches"] + roof["skipped_launches"], 1)
        m = re.search(rf"{counter} (?:void )?{re.escape(roof['kernel'])}: dispatches (\d+) total ([\d.]+)", pm)
        if not m:
            continue
        disp, total_kb = int(m.group(1)), float(m.group(2))
        executed = disp * frac
        d = detail.setdefault(roof["kernel"], {})
        d[counter.lower() + "_kb_total"], d["dispatches"], d["executed_fraction"] = total_kb, disp, round(frac, 4)
        d[counter.lower() + "_bytes_per_executed_launch"] = round(total_kb * 1024 / executed)
        out[key] = out.get(key, 0) + d[counter.lower() + "_bytes_per_executed_launch"]
nj = json.loads(open(os.path.join(src, "bench_pmc_FETCH_SIZE.json")).read())
m = re.search(r"T=(\d+), N_obs=(\d+)", nj["metric"])
traffic = {
    "_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each with --kernel-trace only (tools/profile_round.sh). rocprofv3 reports KB; "
                "bytes per EXECUTED launch = counter total over all dispatches of the pass / (dispatches x executed fraction): launches queued "
                "behind the early-stop flag move no data. The gfx950 x2 FETCH_SIZE correction of MI355X_MICROARCH.md applies to wide (16 B/lane) "
                "streaming reads only; these kernels read 8 B/lane, so the raw value is kept (uncalibrated for this width). Working set << L2, "
                "Infinity-Cache hits are counted by these counters.",
    "source": f"profiles/{tag}_pmc_fetch_write.txt",
    "workload": {"n_obs": int(m.group(2)), "horizon": int(m.group(1))},
    "k_lammuz": out.get("k_lammuz"), "k_su": out.get("k_su"), "detail": detail,
}
json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
