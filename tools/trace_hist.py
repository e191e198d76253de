"""Per-dispatch durations of the solver kernels from a rocprofv3 --kernel-trace CSV: executed launches (not the ones that find the stop flag)
python tools/trace_hist.py <kernel_trace.csv>"""
import csv
import sys
import collections
import numpy as np

d = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Kernel_Name"].split("(")[0].replace("void ", "")
    d[name].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
for name, v in sorted(d.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    if not (name.startswith("k_su") or name.startswith("k_lammuz") or name.startswith("k_lmz") or name.startswith("k_finish")):
        continue
    dur = np.array([e - s for s, e in v]) / 1000.0
    ex = dur[dur > 0.4 * np.percentile(dur, 98)]
    q = np.percentile(ex, [5, 25, 50, 75, 95])
    print(f"{name:28s} dispatches {len(dur):6d} executed {len(ex):6d}  executed us: mean {ex.mean():7.2f}  p5 {q[0]:7.2f} p25 {q[1]:7.2f} p50 {q[2]:7.2f} p75 {q[3]:7.2f} p95 {q[4]:7.2f}   skipped mean {dur[dur <= 0.4 * np.percentile(dur, 98)].mean() if (dur <= 0.4 * np.percentile(dur, 98)).any() else 0:5.2f}")
# gaps between consecutive dispatches on the device (all kernels, in start order)
allk = sorted((s, e, n) for n, v in d.items() for s, e in v)
gaps = np.array([allk[i + 1][0] - allk[i][1] for i in range(len(allk) - 1)]) / 1000.0
g = gaps[(gaps > -50) & (gaps < 20)]
print(f"gaps between consecutive dispatches (us): median {np.median(g):.2f}  p25 {np.percentile(g, 25):.2f} p75 {np.percentile(g, 75):.2f}")
