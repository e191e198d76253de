#!/bin/bash
# round 6: where a fleet tick of the C-ABI closed loop goes - rocprofv3 kernel trace of the C5 size leg, cut into ticks at the k_track_fleet launches
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/fleet_trace; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --output-format csv -d $D -o f -- python bench.py --size-leg --cpu-threads 16 --n-obs 100 --horizon 25 --steps 30 --warmup 8 --fleet-egos ${EGOS:-64} --no-cpu-baseline > $D/run.log 2>&1
F=$(find $D -name '*kernel_trace.csv' | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
ticks = [i for i, r in enumerate(rows) if r[2].startswith("k_track_fleet")]
print(len(rows), "kernels,", len(ticks), "fleet ticks")
out = []
for a, b in zip(ticks[:-1], ticks[1:]):
    seg = rows[a:b]
    wall = (rows[b][0] - rows[a][0]) / 1e3
    if wall > 20000: continue
    busy = sum(e - s for s, e, _ in seg) / 1e3
    # union of busy intervals (kernels may overlap)
    cur_e, uni = 0, 0
    for s, e, _ in seg:
        if s > cur_e: uni += e - s; cur_e = e
        elif e > cur_e: uni += e - cur_e; cur_e = e
    by = collections.Counter()
    for s, e, n in seg: by[n] += (e - s) / 1e3
    out.append((wall, busy, uni / 1e3, by, len(seg)))
out = out[len(out) // 3:]        # the closed-loop ticks come last; skip the warm-up third
if out:
    n = len(out)
    print(f"{n} ticks: wall {sum(o[0] for o in out) / n:.0f} us, kernel time summed {sum(o[1] for o in out) / n:.0f} us, union {sum(o[2] for o in out) / n:.0f} us, launches per tick {sum(o[4] for o in out) / n:.1f}")
    tot = collections.Counter()
    for o in out:
        for k, v in o[3].items(): tot[k] += v / n
    for k, v in tot.most_common(14): print(f"   {k[:70]:70s} {v:9.1f} us per tick")
PY
rm -rf $D
