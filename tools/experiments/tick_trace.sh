#!/bin/bash
# round 6: GPU time against wall time per MPC tick - rocprofv3 kernel trace of `bench.py --only-headline`, cut into ticks at the k_su_tracked launches
#   bash tools/experiments/tick_trace.sh [bench args ...]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/tick_trace; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --output-format csv -d $D -o f -- python bench.py --only-headline --steps 60 --warmup 10 "$@" > $D/run.log 2>&1
F=$(find $D -name '*kernel_trace.csv' | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
ticks = [i for i, r in enumerate(rows) if "k_su_tracked" in r[2]]
out = []
for a, b in zip(ticks[:-1], ticks[1:]):
    seg = rows[a:b]
    wall = (rows[b][0] - rows[a][0]) / 1e3
    if wall > 5000: continue
    cur_e, uni, gaps = 0, 0, []
    for s, e, n in seg:
        if s > cur_e:
            if cur_e: gaps.append(((s - cur_e) / 1e3, n))
            uni += e - s; cur_e = e
        elif e > cur_e: uni += e - cur_e; cur_e = e
    tail = (rows[b][0] - cur_e) / 1e3
    by = collections.Counter()
    for s, e, n in seg: by[n] += (e - s) / 1e3
    out.append((wall, uni / 1e3, by, len(seg), gaps, tail))
out = out[len(out) // 4:]
n = len(out)
print(f"{n} ticks: wall {sum(o[0] for o in out) / n:.1f} us, GPU busy (union of kernels) {sum(o[1] for o in out) / n:.1f} us, launches per tick {sum(o[3] for o in out) / n:.1f}, idle between the tick's last kernel and the next tick's first {sum(o[5] for o in out) / n:.1f} us")
tot, gp = collections.Counter(), collections.Counter()
for o in out:
    for k, v in o[2].items(): tot[k] += v / n
    for g, nm in o[4]: gp[nm] += g / n
for k, v in tot.most_common(10): print(f"   {k[:60]:60s} {v:8.1f} us per tick   idle before its launches {gp.get(k, 0.0):6.1f} us")
PY
rm -rf $D
