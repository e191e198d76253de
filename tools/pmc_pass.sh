#!/bin/bash
# one rocprofv3 counter pass over the headline loop (bench.py --only-headline), summarised per solver kernel and EXECUTED dispatch:
#   bash tools/pmc_pass.sh "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" out.txt [bench args ...]
# (--pmc with --kernel-trace only: gpurun refuses counter passes combined with other trace domains)
GROUP="$1"; OUTF="$2"; shift 2
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
D="gpurun_out/pmc_scratch_$$"; mkdir -p "$D"
rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$D" -o pmc -- python bench.py --only-headline --steps 40 --warmup 10 "$@" > "$D/bench.log" 2>&1 || true
find "$D" -name '*counter_collection.csv' -exec cp {} "$D/counters.csv" \;
python - "$D/counters.csv" >> "$OUTF" <<'PY'
import csv, sys, collections
vals = collections.defaultdict(list)
try:
    for row in csv.DictReader(open(sys.argv[1])):
        vals[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for k in sorted(vals):
        if "k_su" in k[0] or "k_lammuz" in k[0] or "k_lmz" in k[0]:
            v = vals[k]
            top = sorted(v)[max(0, int(0.98 * len(v)) - 1)]
            ex = [x for x in v if x > 0.02 * top] or v
            print(f"{k[0]:28s} {k[1]:28s} dispatches {len(v):5d} per-dispatch {sum(v)/len(v):14.1f} executed {len(ex):5d} per-executed {sum(ex)/len(ex):14.1f}")
except Exception as e:
    print("parse failed", e)
PY
rm -rf "$D"
