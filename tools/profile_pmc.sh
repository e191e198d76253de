#!/bin/bash
# HBM traffic counters of the bench kernels (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   gpurun -- 'bash tools/profile_pmc.sh r01'
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=gpurun_out/pmc_${TAG}_${C}
  rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT" -o pmc -- \
      python bench.py --steps 40 --warmup 5 --no-cpu-baseline --egos 0 --fleet-egos 0 > "$OUT/bench.log" 2>&1 || true
  find "$OUT" -name '*counter_collection.csv' -exec cp {} "$OUT/counters.csv" \;
  python - "$OUT/counters.csv" $C <<'PY'
import csv, sys, collections
path, cname = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); cnt = collections.Counter()
try:
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") == cname:
            k = row["Kernel_Name"].split("(")[0]
            tot[k] += float(row["Counter_Value"]); cnt[k] += 1
    for k in tot:
        print(f"{cname} {k}: dispatches {cnt[k]} total {tot[k]:.1f} per-dispatch {tot[k]/cnt[k]:.3f}")
except Exception as e:
    print("parse failed", e)
PY
  find "$OUT" -name '*counter_collection.csv' -size +1M -delete
  rm -f "$OUT/counters.csv"
done
