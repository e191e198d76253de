#!/bin/bash
# Instruction-mix / issue counters of the bench kernels (evidence for "fp64-issue bound, not HBM bound").
#   gpurun -- 'bash tools/profile_issue.sh r01'      (one rocprofv3 --pmc pass per group; kernel-trace only)
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
SUMMARY=gpurun_out/issue_${TAG}.txt
: > "$SUMMARY"
for GROUP in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"; do
  OUT=gpurun_out/issue_${TAG}_$(echo $GROUP | cut -d' ' -f1)
  rm -rf "$OUT"; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$OUT" -o pmc -- \
      python bench.py --steps 40 --warmup 5 --no-cpu-baseline --egos 0 --fleet-egos 0 > "$OUT/bench.log" 2>&1 || true
  find "$OUT" -name '*counter_collection.csv' -exec cp {} "$OUT/counters.csv" \;
  python - "$OUT/counters.csv" >> "$SUMMARY" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
try:
    for row in csv.DictReader(open(sys.argv[1])):
        k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
        tot[k] += float(row["Counter_Value"]); cnt[k] += 1
    for k in sorted(tot):
        if "k_su" in k[0] or "k_lammuz" in k[0]:
            print(f"{k[0]:28s} {k[1]:28s} dispatches {cnt[k]:5d} per-dispatch {tot[k]/cnt[k]:14.1f}")
except Exception as e:
    print("parse failed", e)
PY
  find "$OUT" -name '*counter_collection.csv' -size +1M -delete
  rm -f "$OUT/counters.csv"
done
cat "$SUMMARY"
