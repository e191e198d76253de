#!/bin/bash
# All profiling passes of one round on the GPU box (gpurun -- 'bash tools/profile_round.sh r02'):
#   1. rocprofv3 --kernel-trace --stats over the default bench workload        -> gpurun_out/prof_<tag>/kernel_stats.csv, bench.json
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each (kernel-trace only) -> gpurun_out/prof_<tag>/pmc_fetch_write.txt
#   3. SQ issue counters, four passes                                            -> gpurun_out/prof_<tag>/issue_counters.txt
#   4. the other BASELINE configurations, plain bench runs                       -> gpurun_out/prof_<tag>/bench_<config>.json
#   5. rocprofv3 kernel stats of the C5 fleet and of N=2000 (split LamMuZ launch)  -> gpurun_out/prof_<tag>/kernel_stats_{c5_fleet,n2000}.csv
# then `python tools/profile_collect.py <tag>` (CPU) turns that into the committed summaries under profiles/.
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_${TAG}
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python bench.py --no-cpu-baseline --egos 0 --fleet-egos 0"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH --steps 200 --warmup 10 > "$OUT/bench_under_rocprof.log" 2>&1 || true
grep '^{' "$OUT/bench_under_rocprof.log" | tail -1 > "$OUT/bench_under_rocprof.json" || true
find "$OUT/stats" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/stats"

: > "$OUT/pmc_fetch_write.txt"
for C in FETCH_SIZE WRITE_SIZE; do
  D="$OUT/pmc_$C"; mkdir -p "$D"
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$D" -o pmc -- $BENCH --steps 40 --warmup 5 > "$D/bench.log" 2>&1 || true
  grep '^{' "$D/bench.log" | tail -1 > "$OUT/bench_pmc_$C.json" || true
  find "$D" -name '*counter_collection.csv' -exec cp {} "$D/counters.csv" \;
  python - "$D/counters.csv" $C >> "$OUT/pmc_fetch_write.txt" <<'PY'
import csv, sys, collections
path, cname = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(list)
try:
    for row in csv.DictReader(open(path)):
        if row.get("Counter_Name") == cname:
            vals[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    for k, v in vals.items():
        # launches queued behind the early-stop flag return at once and move (almost) nothing, the su launch that detects the stop reads
        # the two residual arrays (a third of a first solve): an EXECUTED dispatch is one that moved more than 0.4 of the largest
        # dispatch of that kernel (a later su solve reads half of what a first one reads)
        ex = [x for x in v if x > 0.4 * max(v)] or v
        print(f"{cname} {k}: dispatches {len(v)} total {sum(v):.1f} per-dispatch {sum(v)/len(v):.3f} executed {len(ex)} per-executed {sum(ex)/len(ex):.3f}")
except Exception as e:
    print("parse failed", e)
PY
  rm -rf "$D"
done

: > "$OUT/issue_counters.txt"
for GROUP in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM"; do
  D="$OUT/issue_tmp"; rm -rf "$D"; mkdir -p "$D"
  rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$D" -o pmc -- $BENCH --steps 40 --warmup 5 > "$D/bench.log" 2>&1 || true
  find "$D" -name '*counter_collection.csv' -exec cp {} "$D/counters.csv" \;
  python - "$D/counters.csv" >> "$OUT/issue_counters.txt" <<'PY'
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
  rm -rf "$D"
done

timeout 300 python bench.py --egos 16 --fleet-egos 64 2> /dev/null | grep '^{' > "$OUT/bench.json"
timeout 200 $BENCH --n-obs 20 2> /dev/null | grep '^{' > "$OUT/bench_n20.json"
timeout 300 $BENCH --n-obs 2000 --steps 60 2> /dev/null | grep '^{' > "$OUT/bench_n2000.json"
timeout 300 $BENCH --moving --horizon 30 --steps 60 2> /dev/null | grep '^{' > "$OUT/bench_dynamic_obs.json"
timeout 300 python bench.py --no-cpu-baseline --egos 0 --fleet-egos 64 --n-obs 100 --horizon 25 2> /dev/null | grep '^{' > "$OUT/bench_c5_fleet.json"
# 5. per-kernel times of the fleet / dense-grid form (split LamMuZ launch): BASELINE C5 (64 egos x 100 obstacles, T=25) and N=2000
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_c5" -o f -- python bench.py --no-cpu-baseline --egos 0 --fleet-egos 64 --n-obs 100 --horizon 25 --steps 100 > /dev/null 2>&1 || true
find "$OUT/stats_c5" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_c5_fleet.csv" \;
rm -rf "$OUT/stats_c5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_n2000" -o f -- $BENCH --n-obs 2000 --steps 60 > /dev/null 2>&1 || true
find "$OUT/stats_n2000" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_n2000.csv" \;
rm -rf "$OUT/stats_n2000"
ls -la "$OUT"; head -12 "$OUT/kernel_stats.csv"
