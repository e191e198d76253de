#!/bin/bash
# Per-kernel timing of the bench workload with rocprofv3 (run on the GPU box through gpurun):
#   gpurun -- 'bash tools/profile_bench.sh r01'
# writes gpurun_out/prof_<tag>/ ; copy the *_kernel_stats.csv summary into profiles/.
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_${TAG}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- \
    python bench.py --steps 200 --warmup 10 --no-cpu-baseline --egos 0 --fleet-egos 0 > "$OUT/bench.log" 2>&1 || true
grep '^{' "$OUT/bench.log" > "$OUT/bench.json" || true
find "$OUT" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
cat "$OUT/kernel_stats.csv" 2>/dev/null | head -20
# drop the (large) per-dispatch trace, keep the summary
find "$OUT" -name '*kernel_trace.csv' -size +2M -delete
