cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c5prof; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o f -- python bench.py --size-leg --cpu-threads 16 --n-obs 100 --horizon 25 --steps 30 --warmup 8 --fleet-egos 64 --no-cpu-baseline > $D/run.log 2>&1
find $D -name '*kernel_stats.csv' -exec cp {} gpurun_out/c5_kernel_stats.csv \;
head -30 gpurun_out/c5_kernel_stats.csv | cut -c1-150
rm -rf $D
