#!/bin/bash
# round 6: the `sizes` legs of bench.py (sub-processes) print 10 - 15 % less when the parent has run its fleet / multi-ego / interior-point legs before them
cd "${GRAFT_REPO_ROOT:-/root/repo}"
show() { python -c "
import sys, json
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], j['value'], {k: (v.get('value'), v.get('fleet')) for k, v in (j.get('sizes') or {}).items()})
" $1; }
RDA_BENCH_GC=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/x1.json 2> gpurun_out/x1.err; grep "gc before" gpurun_out/x1.err; show gpurun_out/x1.json
RDA_BENCH_GC=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fleet-egos 0 > gpurun_out/x3.json 2> gpurun_out/x3.err; grep "gc before" gpurun_out/x3.err; show gpurun_out/x3.json
RDA_BENCH_GC=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --egos 0 --no-ip-legs > gpurun_out/x4.json 2> gpurun_out/x4.err; grep "gc before" gpurun_out/x4.err; show gpurun_out/x4.json
