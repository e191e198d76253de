#!/bin/bash
# round 6: does an IDLE process that holds a HIP context on the same GPU slow another process's closed loops down?  (the `sizes` legs of bench.py run as
# sub-processes of a parent that has finished its GPU work but still holds its handles: they print 7 - 12 % less than the same command run alone)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
leg() { python bench.py --gpus 1 --size-leg --no-cpu-baseline --n-obs 200 --horizon 30 --moving --steps 30 --warmup 8 --fleet-egos 0 2>/dev/null | grep '^{' | tail -1 | python -c "import sys, json; j = json.loads(sys.stdin.read()); print(j['value'], j['roofline']['avg_launch_us'])"; }
echo "alone:"; leg; leg
python - <<'PY' &
import ctypes, time, sys, os
sys.path.insert(0, os.getcwd())
from rda_planner_amd._lib import hip_api
from rda_planner_amd import scenarios as sc
from rda_planner_amd.rda_solver import RDA_solver
api = hip_api()
sv = [RDA_solver(20, sc.rectangle_robot(dynamics="acker"), 4, 200, iter_num=4, time_print=False) for _ in range(8)]
time.sleep(45)
PY
HOLD=$!
sleep 8
echo "beside an idle process with 8 handles:"; leg; leg
wait $HOLD
echo "alone again:"; leg
