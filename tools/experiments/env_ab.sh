#!/bin/bash
# same-box A/B of environment switches on the current library: each argument is one "VAR=value [VAR=value ...]" set (X=1 = defaults)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-sizes --no-cpu-baseline --no-ip-legs --egos 0 --fleet-egos 0 --steps ${STEPS:-40} --warmup 10 $EXTRA"
for v in "$@"; do env $v $B 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('$v: value', j['value'], 'iters', j['mean_admm_iters'], 'su us', r['avg_launch_us'], '| fixed', j['fixed_slot_binding']['steps_per_s'], '| follow', (j.get('duals_follow_obstacles') or {}).get('steps_per_s'))"
done
