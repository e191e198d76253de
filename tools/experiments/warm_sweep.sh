#!/bin/bash
# A/B of the su start rules on the headline loop (re-sorted every tick): bench value + mean interior-point iterations per solve
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-sizes --no-cpu-baseline --no-ip-legs --egos 0 --fleet-egos 0 --steps 80 --warmup 10"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('   value', j['value'], 'iters', j['mean_admm_iters'], 'su us', r['avg_launch_us'], 'fixed', j['fixed_slot_binding']['steps_per_s'])"; }
run X=1
run RDA_SU_WARM=1e-2,1e-2
run RDA_SU_WARM=1e-2,1e-1
run RDA_SU_WARM=1e-4,1e-4
run RDA_SU_WARM=0,0
run RDA_SU_COLD_FROM=0
run RDA_SU_COLD_FROM=4,8
run RDA_SU_COLD_FROM=12,8
run RDA_SU_EASY=1e-12,1e-12,1e-12,0.999999,1e-7,4
run RDA_SU_EASY=1e-6,1e-6,1e-6,0.999999,1e-7,6
run RDA_SU_WARM_ENDGAME=0.999,1e-4
