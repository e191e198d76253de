#!/bin/bash
# same-box A/B of builds of librda_hip.so (tools/_bin/librda_hip_<tag>.so): headline (re-sorted), fixed binding, k_su us; two interleaved rounds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-sizes --no-cpu-baseline --no-ip-legs --egos 0 --fleet-egos ${FLEET:-0} --steps ${STEPS:-40} --warmup 10 $EXTRA"
for ROUND in 1 2; do for TAG in "$@"; do
  RDA_HIP_SO=$PWD/tools/_bin/librda_hip_$TAG.so $B 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']; f=j['roofline_fixed_slot_binding_replay']['k_su']
l=j['roofline_fixed_slot_binding_replay']['k_lammuz']; fl=(j.get('multi_ego_fleet') or {}).get('aggregate_steps_per_s')
print('$TAG round $ROUND: value', j['value'], 'su us', r['avg_launch_us'], '| fixed', j['fixed_slot_binding']['steps_per_s'], 'replay', j['device_resident_replay']['steps_per_s'], 'su us', f['avg_launch_us'], 'lmz us', l['avg_launch_us'], l['kernel'][:24], '| fleet', fl)"
done; done
