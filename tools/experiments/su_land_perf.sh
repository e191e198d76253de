#!/bin/bash
# what the landing of the su solve costs (round 6): headline (re-sorted) and fixed binding, default mode vs RDA_SU_LAND=1 at three stops of the interior point
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {
  env $1 python bench.py --no-sizes --no-cpu-baseline --no-ip-legs --egos 0 --fleet-egos 0 --steps 40 --warmup 10 > /dev/null 2>&1
  python - "$1" <<'PY'
import json, sys
j = json.load(open("gpurun_out/bench_detail.json")); r = j["roofline"]; f = j["fixed_slot_binding"]
print(f"{sys.argv[1]:60s} headline {j['value']:8.1f} steps/s  k_su {r['avg_launch_us']:7.2f} us  ipm/step {j['residuals']['su_interior_point_iters_per_step']:6.2f} | fixed binding {f['steps_per_s']:8.1f}  ipm/step {f['residuals']['su_interior_point_iters_per_step']:5.2f}  max_du_vs_python {j['max_du_vs_python_closed_loop']:.1e}")
PY
}
run "X=1"
run "RDA_SU_LAND=1"
run "RDA_SU_LAND=1 RDA_SU_LAND_TOL=1e-5,1e-6,1e-7"
run "RDA_SU_LAND=1 RDA_SU_LAND_TOL=1e-4,1e-5,1e-6"
run "RDA_SU_LAND=1 RDA_SU_LAND_TOL=1e-3,1e-4,1e-5"
run "X=1"
