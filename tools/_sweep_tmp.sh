B="python bench.py --no-cpu-baseline --egos 0 --fleet-egos 0 --no-ip-legs --no-shard-leg"
i=0
for S in "1e-12,1e-12,1e-12,0.999999,1e-7,2" "1e-11,1e-11,1e-11,0.99999999999,1e-12,2" "1e-13,1e-13,1e-13,0.9999999999999,1e-14,2" "1e-14,1e-14,1e-14,0.99999999999999,1e-15,2" "1e-12,1e-13,1e-12,0.999999999999,1e-13,2" "1e-12,1e-12,1e-12,0.999999999999,1e-13,4"; do
  i=$((i+1))
  if [ -z "$S" ]; then $B 2>/dev/null | grep '^{' > gpurun_out/sw_$i.json; else RDA_SU_EASY="$S" $B 2>/dev/null | grep '^{' > gpurun_out/sw_$i.json; fi
  python - "$S" gpurun_out/sw_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read())
print(f"{sys.argv[1] or 'default':48s} value {d['value']:7.1f} replay {d['device_resident_replay']['steps_per_s']:7.1f} iters {d['mean_admm_iters']} k_su {d['roofline']['avg_launch_us']} maxdu {d['max_du_vs_python_closed_loop']}")
PY
done

E="1e-12,1e-12,1e-12,0.999999999999,1e-13,2"
for CFG in "--n-obs 20 --steps 100" "--n-obs 2000 --steps 60 --warmup 5" "--moving --horizon 30 --steps 60 --warmup 5" "--n-obs 100 --horizon 25 --steps 100"; do
  for EE in "" "$E"; do
    if [ -z "$EE" ]; then $B $CFG 2>/dev/null | grep '^{' > gpurun_out/sw_c.json; else RDA_SU_EASY="$EE" $B $CFG 2>/dev/null | grep '^{' > gpurun_out/sw_c.json; fi
    python - "$CFG | ${EE:-default}" gpurun_out/sw_c.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read())
print(f"{sys.argv[1]:90s} value {d['value']:7.1f} replay {d['device_resident_replay']['steps_per_s']:7.1f} iters {d['mean_admm_iters']} {d['roofline']['kernel']} {d['roofline']['avg_launch_us']}")
PY
  done
done
