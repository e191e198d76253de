#!/bin/bash
# Round 5: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS of k_su<20> in the headline loop for builds with padded stage strides
# (tools/_bin/librda_hip_lds_<tag>.so, -DSU_FT/-DSU_HB/-DSU_WN/-DSU_MF), and the launch time of the same loop un-profiled.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
OUT=gpurun_out/lds_conflicts; mkdir -p $OUT/scr
for TAG in head "$@"; do
  SO=""; [ "$TAG" != head ] && SO="$PWD/tools/_bin/librda_hip_lds_$TAG.so"
  D=$OUT/scr/$TAG; mkdir -p $D
  env RDA_HIP_SO=$SO rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $D -o pmc -- python bench.py --only-headline --steps 40 --warmup 5 > $D/log 2>&1
  find $D -name '*counter_collection.csv' -exec cp {} $D/c.csv \;
  python - $D/c.csv $TAG <<'PY'
import csv, sys, collections
v = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    v[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k in ("void k_su<20>", "void k_su_tracked<20>"):
    a, c = v.get((k, "SQ_ACTIVE_INST_LDS"), [0]), v.get((k, "SQ_LDS_BANK_CONFLICT"), [0])
    print(f"{sys.argv[2]:6s} {k:24s} active {sum(a)/len(a):9.0f} conflict {sum(c)/len(c):9.0f} ratio {sum(c)/max(sum(a),1):.3f}")
PY
  for i in 1 2; do env RDA_HIP_SO=$SO python bench.py --only-headline --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$TAG', 'steps/s', j['value'], 'k_su us', j['roofline']['avg_launch_us'], 'lmz us', j['roofline_secondary']['avg_launch_us'])"; done
done
find $OUT/scr -type f -delete
