#!/bin/bash
# round 6: rda_opts::su_land_first = 0 / 1 / 2 on the same box - headline + size legs (tools/ab_headline.py, env switch RDA_SU_LAND_FIRST), closed-loop bits of
# every mode against mode 0 (tools/ab_bits.py: mode 1 must be bit-identical, mode 2 at rounding level), landing statistics
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_land_first; mkdir -p $O; : > $O/headline.txt
for k in 0 1 2; do
  echo "== su_land_first=$k" >> $O/headline.txt
  RDA_SU_LAND_FIRST=$k python tools/ab_headline.py --rounds 2 --steps 20 --warmup 5 cur >> $O/headline.txt 2>&1
  for x in "--n-obs 2000" "--n-obs 20" "--n-obs 200 --horizon 30 --moving" "--n-obs 100 --horizon 25"; do echo "   $x" >> $O/headline.txt; RDA_SU_LAND_FIRST=$k python tools/ab_headline.py --rounds 1 --steps 30 --warmup 10 --extra "$x" cur >> $O/headline.txt 2>&1; done
  RDA_SU_LAND_FIRST=$k python tools/ab_bits.py --worker $PWD/gpurun_out/ab_bits_lf$k.npz > $O/bits_worker_$k.txt 2>&1
done
python - > $O/bits.txt 2>&1 <<'PY'
import numpy as np, sys
sys.path.insert(0, "tools")
import ab_bits
o = {k: np.load(f"gpurun_out/ab_bits_lf{k}.npz") for k in (0, 1, 2)}
for k in (1, 2):
    for name, _, _, steps in ab_bits.SHAPES:
        ua, ub, ia, ib = o[0][name + ":u"], o[k][name + ":u"], o[0][name + ":it"], o[k][name + ":it"]
        d = np.abs(ua - ub).max(axis=1)
        print(f"land_first {k} vs 0  {name:22s} max|du| {d.max():.3e}  admm iters differ on {int((ia[:, 0] != ib[:, 0]).sum())} steps, ipm iters {int(ia[:, 1].sum())} -> {int(ib[:, 1].sum())}, status != 0: {int((ib[:, 2] != 0).sum())}")
PY
cat $O/headline.txt $O/bits.txt
