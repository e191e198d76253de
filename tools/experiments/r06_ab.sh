#!/bin/bash
# round 6: base (round-5 kernels) against the current build: bits, headline A/B, fine phase profiles, a fast slice of the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_ab; mkdir -p $O
python tools/ab_bits.py ${TAGS:-base new} > $O/bits.txt 2>&1
python tools/ab_headline.py --rounds 2 ${TAGS:-base new} > $O/headline.txt 2>&1
for t in ${TAGS:-base new}; do
  RDA_HIP_SO=$PWD/tools/_bin/librda_hip_${t}fine.so python tools/su_phase_profile.py --order --fine --steps 110 > $O/fine_$t.txt 2>&1
done
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt; fi
cat $O/bits.txt $O/headline.txt; tail -20 $O/fine_*.txt
