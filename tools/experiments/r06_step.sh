#!/bin/bash
# round 6, one development step of k_su: bits against the round-5 build, same-box headline A/B, per-wave trace of the new build
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_step; mkdir -p $O
python tools/ab_bits.py base new > $O/bits.txt 2>&1
python tools/ab_headline.py --rounds 2 base new > $O/headline.txt 2>&1
RDA_HIP_SO=$PWD/tools/_bin/librda_hip_trace.so python tools/su_trace.py --steps 30 > $O/trace.txt 2>&1
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt; fi
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/bits.txt; cat $O/headline.txt; cat $O/trace.txt
