#!/bin/bash
# round 6: bits + same-box A/B of the in-tree build against tools/_bin/librda_hip_head.so, headline and size legs; optional tests ($TESTS)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_dev; mkdir -p $O
python tools/ab_bits.py head cur > $O/bits.txt 2>&1
python tools/ab_headline.py --rounds 2 --steps 20 --warmup 5 head cur > $O/headline.txt 2>&1
for x in "--n-obs 2000" "--n-obs 20" "--n-obs 200 --horizon 30 --moving" "--n-obs 100 --horizon 25"; do echo "== $x" >> $O/headline.txt; python tools/ab_headline.py --rounds 1 --steps 30 --warmup 10 --extra "$x" head cur >> $O/headline.txt 2>&1; done
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt; fi
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $O/bits.txt; cat $O/headline.txt
