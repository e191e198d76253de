#!/bin/bash
# su start rules in the interior-point LamMuZ mode (lmz_central = 1e-3): interior-point iterations and ticks per su-solve (tools/su_phase_profile.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $*"; env "$@" python tools/su_phase_profile.py --lmz-central 1e-3 --steps ${STEPS:-60} $EXTRA 2>&1 | grep "su-solves"; }
for v in "$@"; do run $v; done
