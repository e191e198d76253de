#!/bin/bash
# The PROFILING builds of librda_hip.so that tools/profile_round.sh / tools/su_phase_profile.py load through RDA_HIP_SO (they travel to the GPU box with
# the snapshot; tools/_bin/ is git-ignored).  Same sources as the product build, plus the phase counters of the su-solve:
#   librda_hip_prof.so  -DSU_PROF   the 16 set-up / iteration phases        librda_hip_fine.so  -DSU_FINE   the sub-phases of one interior-point iteration
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/_bin
for V in prof:-DSU_PROF fine:-DSU_FINE; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function ${V#*:} -shared -o tools/_bin/librda_hip_${V%%:*}.so rda_planner_amd/csrc/rda_hip.hip || exit 1
done
ls -la tools/_bin/librda_hip_prof.so tools/_bin/librda_hip_fine.so
