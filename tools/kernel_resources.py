"""compact per-kernel resource table of librda_hip.so (VGPRs, spills, scratch, LDS, occupancy): python tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys

src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rda_planner_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-shared",
                      "-o", "/tmp/rda_res.so", "rda_hip.hip", "-Rpass-analysis=kernel-resource-usage"], cwd=src, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r" Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[\w/]+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
    if " error" in line or "warning:" in line:
        print(line.rstrip())
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for k, v in rows.items():
    if flt in k:
        g = lambda n: v.get(n, "?")
        print(f"{k[:60]:60s} vgpr {g('VGPRs'):>4} spill {g('VGPRs Spill'):>3} scratch {g('ScratchSize'):>5} lds {g('LDS Size'):>6} occ {g('Occupancy')} agpr {g('AGPRs')} sgpr {g('TotalSGPRs')} sspill {g('SGPRs Spill')}")
