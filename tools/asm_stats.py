"""static instruction statistics of the device kernels (spill traffic, DPP, transcendental counts): python tools/asm_stats.py [filter ...]"""
import os
import re
import subprocess
import sys
from collections import Counter

src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rda_planner_amd", "csrc")
if "--reuse" not in sys.argv:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S",
                    "-o", "/tmp/rda_dev.s", "rda_hip.hip"], cwd=src, check=True, capture_output=True)
cur, stats = None, {}
for line in open("/tmp/rda_dev.s"):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1); stats[cur] = Counter()
        continue
    if line.startswith(".Lfunc_end"):
        cur = None
    if cur and line.startswith("\t") and not line.strip().startswith((".", ";")):
        stats[cur][line.split()[0]] += 1
flt = [a for a in sys.argv[1:] if not a.startswith("--")] or [""]
for name, c in stats.items():
    if any(f in name for f in flt) and sum(c.values()) > 50:
        g = lambda pred: sum(v for k, v in c.items() if pred(k))
        print(f"{name[:56]:56s} instrs {sum(c.values()):6d} accvgpr {g(lambda k: 'accvgpr' in k):5d} scratch {g(lambda k: k.startswith('scratch')):4d} "
              f"dpp {g(lambda k: 'dpp' in k):5d} s_nop {c.get('s_nop', 0):4d} div {g(lambda k: 'div' in k):4d} rcp/rsq/sqrt {g(lambda k: 'rcp' in k or 'rsq' in k or 'sqrt' in k):4d} "
              f"f64 {g(lambda k: k.endswith('_f64') or '_f64_' in k):6d}")
