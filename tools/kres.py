#!/usr/bin/env python3
"""Dev tool: VGPR / AGPR / spill / LDS / occupancy per kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-c",
       "-I", os.path.join(R, "include"), src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
for l in out.splitlines():
    if "error" in l: print(l)
    m = re.search(r"remark: (.*?) \[-Rpass", l)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "").split("(")[0]
        print(f"\n{cur[:44]:44s}", end="")
    elif any(t.startswith(k) for k in ("VGPRs:", "AGPRs:", "VGPRs Spill", "SGPRs:", "Occupancy", "LDS Size", "ScratchSize")):
        print(" | " + re.sub(r" \[.*?\]", "", t), end="")
print()
