#!/bin/bash
# Prints VGPR/SGPR/LDS/occupancy per kernel of navsim.hip (hipcc -Rpass-analysis=kernel-resource-usage).
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c -I include \
  navbot_ppo_amd/csrc/navsim.hip -o /tmp/navsim_res.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | \
  python3 -c '
import re,sys
cur=None
for l in sys.stdin:
    m=re.search(r"remark: (.*?) \[-Rpass",l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        cur=t.split(":",1)[1].strip(); cur=re.sub(r"_ZN12_GLOBAL__N_1\d+","",cur)[:48]; print(); print(f"{cur:50s}",end="")
    elif any(t.startswith(k) for k in ("VGPRs:","SGPRs:","Occupancy","LDS Size","ScratchSize")):
        print(" | "+t.replace(" [bytes/lane]","").replace(" [bytes/block]","").replace(" [waves/SIMD]",""),end="")
print()'
