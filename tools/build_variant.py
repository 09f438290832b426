#!/usr/bin/env python3
"""Dev tool: build/libnavsim_<name>.so from navbot_ppo_amd/csrc/navsim.hip with a list of textual patches applied
(a Python file defining PATCHES = [(old, new), ...]); lets two kernel variants be timed in ONE gpurun call (same box:
box-to-box differences are several per cent).  usage: build_variant.py <name> [patch.py]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
t = open(os.path.join(R, "navbot_ppo_amd/csrc/navsim.hip")).read()
if len(sys.argv) > 2:
    ns = {}
    exec(open(sys.argv[2]).read(), ns)
    for old, new in ns["PATCHES"]:
        assert t.count(old) == 1, old[:80]
        t = t.replace(old, new)
src = f"/tmp/navsim_{name}.hip"
open(src, "w").write(t)
os.makedirs(os.path.join(R, "build"), exist_ok=True)
sys.path.insert(0, R)
from navbot_ppo_amd.build import build_native   # the product's flags, per source
print(build_native(navsim_src=src, out=os.path.join(R, "build", f"libnavsim_{name}.so")))
