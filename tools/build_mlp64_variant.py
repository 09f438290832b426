#!/usr/bin/env python3
"""Dev tool: build/libnavsim_<name>.so = the product library with csrc/ppo_mlp64.hip compiled under extra flags (e.g.
-DX3_EXPERIMENT_NO_MFMA, -DX3_EXPERIMENT_NO_SPLIT: timing experiments of the split-bf16 pass; their results are wrong by design) -- for
A/B timing in one gpurun call through NAVSIM_LIB=build/libnavsim_<name>.so.  The other objects are the product's (build/obj).
usage: python tools/build_mlp64_variant.py <name> [flags...]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from navbot_ppo_amd import build
name, flags = sys.argv[1], sys.argv[2:]
SRC = os.environ.get("VARIANT_SRC", "ppo_mlp64.hip")   # or ppo_resmlp512.hip
build.build_native()   # the product objects are current
objdir = os.path.join(R, "build", "obj_" + name)
os.makedirs(objdir, exist_ok=True)
obj = os.path.join(objdir, SRC + ".o")
cflags = [f for f in build.HIPCC_FLAGS if f != "-shared"]
per_src = build.EXTRA_FLAGS.get(SRC, [])
if per_src and build.flags_accepted(per_src):
    cflags += list(per_src)
subprocess.check_call([build.hipcc()] + cflags + flags + ["-I", build.INC, "-I", os.path.join(build.HERE, "csrc"), "-c",
                                                         os.path.join(build.HERE, "csrc", SRC), "-o", obj])
objs = [os.path.join(R, "build", "obj", f) for f in ("navsim.hip.o", "ppo_mlp64.hip.o", "ppo_resmlp512.hip.o") if f != SRC + ".o"] + [obj]
out = os.path.join(R, "build", f"libnavsim_{name}.so")
subprocess.check_call([build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + objs + ["-o", out])
print(out)
