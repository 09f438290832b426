#!/usr/bin/env python3
"""Dev tool: build/libnavsim_raytime.so = navsim.hip with wall_clock64 stamps inside the per-env ray work loop
(per sampled workgroup, wave and work item: grabbed / operands ready / tiles done / minima merged); read by --report."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--report" in sys.argv:
    sys.path.insert(0, R)
    import ctypes as C, numpy as np, torch
    from navbot_ppo_amd import _native, maps
    _native.LIB_PATH = os.path.join(R, "build", "libnavsim_raytime.so")
    from navbot_ppo_amd.env import NavSim
    N = 16384
    sim = NavSim(N, max_episode_steps=500, auto_reset=True, seed=0)
    sim.set_map(maps.replicate_per_env(maps.stage_2(), N, seed=0), per_env=True)
    io = sim.alloc_io(); sim.reset(io.obs)
    acts = torch.rand((N, 2), device="cuda"); acts[:, 1] = acts[:, 1] * 2 - 1
    for k in range(20): sim.step(acts, io.obs, io.reward, io.done, io.arrive, io.ended)
    torch.cuda.synchronize()
    buf = np.zeros(4 * 4 * 8 * 4, dtype=np.int64)
    L = _native.lib(); L.navsim_ray_read.argtypes = [C.c_void_p]; L.navsim_ray_read(buf.ctypes.data_as(C.c_void_p))
    b = buf.reshape(4, 4, 8, 4)
    for blk in range(4):
        t0 = b[blk][b[blk] > 0].min()
        for w in range(4):
            print(f"block {blk} wave {w}: " + " | ".join(
                "item%d grab+%4d ops+%4d tiles+%5d merge+%4d" % (it, (b[blk, w, it, 0] - t0) * 10, (b[blk, w, it, 1] - b[blk, w, it, 0]) * 10,
                (b[blk, w, it, 2] - b[blk, w, it, 1]) * 10, (b[blk, w, it, 3] - b[blk, w, it, 2]) * 10) for it in range(8) if b[blk, w, it, 0] > 0))
    sys.exit(0)
t = open(os.path.join(R, "navbot_ppo_amd/csrc/navsim.hip")).read()
def rep(a, b):
    global t
    assert a in t, a
    t = t.replace(a, b, 1)
rep("template <int NB, int EPB>\nstruct StepSmem {",
    "__device__ long long g_ray[4 * 4 * 8 * 4];\n"
    "#define RSTAMP(k) do { if (blockIdx.x % 300 == 0 && lane == 0 && item_ < 8) g_ray[(((blockIdx.x / 300) * 4 + wave) * 8 + item_) * 4 + (k)] = wall_clock64(); } while (0)\n"
    "template <int NB, int EPB>\nstruct StepSmem {")
rep("        while (cur < nloc) {  // wave-uniform\n            const int nxt = grab();", "        int item_ = 0;\n        while (cur < nloc) {  // wave-uniform\n            RSTAMP(0);\n            const int nxt = grab();")
rep("            auto accumulate = [&](const float4 g) {", "            RSTAMP(1);\n            auto accumulate = [&](const float4 g) {")
rep("#pragma unroll\n            for (int b = 0; b < NB; ++b)\n                if (best[b] < kInfBits) atomicMin(&sm.rng[b * EPB + cur], best[b]);",
    "            RSTAMP(2);\n#pragma unroll\n            for (int b = 0; b < NB; ++b)\n                if (best[b] < kInfBits) atomicMin(&sm.rng[b * EPB + cur], best[b]);\n            RSTAMP(3);\n            ++item_;")
rep("int navsim_version(void) { return NAVSIM_ABI_VERSION; }",
    "int navsim_version(void) { return NAVSIM_ABI_VERSION; }\n"
    "int navsim_ray_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ray), sizeof(long long) * 512); }")
open("/tmp/navsim_raytime.hip", "w").write(t)
out = os.path.join(R, "build", "libnavsim_raytime.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
                       "-fvisibility=hidden", "-I", os.path.join(R, "include"), "/tmp/navsim_raytime.hip",
                       os.path.join(R, "navbot_ppo_amd/csrc/ppo_mlp64.hip"), "-o", out])
print(out)
