#!/usr/bin/env python3
"""Dev tool: builds build/libnavsim_timing.so = navsim.hip + wall_clock64 stamps at the phase boundaries and
per-block start/end/placement records (read back by tools/phase_timing.py)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.environ.get("NAVSIM_SRC", os.path.join(R, "navbot_ppo_amd/csrc/navsim.hip"))   # another revision of the kernel file
OUT = os.environ.get("NAVSIM_TIMING_OUT", "libnavsim_timing.so")
t = open(SRC).read()
extra_flags = sys.argv[1:]
def rep(a, b, n=1):
    global t
    assert a in t, a
    t = t.replace(a, b, n)
rep("template <int NB, int EPB, int NW = 4>\nstruct StepSmem {",
    "__device__ long long g_dbg[128 * 8];\n__device__ long long g_blk[8192 * 3];\n"
    "#define STAMP(slot) do { if (blockIdx.x % 97 == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 16) g_dbg[((blockIdx.x / 97) % 8) * 128 + (threadIdx.x >> 6) * 8 + (slot)] = wall_clock64(); } while (0)\n"
    "template <int NB, int EPB, int NW = 4>\nstruct StepSmem {")
rep("    if (wave < PW) {\n        // ---------------- pose lanes, part 1: motion + sensor frame\n",
    "    STAMP(0);\n    if (threadIdx.x == 0 && blockIdx.x < 8192) { g_blk[blockIdx.x * 3] = wall_clock64(); unsigned hw; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(hw)); unsigned xcc; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc)); g_blk[blockIdx.x * 3 + 2] = ((long long)xcc << 32) | hw; }\n"
    "    if (wave < PW) {\n        // ---------------- pose lanes, part 1: motion + sensor frame\n")
rep("    __syncthreads();  // barrier A:", "    STAMP(1);\n    __syncthreads();  STAMP(2); // barrier A:")
rep("    __syncthreads();  // barrier B:", "    STAMP(3);\n    __syncthreads();  STAMP(4); // barrier B:")
rep("    __syncthreads();  // barrier C:", "    STAMP(5);\n    __syncthreads();  // barrier C:")
# slot 6: the wave has seen the exact pose published (front waves: before part 2; ray waves: at their first stage-B pass)
if "            pose_ok = true;\n" in t:
    rep("            pose_ok = true;\n", "            pose_ok = true;\n            STAMP(6);\n")
# end of step_body = the closing brace after the observation tile's store (next_obs4 follows it)
idx = t.index("// Entries f0 .. f0 + KS - 1 of the observation env `e` (local) will hold when this step is over")
j = t.rfind("}\n\n", 0, idx)
assert "o[k] = sm.obs[(k / D) * DP + (k % D)];" in t[j - 300:j], "end of step_body not where the timing patch expects it"
t = t[:j] + "    STAMP(7);\n    if (threadIdx.x == 0 && blockIdx.x < 8192) g_blk[blockIdx.x * 3 + 1] = wall_clock64();\n" + t[j:]
rep("int navsim_version(void) { return NAVSIM_ABI_VERSION; }",
    "int navsim_version(void) { return NAVSIM_ABI_VERSION; }\n"
    "int navsim_dbg_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(long long) * 1024); }\n"
    "int navsim_blk_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk), sizeof(long long) * 8192 * 3); }")
# stamps of the persistent rollout kernel (last step wins): slot 6 of g_dbg = the wave is through the in-step policy hook;
# g_pol: 0 = step body left (observation tile stored), 1 = finish done (wave 0), 2 = past the step's last barrier
rep("__device__ long long g_dbg[128 * 8];", "__device__ long long g_dbg[128 * 8];\n__device__ long long g_pol[8 * 8 * 4];\n"
    "#define PSTAMP(slot) do { if (blockIdx.x % 97 == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 8) g_pol[((blockIdx.x / 97) % 8) * 32 + (threadIdx.x >> 6) * 4 + (slot)] = wall_clock64(); } while (0)")
rep("    hook(wave, lane);\n", "    hook(wave, lane);\n    STAMP(6);\n")
# (timing build only: the last step runs the in-step policy too, so that its stamps show the phase; its row lands on row T - 1)
rep("        const bool more = t + 1 < R.T;\n        auto hook = [&](const int wv, const int ln) __attribute__((always_inline)) {\n            // (measured",
    "        const bool more_real = t + 1 < R.T; const bool more = true;\n        auto hook = [&](const int wv, const int ln) __attribute__((always_inline)) {\n            // (measured")
rep("                else if (ln < nloc) draw_noise(step0 + (uint32_t)(t + 1));\n", "                else if (ln < nloc) draw_noise(step0 + (uint32_t)(t + 1));\n                PSTAMP(1);\n")
rep("                    if (ln < nloc) finish(tn + N);\n", "                    if (ln < nloc) finish(more_real ? tn + N : tn);\n                    PSTAMP(2);\n")
rep("        // the observation tile of step t + 1 is in sm.obs (its store only reads it), its action in sm.act_l\n    }\n}\n",
    "        PSTAMP(0);\n    }\n}\n")
rep("        const bool more = t + 1 < T;\n        auto hook = [&](const int wv, const int ln) __attribute__((always_inline)) {\n            if (more) {\n                if (wv >= kTileWave0 && wv < kTileWave0 + TW) tile_policy(wv - kTileWave0, tn + N, (t + 1) & 1, std::true_type{});",
    "        const bool more = true; const bool more_real = t + 1 < T;\n        auto hook = [&](const int wv, const int ln) __attribute__((always_inline)) {\n            if (more) {\n                if (wv >= kTileWave0 && wv < kTileWave0 + TW) tile_policy(wv - kTileWave0, more_real ? tn + N : tn, (t + 1) & 1, std::true_type{});")
rep("int navsim_blk_read(long long* out)", "int navsim_pol_read(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pol), sizeof(long long) * 256); }\nint navsim_blk_read(long long* out)")
# (--lb4, a forced launch bound of four waves per SIMD on step_kernel, is gone: the kernels carry their own bound since round 4 --
# min_waves_per_simd in csrc/navsim.hip)
os.makedirs(os.path.join(R, "build"), exist_ok=True)
open("/tmp/navsim_timing.hip", "w").write(t)
sys.path.insert(0, R)
from navbot_ppo_amd.build import build_native   # the product's flags, per source
print(build_native(navsim_src="/tmp/navsim_timing.hip", out=os.path.join(R, "build", OUT)))
