#!/usr/bin/env python3
"""Dev tool: where one step of the persistent resmlp512 rollout (navsim_rollout_resmlp512) spends its time.
  build (here, no GPU):  python tools/resmlp_rollout_phases.py build     -> build/libnavsim_resmlp_phases.so
  run (GPU box):         python tools/resmlp_rollout_phases.py            -> the table in profiles/r05_rollout_resmlp_phases.txt
The build is the product's navsim.hip + resmlp_policy.h with wall_clock64() differences accumulated at the phase boundaries by
lane 0 of wave 0 of workgroup 7 (100 MHz counter; the sums over the 512 steps of a launch are what is printed).  The stamps cost
about 1 us per step in total (global read-modify-writes), so the phases add up to more than the product's step."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
LIB = os.path.join(R, "build", "libnavsim_resmlp_phases.so")
NAMES = ["(loop top)", "block 1: 32 MFMAs per wave, weights from LDS; block 2's weights requested in front",
         "partial sums -> LDS, barrier (waits for block 2's weights), sum over the 8 waves",
         "block 2: 64 MFMAs per wave", "partial sums -> LDS, barrier, sums + heads on wave 0",
         "finish on wave 0 (clamp, log-prob, stores), barrier", "env step (step_body: 16 envs on 8 waves)"]

def build():
    csrc = os.path.join(R, "navbot_ppo_amd", "csrc")
    h = open(os.path.join(csrc, "resmlp_policy.h")).read()
    t = open(os.path.join(csrc, "navsim.hip")).read()
    def rep(s, a, b):
        assert s.count(a) == 1, a
        return s.replace(a, b)
    h = rep(h, "    *reinterpret_cast<float4*>(&ps.part1[w][4 * lane]) =", "    RESMLP_MARK(1);\n    *reinterpret_cast<float4*>(&ps.part1[w][4 * lane]) =")
    h = rep(h, "    // rb2\n", "    RESMLP_MARK(2);\n")
    h = rep(h, "    *reinterpret_cast<float4*>(&ps.part2[w][0][4 * lane]) =", "    RESMLP_MARK(3);\n    *reinterpret_cast<float4*>(&ps.part2[w][0][4 * lane]) =")
    h = rep(h, "    z3_out = z3;\n", "    RESMLP_MARK(4);\n    z3_out = z3;\n")
    os.makedirs("/tmp/resmlp_phases", exist_ok=True)
    open("/tmp/resmlp_phases/resmlp_policy.h", "w").write(h)
    t = rep(t, '#include "resmlp_policy.h"',
            "__device__ unsigned long long g_dbg[16];\n"
            "#define RESMLP_MARK(i) do { if (blockIdx.x == 7 && threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); "
            "g_dbg[i] += now_ - g_dbg[15]; g_dbg[15] = now_; } } while (0)\n"
            '#include "/tmp/resmlp_phases/resmlp_policy.h"')
    t = rep(t, "        resmlp::load_weights_b(R.params, lane, wave, W);", "        RESMLP_MARK(0);\n        resmlp::load_weights_b(R.params, lane, wave, W);")
    t = rep(t, "            R.logp_buf[tn + base + l15] = o.logp;\n        }\n        __syncthreads();\n",
            "            R.logp_buf[tn + base + l15] = o.logp;\n        }\n        __syncthreads();\n        RESMLP_MARK(5);\n")
    t = rep(t, "        __syncthreads();   // the observation tile of step t + 1 is complete in sm.obs", "        RESMLP_MARK(6);\n        __syncthreads();   //")
    t = rep(t, "int navsim_version(void) { return NAVSIM_ABI_VERSION; }",
            "int navsim_version(void) { return NAVSIM_ABI_VERSION; }\n"
            "int navsim_resmlp_phases(unsigned long long* out, int reset) {\n"
            "    if (hipDeviceSynchronize() != hipSuccess) return -1;\n"
            "    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;\n"
            "    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), z, sizeof z) != hipSuccess) return -1; }\n"
            "    return 0;\n}")
    open("/tmp/resmlp_phases/navsim_phases.hip", "w").write(t)
    from navbot_ppo_amd.build import build_native
    print(build_native(navsim_src="/tmp/resmlp_phases/navsim_phases.hip", out=LIB))

def run():
    os.environ["NAVSIM_LIB"] = LIB
    import torch
    from navbot_ppo_amd import ppo, _native
    from navbot_ppo_amd.env import VecEnv
    L = _native.lib()
    L.navsim_resmlp_phases.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    T = 512
    for N in (4096, 1024):
        env = VecEnv(N, map="stage_1", max_episode_steps=500, seed=0)
        tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=T, policy="resmlp512", seed=0, persistent_rollout=True))
        tr.rollout(); torch.cuda.synchronize()
        d = (ctypes.c_ulonglong * 16)()
        assert L.navsim_resmlp_phases(d, 1) == 0
        reps = 3
        for _ in range(reps): tr.rollout()
        assert L.navsim_resmlp_phases(d, 1) == 0
        print(f"resmlp512 persistent rollout, {N} envs, stage_1, 10 beams: us per step by phase (workgroup 7, wave 0; {reps} x {T} steps)")
        tot = 0.0
        for i in range(1, 7):
            us = d[i] * 0.01 / (reps * T); tot += us
            print(f"  {us:6.2f}  {NAMES[i]}")
        print(f"  {tot:6.2f}  sum (without the barrier that closes the step: that interval lands in slot 0 with the launch's first stamp)")
        env.close()

if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
