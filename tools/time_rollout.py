#!/usr/bin/env python3
"""Dev tool: the persistent rollout kernel (navsim_rollout_mlp64) at the shard sizes of a strong-scaling run: 4096 envs over
1 / 2 / 4 / 8 GPUs = 4096 / 2048 / 1024 / 512 envs per GPU.  NAVSIM_EPB = 4 | 8 | 16 forces the envs per workgroup (64: the
big-shard kernel, rollout_big_kernel, which shards beyond 4096 envs select by themselves).
TR_SIZES=16384 TR_MAP=stage_2 TR_PER_ENV=1 TR_T=256 [TR_SIDES=248]: the closed-loop form of BASELINE configs[2] (S=1024 with TR_SIDES).
usage: python tools/time_rollout.py [lib.so] [policy]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navbot_ppo_amd import ppo
from navbot_ppo_amd.env import VecEnv
from navbot_ppo_amd import _native
_libs = [a for a in sys.argv[1:] if a.endswith(".so")]
if _libs:
    _native.LIB_PATH = os.path.abspath(_libs[0])
if "--cfg3" in sys.argv:   # the closed-loop form of BASELINE configs[2] (what bench.py's roofline_closed_loop runs)
    os.environ.update(TR_SIZES="16384", TR_MAP="stage_2", TR_PER_ENV="1", TR_T="256")
if "--s1024" in sys.argv:   # the same shard past the Infinity Cache (bench.py's roofline_closed_loop_beyond_l3)
    os.environ.update(TR_SIZES="16384", TR_PER_ENV="1", TR_T="64", TR_SIDES="248")
if "--s2048" in sys.argv:   # 2x the Infinity Cache (bench.py's roofline_hbm_closed_loop)
    os.environ.update(TR_SIZES="16384", TR_PER_ENV="1", TR_T="32", TR_SIDES="504")
_pol = [a for a in sys.argv[1:] if not a.endswith(".so") and not a.startswith("--")]
policy = _pol[0] if _pol else "mlp64x2"
from navbot_ppo_amd import maps
MAP, PER_ENV, T = os.environ.get("TR_MAP", "stage_1"), os.environ.get("TR_PER_ENV", "0") == "1", int(os.environ.get("TR_T", "512"))
SIDES = int(os.environ.get("TR_SIDES", "0"))
for N in [int(x) for x in os.environ.get("TR_SIZES", "4096,2048,1024,512").split(",")]:
    if SIDES:   # stage_2 with SIDES-gon pillars (248: 1024 segments), stage_2's goal rectangles
        env = VecEnv(N, map=maps.stage_2(sides=SIDES), max_episode_steps=500, seed=0, per_env_map=PER_ENV)   # (segments: 32 + 4 SIDES)
        rr, rs = maps.goal_rects("stage_2")
        env.sim.set_goal_rects(0, rr)
        env.sim.set_goal_rects(1, rs)
    else:
        env = VecEnv(N, map=MAP, max_episode_steps=500, seed=0, per_env_map=PER_ENV, sampler="small_house" if MAP == "house" else None)
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(policy=policy, rollout_len=T, seed=0))
    for _ in range(2):
        tr.rollout()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = int(os.environ.get("TR_REPS", "16"))   # (the rocprofv3 average of a run includes its 2 warm-up launches: keep them a small share)
    for _ in range(reps):
        tr.rollout()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tag = MAP + (" per-env" if PER_ENV else "") + (" sides=%d" % SIDES if SIDES else "")
    print(f"{policy} {tag} T={T} N={N:5d} EPB={os.environ.get('NAVSIM_EPB', 'auto'):>4s}: rollout {ms:7.3f} ms = {ms / T * 1e3:6.2f} us per step, "
          f"{N * T / ms / 1e3:8.1f} M env-steps/s", flush=True)
    env.close()
