#!/usr/bin/env python3
"""Dev tool: the persistent rollout kernel (navsim_rollout_mlp64) at the shard sizes of a strong-scaling run: 4096 envs over
1 / 2 / 4 / 8 GPUs = 4096 / 2048 / 1024 / 512 envs per GPU.  NAVSIM_EPB = 4 | 8 | 16 forces the envs per workgroup.
usage: python tools/time_rollout.py [lib.so] [policy]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from navbot_ppo_amd import ppo
from navbot_ppo_amd.env import VecEnv
from navbot_ppo_amd import _native
_libs = [a for a in sys.argv[1:] if a.endswith(".so")]
if _libs:
    _native.LIB_PATH = os.path.abspath(_libs[0])
_pol = [a for a in sys.argv[1:] if not a.endswith(".so")]
policy = _pol[0] if _pol else "mlp64x2"
for N in [int(x) for x in os.environ.get("TR_SIZES", "4096,2048,1024,512").split(",")]:
    env = VecEnv(N, map="stage_1", max_episode_steps=500, seed=0)
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(policy=policy, rollout_len=512, seed=0))
    for _ in range(2):
        tr.rollout()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        tr.rollout()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{policy} N={N:5d} EPB={os.environ.get('NAVSIM_EPB', 'auto'):>4s}: rollout {ms:7.3f} ms = {ms / 512 * 1e3:6.2f} us per step, "
          f"{N * 512 / ms / 1e3:8.1f} M env-steps/s", flush=True)
    env.close()
