#!/usr/bin/env python3
"""Dev tool: the 512-step rollout of the reference's ACTIVE nets (resmlp512) at 4096 envs: one persistent launch (navsim_rollout_resmlp512)
against the hipGraph of 512 x (navppo_resmlp512_act + navsim_step) launches.  usage: python tools/time_rollout_resmlp.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navbot_ppo_amd import ppo
from navbot_ppo_amd.env import VecEnv
for N in (4096, 1024, 8192):
    for persistent in (True, False):
        env = VecEnv(N, map="stage_1", max_episode_steps=500, seed=0)
        tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=512, policy="resmlp512", seed=0, persistent_rollout=persistent))
        tr.rollout(); tr.rollout(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): tr.rollout()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f"resmlp512 N={N} {'persistent kernel' if persistent else 'hipGraph of per-step launches'}: {ms:.3f} ms per 512-step rollout = {ms / 512 * 1e3:.2f} us per step")
        env.close()
