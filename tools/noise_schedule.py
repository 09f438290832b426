#!/usr/bin/env python3
"""Dev tool: learning curves at BASELINE configs[1] under three readings of the reference's exploration-covariance decay
(ppo.py:694-695: cov *= 0.995 at every episode start once t_so_far > 50000, floor 0.1) for N parallel envs:
  default      : one decay step per N episode starts (per mean episode), from t_so_far > 50000 env-steps in all (PPOConfig default)
  per-env-time : the same, but the decay starts after 50000 steps PER ENV (var_decay_after = 50000 N): what a single reference
                 env would have seen at that point of its own life
  none         : no decay
usage: python tools/noise_schedule.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navbot_ppo_amd import ppo
from navbot_ppo_amd.env import VecEnv
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
N = 4096
for name, kw in (("default", {}), ("per-env-time", {"var_decay_after": 50000 * N}), ("none", {"var_decay": 1.0})):
    env = VecEnv(N, map="stage_1", max_episode_steps=500, seed=0)
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(policy="mlp64x2", seed=0, **kw))
    rows = []
    for it in range(iters):
        lg = tr.iteration()
        if it % 5 == 4 or it == 0:
            rows.append(f"it {it + 1:3d}: reward {lg['avg_ep_rews']:8.1f} success {lg['success_rate']:.3f} var {lg['var']:.3f}")
    torch.cuda.synchronize()
    print(f"== {name}: " + " | ".join(rows), flush=True)
    env.close()
