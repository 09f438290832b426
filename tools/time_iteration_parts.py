#!/usr/bin/env python3
"""Dev tool: where one PPO iteration at BASELINE configs[1] spends its time outside the 50 update epochs."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from navbot_ppo_amd import ppo
from navbot_ppo_amd.env import VecEnv
env = VecEnv(4096, map="stage_1", max_episode_steps=500, seed=0)
tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=512, policy="mlp64x2", n_updates_per_iteration=50))
for _ in range(2): tr.iteration()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
T, N = 512, 4096
obs, acts, logp, rtg = tr.obs_buf[:T].reshape(T * N, 16), tr.act_buf.reshape(T * N, 2), tr.logp_buf.reshape(T * N), tr.rtg_buf.reshape(T * N)
up = tr.updater
print(f"iteration           {t(tr.iteration, 3):8.3f} ms")
print(f"rollout             {t(tr.rollout):8.3f} ms")
print(f"_rollout_metrics    {t(tr._rollout_metrics):8.3f} ms")
print(f"update (50 epochs)  {t(lambda: up.update(obs, acts, logp, rtg, tr.var), 3):8.3f} ms")
with torch.no_grad():
    print(f"  V0                {t(lambda: up._fused_value(obs)):8.3f} ms")
    V0 = up._fused_value(obs)
    print(f"  normalise adv     {t(lambda: ppo.normalise_advantages(rtg - V0, None)):8.3f} ms")
    adv = ppo.normalise_advantages(rtg - V0, None)
hist = torch.zeros((50, 8), device="cuda")
def epochs():
    for ep in range(50): up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, hist[ep])
print(f"  50 epochs         {t(epochs, 3):8.3f} ms")
print(f"  float(var)        {t(lambda: float(tr.var)):8.3f} ms")
