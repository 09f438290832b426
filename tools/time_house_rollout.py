import os, sys, torch
sys.path.insert(0, os.getcwd())
from navbot_ppo_amd import ppo
from navbot_ppo_amd.env import VecEnv
T = 128
for N in (1024, 4096, 8192):
    for epb in (None, 16, 64):
        env = VecEnv(N, map="house", max_episode_steps=500, seed=0, sampler="small_house", obs_f16=True, envs_per_workgroup=epb)
        tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=T, policy="mlp64x2", seed=0))
        inf = env.sim.info()
        tr.rollout(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): tr.rollout()
        e1.record(); torch.cuda.synchronize()
        print(f"house f16 N={N} forced={epb}: kind {inf['rollout_kind']} epb {inf['rollout_epb']} cast {inf['rollout_cast']}: {e0.elapsed_time(e1) / 3 / T * 1e3:.2f} us per step")
        env.close()
