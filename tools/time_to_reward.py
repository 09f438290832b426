#!/usr/bin/env python3
"""BASELINE metric part 2: PPO wall-clock until avg_ep_rews (mean episode return of the iteration, ppo.py:833) >= +100.
Workload = BASELINE configs[1] (4096 envs, stage_1, rollout 512, 50 epochs).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navbot_ppo_amd import ppo
from navbot_ppo_amd.env import VecEnv
policy = sys.argv[1] if len(sys.argv) > 1 else "mlp64x2"
target = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
# TTR_ENVS / TTR_MAP / TTR_PER_ENV / TTR_ROLLOUT: another shard, e.g. 16384 envs on per-env stage_2 maps (BASELINE configs[2] as a
# training run: the rollout is then rollout_big_kernel, the update runs on 16384 x rollout samples)
N_ENVS, MAP = int(os.environ.get("TTR_ENVS", "4096")), os.environ.get("TTR_MAP", "stage_1")
env = VecEnv(N_ENVS, map=MAP, max_episode_steps=500, seed=0, per_env_map=os.environ.get("TTR_PER_ENV", "0") == "1")
tr = ppo.PPOTrainer(env, ppo.PPOConfig(policy=policy, seed=0, rollout_len=int(os.environ.get("TTR_ROLLOUT", "512"))))
torch.cuda.synchronize(); t0 = time.perf_counter()
hist = []
for it in range(60):
    lg = tr.iteration()
    torch.cuda.synchronize()
    hist.append((round(time.perf_counter() - t0, 3), round(lg["avg_ep_rews"], 2), round(lg["success_rate"], 4)))
    if lg["avg_ep_rews"] >= target and it >= 1:
        break
print(json.dumps({"metric": "ppo_wall_clock_to_mean_reward", "target": target, "policy": policy, "n_envs": N_ENVS, "map": MAP, "reached": hist[-1][1] >= target,
                  "seconds_incl_graph_capture": hist[-1][0], "iterations": len(hist), "env_steps": tr.env_steps,
                  "trace_(sec,mean_ep_reward,success_rate)": hist}))
# keep going a little to show learning beyond the threshold
for it in range(20):
    lg = tr.iteration()
print(json.dumps({"after_iterations": tr.i_so_far, "mean_ep_reward": round(lg["avg_ep_rews"], 2), "success_rate": round(lg["success_rate"], 4),
                  "collision_rate": round(lg["collisions"] / max(lg["episodes"], 1), 4), "avg_ep_len": round(lg["avg_ep_lens"], 1), "var": lg["var"]}))
