"""Evaluation mode: the reference's ``main.evaluate`` (project_ppo/src/main.py:135-252) on the batched simulator.

Loads the newest ``actor_iter*_step*.pth`` (or an explicit path), runs ``num_episodes`` episodes with the
DETERMINISTIC mean action (no sampling, main.py:197-199) and the evaluation arrival threshold 0.4
(``is_training=False``, environment_new.py:44-47), and writes the reference's per-episode CSV
``<output_dir>/<method>/logs/<method>_eval_episodes.csv`` with its columns
``episode,success,collision,timeout,length,return,path_length,time`` (main.py:179) plus the summary it prints
(main.py:236-250).  Episodes run ``n_parallel`` at a time; each env plays whole episodes back to back.

(The reference's own loop crashes on its first step -- it feeds a (1,2) action so ``action[1]`` raises, SURVEY A3#8;
this implements the intended behaviour.)
"""
import csv
import glob
import os
import time

import numpy as np
import torch

from . import nets
from .env import VecEnv


def find_latest_checkpoint(output_dir, method_name, prefix="actor"):
    d = os.path.join(output_dir, method_name, "checkpoints")
    paths = sorted(glob.glob(os.path.join(d, f"{prefix}_iter*_step*.pth"))) or sorted(glob.glob(os.path.join(d, f"{prefix}_step*.pth")))
    return paths[-1] if paths else None


def load_actor(path, device):
    sd = torch.load(path, map_location="cpu")
    policy = "resmlp512" if any(k.startswith("rb1.") for k in sd) else "mlp64x2"
    in_dim = (sd["rb1.fc1.weight"] if policy == "resmlp512" else sd["layer1.weight"]).shape[1]  # main.py:66-75
    actor, _ = nets.make_policy(policy, in_dim, 2)
    actor.load_state_dict(sd)
    return actor.to(device).eval(), policy


@torch.no_grad()
def evaluate(actor, num_episodes=100, max_timesteps_per_episode=500, map="stage_1", n_parallel=None, seed=0, device=None,
             output_dir="", method_name="baseline", log=print):
    n_par = int(n_parallel or min(num_episodes, 1024))
    env = VecEnv(n_par, map=map, max_episode_steps=max_timesteps_per_episode, auto_reset=True, is_training=False, seed=seed,
                 device=device)
    dev = env.device
    actor = actor.to(dev).eval()
    obs = env.reset()
    pos_prev = torch.from_numpy(env.sim.get_state()["pose"][:, :2]).to(dev)
    path_len = torch.zeros(n_par, dtype=torch.float64, device=dev)
    rows, t_ep = [], time.time()
    started = torch.full((n_par,), t_ep, dtype=torch.float64)
    while len(rows) < num_episodes:
        action = actor(obs.float())                                   # deterministic mean action, main.py:197-199
        obs, rew, done, arrive = env.step(action)
        ended = env.io.ended.bool()
        pose = torch.from_numpy(env.sim.get_state()["pose"][:, :2]).to(dev)  # post-reset pose for ended envs
        step_len = (pose - pos_prev).norm(dim=1)
        path_len += torch.where(ended, torch.zeros_like(step_len), step_len)  # the terminal step's move is not observed
        pos_prev = pose
        if ended.any():
            now = time.time()
            idx = torch.nonzero(ended).flatten().tolist()
            d, a = done.cpu().numpy(), arrive.cpu().numpy()
            ln, rt = env.io.ep_length.cpu().numpy(), env.io.ep_return.cpu().numpy()
            for i in idx:
                if len(rows) >= num_episodes:
                    break
                succ = int(a[i])
                coll = int(d[i] and not a[i])
                tmo = int((not d[i]) and (not a[i]) and ln[i] >= max_timesteps_per_episode)     # main.py:214-216
                rows.append([len(rows), succ, coll, tmo, int(ln[i]), float(rt[i]), float(path_len[i]), now - float(started[i])])
                path_len[i] = 0.0
                started[i] = now
    env.close()
    arr = np.array([r[1:] for r in rows], dtype=np.float64)
    summary = dict(episodes=len(rows), success_rate=arr[:, 0].mean(), collision_rate=arr[:, 1].mean(), timeout_rate=arr[:, 2].mean(),
                   mean_length=arr[:, 3].mean(), std_length=arr[:, 3].std(), mean_return=arr[:, 4].mean(), std_return=arr[:, 4].std(),
                   mean_path_length=arr[:, 5].mean())
    csv_path = None
    if output_dir:
        log_dir = os.path.join(output_dir, method_name, "logs")
        os.makedirs(log_dir, exist_ok=True)
        csv_path = os.path.join(log_dir, f"{method_name}_eval_episodes.csv")
        with open(csv_path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["episode", "success", "collision", "timeout", "length", "return", "path_length", "time"])  # main.py:179
            w.writerows(rows)
    if log:
        log(f"EVALUATION SUMMARY  method={method_name} episodes={len(rows)} success={summary['success_rate'] * 100:.2f}% "
            f"collision={summary['collision_rate'] * 100:.2f}% timeout={summary['timeout_rate'] * 100:.2f}% "
            f"len={summary['mean_length']:.2f}+-{summary['std_length']:.2f} return={summary['mean_return']:.2f}+-{summary['std_return']:.2f}"
            + (f" csv={csv_path}" if csv_path else ""))
    summary["csv"] = csv_path
    summary["rows"] = rows
    return summary
