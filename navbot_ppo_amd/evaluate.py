"""Evaluation mode: the reference's ``main.evaluate`` (project_ppo/src/main.py:135-252) on the batched simulator.

Loads the newest ``actor_iter*_step*.pth`` (or an explicit path), runs ``num_episodes`` episodes with the
DETERMINISTIC mean action (no sampling, main.py:197-199) and the evaluation arrival threshold 0.4
(``is_training=False``, environment_new.py:44-47), and writes the reference's per-episode CSV
``<output_dir>/<method>/logs/<method>_eval_episodes.csv`` with its columns
``episode,success,collision,timeout,length,return,path_length,time`` (main.py:179) plus the summary it prints
(main.py:236-250).  Episodes run ``n_parallel`` at a time; each env plays a fixed quota of whole episodes back to back; the
``time`` column is the episode's share of the wall clock (length x mean step time).

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
    """Every env plays a FIXED quota of q = ceil(num_episodes / n_parallel) whole episodes (its first q; later ones are
    ignored) and stepping continues until every env has met it, so which episodes are reported does not depend on how
    long they last -- taking "the first num_episodes to finish" would over-sample short episodes (early collisions) and
    under-report timeouts.  Rows are ordered episode-slot-major (slot 0 of every env, then slot 1, ...) and cut to
    num_episodes.  Everything stays on the device; the host looks at a counter every 16 steps."""
    n_par = int(n_parallel or min(num_episodes, 1024))
    quota = -(-int(num_episodes) // n_par)
    env = VecEnv(n_par, map=map, max_episode_steps=max_timesteps_per_episode, auto_reset=True, is_training=False, seed=seed,
                 device=device)
    dev = env.device
    actor = actor.to(dev).eval()
    obs = env.reset()
    count = torch.zeros(n_par, dtype=torch.int64, device=dev)
    res = torch.zeros((quota, n_par, 6), dtype=torch.float64, device=dev)   # success, collision, timeout, length, return, path
    env_ids = torch.arange(n_par, device=dev)
    t0, steps = time.time(), 0
    max_steps = quota * max_timesteps_per_episode + 1
    while steps < max_steps:
        action = actor(obs.float())                                   # deterministic mean action, main.py:197-199
        obs, rew, done, arrive = env.step(action)
        io = env.io
        take = io.ended.bool() & (count < quota)
        a, d = arrive.bool(), done.bool()
        ln = io.ep_length.double()
        row = torch.stack([a.double(), (d & ~a).double(),
                           (~d & ~a & (io.ep_length >= max_timesteps_per_episode)).double(),      # main.py:214-216
                           ln, io.ep_return.double(), io.ep_path.double()], 1)                    # path: main.py:200-204
        slot = torch.clamp(count, max=quota - 1)
        cur = res[slot, env_ids]
        res[slot, env_ids] = torch.where(take[:, None], row, cur)
        count += take.long()
        steps += 1
        if steps % 16 == 0 and bool((count >= quota).all()):
            break
    torch.cuda.synchronize(dev)
    sec_per_step = (time.time() - t0) / max(steps, 1)
    env.close()
    flat = res.reshape(quota * n_par, 6)[:num_episodes].cpu().numpy()
    rows = [[k, int(r[0]), int(r[1]), int(r[2]), int(r[3]), float(r[4]), float(r[5]), float(r[3]) * sec_per_step]
            for k, r in enumerate(flat)]
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
