#!/usr/bin/env python3
"""Dev tool: where do the persistent rollout (navsim_rollout_mlp64) and the per-step rollout differ?  usage: rollout_diff.py N T map sampler sens [force]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from navbot_ppo_amd import ppo
from navbot_ppo_amd.env import VecEnv
N, T, map_name, sampler, sens = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5] == "1"
force = len(sys.argv) > 6
kw = dict(lidar_noise_sigma=0.01, lidar_below_min="gazebo") if sens else {}
outs = []
for persistent in (True, False):
    env = VecEnv(N, map=map_name, max_episode_steps=30, seed=3, sampler=None if sampler == "none" else sampler, **kw)
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=T, max_episode_steps=30, n_updates_per_iteration=1, policy="mlp64x2", seed=5,
                                           persistent_rollout=persistent, use_graph=False))
    if persistent and force:   # the persistent kernel whatever ppo.py's rule says
        tr._decay_exploration(); tr.epret_buf.zero_(); tr.eplen_buf.zero_(); env.sim.reset(tr.obs_buf[0]); tr._persistent_rollout()
    else:
        tr.rollout()
    torch.cuda.synchronize()
    outs.append({k: getattr(tr, k + "_buf").cpu().numpy().copy() for k in ("obs", "act", "logp", "rew", "done", "arrive", "ended")})
    env.close()
a, b = outs
tag = f"N={N} T={T} {map_name} sampler={sampler} sens={sens} EPB={os.environ.get('NAVSIM_EPB', 'auto')} force={force}"
first = None
for k in a:
    d = a[k] != b[k]
    if d.any():
        idx = np.argwhere(d)
        t0 = idx[:, 0].min()
        at = idx[idx[:, 0] == t0]
        print(tag, "DIFF", k, "count", int(d.sum()), "first row", int(t0), "envs", sorted(set(at[:, 1].tolist()))[:8], "cols", sorted(set(at[:, 2].tolist())) if at.shape[1] > 2 else "")
        if first is None or t0 < first[0]: first = (t0, k, at)
if first is None:
    print(tag, "IDENTICAL; ended", int(a["ended"].sum()))
else:
    t0, k, at = first
    e = int(at[0, 1])
    print(" first diff at row", int(t0), k, "env", e, "persistent", a[k][t0, e], "per-step", b[k][t0, e])
    if k == "obs":
        print("  ended at row", t0 - 1, ":", a["ended"][t0 - 1, e], b["ended"][t0 - 1, e], " prev obs equal:", (a["obs"][t0 - 1, e] == b["obs"][t0 - 1, e]).all())
