#!/usr/bin/env python3
"""Dev tool: steps/s of the N=1 drop-in `Env` (one launch + one stream wait per step), the surface project_ppo/src/ppo.py drives."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from navbot_ppo_amd.env import Env
env = Env(is_training=True)
obs = env.reset()
rng = np.random.default_rng(0)
past = [0.0, 0.0]
for k in range(200):
    a = [rng.uniform(0, 1), rng.uniform(-1, 1)]
    obs, r, d, ar = env.step(a, past); past = a
    if d or ar: obs = env.reset(); past = [0.0, 0.0]
t0 = time.perf_counter(); n = 0; resets = 0
while time.perf_counter() - t0 < 3.0:
    a = [rng.uniform(0, 1), rng.uniform(-1, 1)]
    obs, r, d, ar = env.step(a, past); past = a; n += 1
    if d or ar: obs = env.reset(); past = [0.0, 0.0]; resets += 1
dt = time.perf_counter() - t0
print(f"Env (N=1 drop-in) step(): {n/dt:.0f} steps/s = {dt/n*1e6:.1f} us per step incl. {resets} resets")
