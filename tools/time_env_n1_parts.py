#!/usr/bin/env python3
"""Dev tool: where the 27 us of one N = 1 `Env.step` from Python go (launch through ctypes / waiting for the stream / numpy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from navbot_ppo_amd.env import Env
env = Env(is_training=True); env.reset()
a, p = np.array([0.3, 0.1]), np.zeros(2)
n = 3000
for _ in range(300): env.step(a, p)
t0 = time.perf_counter()
for _ in range(n): env.step(a, p)
full = (time.perf_counter() - t0) / n * 1e6
# launch only (no wait), then one wait at the end
sim = env._sim
t0 = time.perf_counter()
for _ in range(n):
    sim.step(env._act_t, env._obs_t, env._rew_t, env._done_t, env._arrive_t, env._ended_t, None, None, past_action=env._past_t)
launch = (time.perf_counter() - t0) / n * 1e6
torch.cuda.synchronize()
# kernel time on the device: graph of 64 launches
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(64):
        sim.step(env._act_t, env._obs_t, env._rew_t, env._done_t, env._arrive_t, env._ended_t, None, None, past_action=env._past_t)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
kern = e0.elapsed_time(e1) / (20 * 64) * 1e3
# launch + wait, no numpy
st = torch.cuda.current_stream()
t0 = time.perf_counter()
for _ in range(n):
    sim.step(env._act_t, env._obs_t, env._rew_t, env._done_t, env._arrive_t, env._ended_t, None, None, past_action=env._past_t)
    st.synchronize()
lw = (time.perf_counter() - t0) / n * 1e6
print(f"Env.step {full:.1f} us = launch through ctypes {launch:.1f} (async, back to back) | launch + stream wait {lw:.1f} | kernel alone {kern:.1f} (graph replay, device) | numpy in / out {full - lw:.1f}")
# the same with a query spin instead of the blocking wait
t0 = time.perf_counter()
for _ in range(n):
    sim.step(env._act_t, env._obs_t, env._rew_t, env._done_t, env._arrive_t, env._ended_t, None, None, past_action=env._past_t)
    while not st.query():
        pass
print(f"launch + stream.query() spin {(time.perf_counter() - t0) / n * 1e6:.1f} us")
ev = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(n):
    sim.step(env._act_t, env._obs_t, env._rew_t, env._done_t, env._arrive_t, env._ended_t, None, None, past_action=env._past_t)
    ev.record(st)
    while not ev.query():
        pass
print(f"launch + event.query() spin {(time.perf_counter() - t0) / n * 1e6:.1f} us")
# the same with a spin on the results themselves: the pinned block is host-coherent, the kernel's stores land in it while it runs;
# a NaN planted in the reward and in the last observation entry (stored last, after barrier C) is overwritten when the step is done
pin = env._pin_np
nan = np.float32(np.nan)
t0 = time.perf_counter(); spins = 0
for _ in range(n):
    pin[32] = nan; pin[31] = nan
    env._call(env._lib.navsim_step, env._step_args, "navsim_step")
    while pin[31] != pin[31] or pin[32] != pin[32]:
        spins += 1
print(f"launch + spin on the pinned results {(time.perf_counter() - t0) / n * 1e6:.1f} us ({spins / n:.1f} polls per step)")
torch.cuda.synchronize()
