#!/usr/bin/env python3
"""Dev tool: navsim_step_seq (all steps of an action tape in one persistent launch) against navsim_step (one launch per step, 64
per hipGraph replay) on the same workloads.  usage: python tools/time_step_seq.py [lib.so] [--cfg3 --cfg2 --s=1024 --cfg5 --cfg4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from navbot_ppo_amd import _native, maps
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
from navbot_ppo_amd.env import NavSim

def run(N, seg, per_env, T=int(os.environ.get("TS_T", "256")), B=10, rects=None, f16=False, sampler=None):
    sim = NavSim(N, n_beams=B, max_episode_steps=500, auto_reset=True, seed=0, obs_f16=f16)
    if rects:
        rr, rs = maps.goal_rects(rects)
        sim.set_goal_rects(0, rr); sim.set_goal_rects(1, rs)
    sim.set_map(seg, per_env=per_env)
    if sampler:
        sim.set_spawn_sampler(*sampler)
    io = sim.alloc_io(); sim.reset(io.obs)
    dev = sim.device
    acts = torch.rand((T, N, 2), device=dev); acts[..., 1] = acts[..., 1] * 2 - 1
    obs = torch.zeros((T, N, sim.D), dtype=sim.obs_dtype, device=dev)
    rew = torch.zeros((T, N), device=dev)
    done, arrive, ended = (torch.zeros((T, N), dtype=torch.uint8, device=dev) for _ in range(3))
    epr = torch.zeros((T, N), device=dev); epl = torch.zeros((T, N), dtype=torch.int32, device=dev)
    for _ in range(2): sim.step_seq(acts, obs, rew, done, arrive, ended, epr, epl)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = int(os.environ.get("TS_REPS", "16"))   # (the rocprofv3 average of a run includes its 2 warm-up launches: keep them a small share)
    e0.record()
    for _ in range(reps): sim.step_seq(acts, obs, rew, done, arrive, ended, epr, epl)
    e1.record(); torch.cuda.synchronize()
    us_seq = e0.elapsed_time(e1) / (reps * T) * 1e3
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(64): sim.step(acts[k % T], obs[k % T], rew[k % T], done[k % T], arrive[k % T], ended[k % T], epr[k % T], epl[k % T])
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(8): g.replay()
    e1.record(); torch.cuda.synchronize()
    us_step = e0.elapsed_time(e1) / (8 * 64) * 1e3
    sim.close()
    return us_seq, us_step

which = [a for a in sys.argv[1:] if a.startswith("--")] or ["--cfg3", "--cfg2"]
for w in which:
    if w == "--cfg3":
        S = 128; seg = maps.replicate_per_env(maps.stage_2(), 16384, seed=0); a, b = run(16384, seg, True, rects="stage_2")
        print(f"cfg3 16384 envs per-env S=128: step_seq {a:7.2f} us/step = {16384*(134+16*S)/a/1e3:7.1f} GB/s | step launches {b:7.2f} us = {16384*(134+16*S)/b/1e3:7.1f} GB/s", flush=True)
    elif w == "--cfg2":
        a, b = run(4096, maps.stage_1(), False); print(f"cfg2 4096 envs shared S=32  : step_seq {a:7.2f} us/step | step launches {b:7.2f} us", flush=True)
    elif w == "--cfg4":
        a, b = run(4096, maps.stage_4(), False, B=36, rects="stage_4", T=128); print(f"cfg4 4096 envs shared S=64 B=36: step_seq {a:7.2f} us/step | step launches {b:7.2f} us", flush=True)
    elif w.startswith("--s="):
        S = int(w[4:]); seg = maps.replicate_per_env(maps.stage_2(sides=(S - 32) // 4), 16384, seed=0); S = seg.shape[1]
        a, b = run(16384, seg, True, T=64 if S <= 1024 else 32, rects="stage_2")
        print(f"16384 envs per-env S={S}: step_seq {a:7.2f} us/step = {16384*(134+16*S)/a/1e3:7.1f} GB/s | step launches {b:7.2f} us = {16384*(134+16*S)/b/1e3:7.1f} GB/s", flush=True)
    elif w == "--cfg5":
        seg = maps.house(2048); st, g, lo, hi = maps.spawn_tables("small_house")
        a, b = run(8192, seg, False, f16=True, sampler=maps.open_tables(seg, st, g) + (lo, hi), T=64)
        print(f"cfg5 8192 envs shared S={seg.shape[0]} f16: step_seq {a:7.2f} us/step | step launches {b:7.2f} us", flush=True)
