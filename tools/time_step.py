#!/usr/bin/env python3
"""Times navsim_step alone (HIP events) for a given library build.  Dev tool for kernel work.
usage: python tools/time_step.py [libpath] [--cfg3|--cfg2|--cfg5]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from navbot_ppo_amd import _native, maps
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
from navbot_ppo_amd.env import NavSim

def run(N, seg, per_env, iters=int(os.environ.get("TS_ITERS", "300")), B=10, rects=None, f16=False, sampler=None):
    sim = NavSim(N, n_beams=B, max_episode_steps=500, auto_reset=os.environ.get("TS_AUTORESET", "1") == "1", seed=0, obs_f16=f16)
    if rects and os.environ.get("TS_RECTS", rects) != "none":   # the goal rejection rectangles bench.py sets for its roofline legs
        rr, rs = maps.goal_rects(os.environ.get("TS_RECTS", rects))
        sim.set_goal_rects(0, rr); sim.set_goal_rects(1, rs)
    sim.set_map(seg, per_env=per_env)
    if sampler:
        sim.set_spawn_sampler(*sampler)
    io = sim.alloc_io(); sim.reset(io.obs)
    acts = torch.rand((64, N, 2), device="cuda"); acts[..., 1] = acts[..., 1] * 2 - 1
    def launch(k): sim.step(acts[k & 63], io.obs, io.reward, io.done, io.arrive, io.ended, io.ep_return, io.ep_length)
    for k in range(30): launch(k)
    torch.cuda.synchronize()
    # 64 launches captured in one hipGraph: python/ctypes launch overhead (~12 us) is out of the measurement
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(64): launch(k)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(1, iters // 64)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (reps * 64) * 1e3
    sim.close()
    return us

which = [a for a in sys.argv[1:] if a.startswith("--")] or ["--cfg3", "--cfg2"]
for w in which:
    if w == "--cfg3":
        seg = maps.replicate_per_env(maps.stage_2(), 16384, seed=0); us = run(16384, seg, True, rects="stage_2")
        print(f"cfg3 16384 envs per-env S=128: {us:8.2f} us  -> {16384*(134+16*128)/us/1e3:8.1f} GB/s")
    elif w == "--cfg2":
        us = run(4096, maps.stage_1(), False); print(f"cfg2 4096 envs shared S=32  : {us:8.2f} us")
    elif w == "--cfg4":
        us = run(4096, maps.stage_4(), False, B=36, rects="stage_4"); print(f"cfg4 4096 envs shared S=64 B=36: {us:8.2f} us")
    elif w.startswith("--s="):
        S = int(w[4:]); sides = (S - 32) // 4
        seg = maps.replicate_per_env(maps.stage_2(sides=sides), 16384, seed=0); us = run(16384, seg, True, rects="stage_2")
        print(f"16384 envs per-env S={seg.shape[1]}: {us:8.2f} us  -> {16384*(134+16*seg.shape[1])/us/1e3:8.1f} GB/s")
    elif w == "--cfg5":   # BASELINE configs[4] per GPU: 65536 / 8 envs, house map (~2k segments, shared), 10 beams
        seg = maps.house(2048); st, g, lo, hi = maps.spawn_tables("small_house")   # f16 observations, start / goal tables: bench.py's cfg5_shard
        us = run(8192, seg, False, f16=True, sampler=maps.open_tables(seg, st, g) + (lo, hi))
        print(f"cfg5 8192 envs shared S={seg.shape[0]} f16 + tables: {us:8.2f} us  -> {8192*seg.shape[0]*10/us/1e3:8.1f} G ray-segment tests/s")
    elif w == "--big":
        us = run(65536, maps.stage_1(), False); print(f"65536 envs shared S=32: {us:8.2f} us")
