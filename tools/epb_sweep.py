#!/usr/bin/env python3
"""Dev tool: workgroup shape (NAVSIM_EPB = 8 / 16 / 32 / rule) x shard size x beam count on shared maps (stage_4 with 36 beams, stage_1 with
10; EPB_MAP=house: the 2048-segment house map with start / goal tables; EPB_MAP=per_env: per-env stage_2 maps), tape form and one launch per step, us per step."""
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from navbot_ppo_amd import maps
from navbot_ppo_amd.env import NavSim
N, B = int(sys.argv[1]), int(sys.argv[2])
HOUSE = os.environ.get("EPB_MAP") == "house"
PERENV = os.environ.get("EPB_MAP") == "per_env"   # per-env stage_2 maps (128 segments per env)
seg = maps.house(2048) if HOUSE else (maps.stage_4() if B == 36 else maps.stage_1())
sim = NavSim(N, n_beams=B, max_episode_steps=500, auto_reset=True, seed=0)
rr, rs = maps.goal_rects("stage_4" if B == 36 else "stage_1")
if not HOUSE: sim.set_goal_rects(0, rr); sim.set_goal_rects(1, rs)
if PERENV:
    rr, rs = maps.goal_rects("stage_2"); sim.set_goal_rects(0, rr); sim.set_goal_rects(1, rs)
    sim.set_map(maps.replicate_per_env(maps.stage_2(), N, seed=0), per_env=True)
else:
    sim.set_map(seg)
if HOUSE:
    st, g, lo, hi = maps.spawn_tables("small_house"); sim.set_spawn_sampler(*(maps.open_tables(seg, st, g) + (lo, hi)))
io = sim.alloc_io(); sim.reset(io.obs)
T = 32 if HOUSE else 128
acts = torch.rand((T, N, 2), device="cuda"); acts[..., 1] = acts[..., 1] * 2 - 1
obs = torch.zeros((T, N, sim.D), device="cuda"); rew = torch.zeros((T, N), device="cuda")
fl = [torch.zeros((T, N), dtype=torch.uint8, device="cuda") for _ in range(3)]
for _ in range(2): sim.step_seq(acts, obs, rew, *fl)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(8): sim.step_seq(acts, obs, rew, *fl)
e1.record(); torch.cuda.synchronize()
seq = e0.elapsed_time(e1) / (8 * T) * 1e3
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for k in range(64): sim.step(acts[k % T], obs[k % T], rew[k % T], fl[0][k % T], fl[1][k % T], fl[2][k % T])
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(8): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"{'house ' if HOUSE else 'per-env ' if PERENV else ''}N={N:6d} B={B} EPB={os.environ.get('NAVSIM_EPB','auto'):>4s}: tape {seq:6.2f} us/step | step launches {e0.elapsed_time(e1) / (8 * 64) * 1e3:6.2f} us")
'''
open("/tmp/epb_one.py", "w").write(code)
for B in ((10,) if os.environ.get("EPB_MAP") in ("house", "per_env") else (36, 10)):
    for N in (1024, 2048, 4096, 8192, 16384):
        for e in ("auto", "8", "16", "32"):
            env = dict(os.environ)
            if e != "auto": env["NAVSIM_EPB"] = e
            out = subprocess.run([sys.executable, "/tmp/epb_one.py", str(N), str(B)], env=env, capture_output=True, text=True)
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
