import os, sys, subprocess
sys.path.insert(0, os.getcwd())
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from navbot_ppo_amd import maps
from navbot_ppo_amd.env import NavSim
N, B = int(sys.argv[1]), int(sys.argv[2])
seg = maps.stage_4() if B == 36 else maps.stage_1()
sim = NavSim(N, n_beams=B, max_episode_steps=500, auto_reset=True, seed=0)
rr, rs = maps.goal_rects("stage_4" if B == 36 else "stage_1"); sim.set_goal_rects(0, rr); sim.set_goal_rects(1, rs)
sim.set_map(seg); io = sim.alloc_io(); sim.reset(io.obs)
T = 128
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
    for k in range(64): sim.step(acts[k], obs[k], rew[k], fl[0][k], fl[1][k], fl[2][k])
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(8): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"N={N:6d} B={B} EPB={os.environ.get('NAVSIM_EPB','auto'):>4s}: tape {seq:6.2f} us/step | step launches {e0.elapsed_time(e1) / (8 * 64) * 1e3:6.2f} us")
'''
open("/tmp/epb_one.py", "w").write(code)
for B in (36, 10):
    for N in (1024, 2048, 4096, 8192, 16384):
        for e in ("auto", "8", "16", "32"):
            env = dict(os.environ)
            if e != "auto": env["NAVSIM_EPB"] = e
            out = subprocess.run([sys.executable, "/tmp/epb_one.py", str(N), str(B)], env=env, capture_output=True, text=True)
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
