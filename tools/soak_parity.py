import sys, os, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from navbot_ppo_amd import maps
from navbot_ppo_amd.env import NavSim
from oracle import navsim_oracle as O
def soak(N, seg, per_env, K, B=10, cap=60, seed=0, sampler=None, amax=1.0):
    gpu = NavSim(N, n_beams=B, max_episode_steps=cap, auto_reset=True, seed=seed)
    cpu = O.OracleSim(N, n_beams=B, max_episode_steps=cap, auto_reset=True, seed=seed)
    for s in (gpu, cpu):
        s.set_map(seg, per_env=per_env)
        if sampler: s.set_spawn_sampler(*sampler)
    io = gpu.alloc_io(); og = gpu.reset(io.obs).cpu().numpy(); oc = cpu.reset()
    assert np.abs(og - oc).max() <= 1e-6
    rng = np.random.default_rng(seed); bad = 0; exact = 0; tot = 0; ends = 0; mx = 0.0
    for k in range(K):
        a = np.stack([rng.uniform(0, amax, N), rng.uniform(-amax, amax, N)], 1).astype(np.float32)
        a[: N // 3, 1] *= 0.05
        gpu.step(torch.from_numpy(a).cuda(), io.obs, io.reward, io.done, io.arrive, io.ended)
        out = cpu.step(a)
        o = io.obs.cpu().numpy()
        d = np.abs(o - out["obs"]).max(); mx = max(mx, d)
        fl = (io.done.cpu().numpy() != out["done"]).sum() + (io.arrive.cpu().numpy() != out["arrive"]).sum() + (io.ended.cpu().numpy() != out["ended"]).sum()
        bad += int(fl) + int(d > 1e-6)
        exact += int((o == out["obs"]).all(1).sum()); tot += N; ends += int(out["ended"].sum())
        if fl or d > 1e-6:
            print("MISMATCH step", k, "maxdiff", d, "flags", fl); break
    print(f"N={N} S={seg.shape[-2]} per_env={per_env} B={B} steps={K}: bad={bad} max|dobs|={mx:.2e} exact rows {exact/tot:.5f} episode ends {ends}")
soak(4096, maps.replicate_per_env(maps.stage_2(), 4096, seed=1), True, 600)
soak(16384, maps.replicate_per_env(maps.stage_2(), 16384, seed=2), True, 80)
soak(4096, maps.stage_1(), False, 800)
st, g, lo, hi = maps.spawn_tables("small_house"); seg = maps.house(2048)
soak(2048, seg, False, 400, cap=40, sampler=maps.open_tables(seg, st, g) + (lo, hi))
soak(8192, seg, False, 60, cap=40, sampler=maps.open_tables(seg, st, g) + (lo, hi))
soak(2048, maps.stage_4(), False, 300, B=36)
soak(2048, maps.replicate_per_env(maps.stage_2(sides=56), 2048, seed=3), True, 300, amax=3.0)
