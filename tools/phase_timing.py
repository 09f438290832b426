#!/usr/bin/env python3
"""Dev tool: per-phase timestamps (wall_clock64, 100 MHz -> 10 ns ticks) from the instrumented build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from navbot_ppo_amd import _native, maps
CFG2 = "--cfg2" in sys.argv
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
_native.LIB_PATH = os.path.abspath(_args[0] if _args else "build/libnavsim_timing.so")
from navbot_ppo_amd.env import NavSim
names = ["start", "pre-A", "post-A", "pre-B", "post-B", "pre-C", "pose-seen", "end"]
if "--rollout" in sys.argv:   # the persistent rollout kernel: stamps of its LAST step + the whole launch
    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv
    T = 256
    BIG = "--big" in sys.argv   # configs[2] closed loop: 16384 envs, per-env stage_2 maps -> rollout_big_kernel (16 waves, 64 envs)
    if BIG:
        env = VecEnv(16384, map="stage_2", max_episode_steps=500, seed=0, per_env_map=True)
    elif "--beams36" in sys.argv:   # configs[3]'s shard: 4096 envs, stage_4, 36 beams -> rollout_kernel<36, 16, ., 8>
        env = VecEnv(4096, map="stage_4", n_beams=36, max_episode_steps=500, seed=0)
    else:
        env = VecEnv(4096, map="stage_1", max_episode_steps=500, seed=0)
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=T, policy="mlp64x2", n_updates_per_iteration=1))
    tr.rollout(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); tr._persistent_rollout(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / T
    buf = np.zeros(1024, dtype=np.int64)
    L = _native.lib(); L.navsim_dbg_read.argtypes = [C.c_void_p]; L.navsim_dbg_read(buf.ctypes.data_as(C.c_void_p))
    b = buf.reshape(8, 16, 8)
    print(f"persistent rollout: {us:.2f} us per step ({T} steps)")
    names = ["start", "pre-A", "post-A", "pre-B", "post-B", "pre-C", "hook-done", "end"]
    pol = np.zeros(256, dtype=np.int64)
    L.navsim_pol_read.argtypes = [C.c_void_p]; L.navsim_pol_read(pol.ctypes.data_as(C.c_void_p))
    pol = pol.reshape(8, 8, 4)
    for blk in range(3):   # stamps of the step before the last one that ran the hook (T - 2) are overwritten by T - 1: the last step has no hook
        NWV = 16 if BIG else 8
        t0 = b[blk, :NWV, 0].min()
        print(f"block sample {blk}: last step (ns after its first wave entered the step body)")
        for w in range(NWV):
            print("  wave", w, " ".join(f"{names[s]}={(b[blk, w, s] - t0) * 10:6d}" for s in range(8)),
                  *(() if BIG or w >= 8 else ("| tile part / noise done", (pol[blk, w, 1] - t0) * 10, "finish done (last arriver)", (pol[blk, w, 2] - t0) * 10, "left body", (pol[blk, w, 0] - t0) * 10)))
    sys.exit(0)
N = 4096 if CFG2 else 16384
EPBv = int(os.environ.get('NAVSIM_EPB', '64' if N >= 16384 else '32' if N >= 4096 else '16'))   # pick_epb's rule
sim = NavSim(N, max_episode_steps=500, auto_reset=True, seed=0)
if CFG2:
    sim.set_map(maps.stage_1(), per_env=False)
else:
    if os.environ.get("TS_RECTS", "stage_2") != "none":   # configs[2] as bench.py runs it: stage_2's goal rejection rectangles
        rr, rs = maps.goal_rects(os.environ.get("TS_RECTS", "stage_2"))
        sim.set_goal_rects(0, rr); sim.set_goal_rects(1, rs)
    sim.set_map(maps.replicate_per_env(maps.stage_2(), N, seed=0), per_env=True)
io = sim.alloc_io(); sim.reset(io.obs)
acts = torch.rand((N, 2), device="cuda"); acts[:, 1] = acts[:, 1] * 2 - 1
acts2 = torch.rand((N, 2), device="cuda"); acts2[:, 1] = acts2[:, 1] * 2 - 1
for k in range(int(os.environ.get("PT_WARM", "20"))): sim.step(acts if k % 2 == 0 else acts2, io.obs, io.reward, io.done, io.arrive, io.ended)
torch.cuda.synchronize()
sim.step(acts, io.obs, io.reward, io.done, io.arrive, io.ended); torch.cuda.synchronize()
buf = np.zeros(1024, dtype=np.int64)
L = _native.lib(); L.navsim_dbg_read.argtypes = [C.c_void_p]; L.navsim_dbg_read(buf.ctypes.data_as(C.c_void_p))
b = buf.reshape(8, 16, 8)  # [sampled block][wave][slot]
NWV = 16 if EPBv >= 64 else (8 if EPBv >= 32 else 4)
t0 = b[:, :NWV, 0][b[:, :NWV, 0] > 0].min()
for blk in range(2):
    print(f"block sample {blk}")
    for w in range(NWV):
        print("  wave", w, " ".join(f"{names[s]}={(b[blk, w, s] - t0) * 10:6d}ns" for s in range(8)))

EPB = EPBv
blk = np.zeros(8192 * 3, dtype=np.int64)
L.navsim_blk_read.argtypes = [C.c_void_p]; L.navsim_blk_read(blk.ctypes.data_as(C.c_void_p))
nb = (N + EPB - 1) // EPB
blk = blk.reshape(8192, 3)[:nb]
blk = blk[blk[:, 0] > 0]   # blocks that ran (a wrong EPB guess must not turn the time axis into hours)
st, en, hw = blk[:, 0], blk[:, 1], blk[:, 2]
t0 = st.min()
st = (st - t0) * 10; en = (en - t0) * 10
print("blocks", nb, "kernel span ns", en.max(), "mean block life ns", (en - st).mean())
xcc = (hw >> 32) & 0xf; cu = (hw & 0xffffffff) >> 8 & 0xf; se = (hw & 0xffffffff) >> 13 & 0x7
print("start time histogram (us):", np.histogram(st / 1000, bins=12)[0].tolist())
for tq in range(0, min(int(en.max()), 200000), 2500):
    print(f"  t={tq/1000:5.1f}us running blocks: {int(((st <= tq) & (en > tq)).sum())}")
print("xcc counts", np.bincount(xcc.astype(int)).tolist())
print("first 16 blocks xcc", xcc[:16].tolist(), "start", st[:16].tolist())
if "--per-xcd" in sys.argv:
    for q in range(8):
        m = xcc == q
        order = np.argsort(st[m])
        print(f"xcc {q}: starts(ns) {st[m][order][:40].tolist()}")
        print(f"        cu/se     {[(int(s), int(c)) for s, c in zip(se[m][order][:40], cu[m][order][:40])]}")
        print(f"        ends(ns)  {en[m][order][:40].tolist()}")
if "--simd" in sys.argv:
    simd = (hw & 0xffffffff) >> 4 & 0x3
    key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 8) | cu.astype(np.int64)
    import collections
    per_cu = collections.defaultdict(list)
    for k_, s_ in zip(key.tolist(), simd.tolist()): per_cu[k_].append(int(s_))
    print("SIMD of wave 0 of the workgroups sharing a CU (first 12 CUs):", [v for _, v in list(sorted(per_cu.items()))[:12]])
    print("wave-0 SIMD histogram:", np.bincount(simd.astype(int), minlength=4).tolist())
