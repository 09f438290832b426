#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: env-steps/sec at 4096 envs per GPU (PPO end to end).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 4096 envs per GPU, stage_1 map (32 segments), 10 beams, PPO with the
2x64 MLP heads, rollout T=512, episode cap 500, 50 full-batch update epochs, lr 3e-4, clip 0.2, gamma 0.99.
One "step" = one PPO iteration = T env steps of all envs (policy forward + sampling + HIP step kernel,
captured in one hipGraph) + the HIP return scan + the 50-epoch update (+ one RCCL all-reduce of the flat
gradient per epoch when N > 1).  value = K * T * n_envs_total / wall, the reference's own
`perf/steps_per_sec` (project_ppo/src/ppo.py:855), all inputs resident in HBM, synthetic (random-init
policy, seeded goals).  Weak scaling: every GPU owns its own 4096-env shard.

Besides the contract fields, rank 0 adds
  roofline      step kernel on BASELINE configs[2] (16384 envs, per-env stage_2 segment buffers): algorithmic
                bytes (134 + 16*S per env-step) / mean launch duration from HIP events
  roofline_timed_region   the same kernel at the timed workload (4096 envs, shared map: launch-latency bound)
  cpu_baseline  the CPU oracle (scalar C port, 1 core) stepping the same 4096-env workload for ~10 s
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def _event_time_ms(fn, iters, warm=20, per_graph=64):
    """Mean duration of one `fn()` (one kernel launch on torch's current stream) from HIP events recorded on that
    stream.  The launches are replayed from a hipGraph of `per_graph` launches so the ~12 us python/ctypes launch path
    is not what gets measured (the kernels are shorter than that); back-to-back replays keep the queue full, so
    the mean includes the ~1.5 us kernel-to-kernel boundary, as in the real rollout graph."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(per_graph):
            fn()
    g.replay()
    torch.cuda.synchronize()
    reps = max(1, iters // per_graph)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * per_graph)


def step_kernel_roofline(n_envs, map_name, per_env, iters=640, seed=0):
    """HIP-event timing of navsim_step alone (random actions resident in HBM)."""
    from navbot_ppo_amd import maps
    from navbot_ppo_amd.env import NavSim
    seg = maps.by_name(map_name)
    S = int(seg.shape[0])
    sim = NavSim(n_envs, max_episode_steps=500, auto_reset=True, seed=seed)
    rr, rs = maps.goal_rects(map_name)
    sim.set_goal_rects(0, rr)
    sim.set_goal_rects(1, rs)
    sim.set_map(maps.replicate_per_env(seg, n_envs, seed=seed) if per_env else seg)
    io = sim.alloc_io()
    sim.reset(io.obs)
    g = torch.Generator(device="cuda").manual_seed(seed)
    acts = torch.rand((64, n_envs, 2), device="cuda", generator=g)
    acts[..., 1] = acts[..., 1] * 2 - 1
    k = [0]

    def launch():
        sim.step(acts[k[0] & 63], io.obs, io.reward, io.done, io.arrive, io.ended, io.ep_return, io.ep_length)
        k[0] += 1

    ms = _event_time_ms(launch, iters)
    bytes_per_env_step = 134 + (16 * S if per_env else 0)
    alg_bytes = n_envs * bytes_per_env_step + (0 if per_env else 16 * S)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    sim.close()
    return dict(bound="hbm", kernel="step_kernel<10,%s>" % ("per_env" if per_env else "shared"),
                workload=f"{n_envs} envs, {map_name} ({S} segments, {'per-env' if per_env else 'shared'} map), 10 beams",
                achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5),
                traffic=None, launch_us=round(ms * 1e3, 3), algorithmic_bytes_per_launch=int(alg_bytes),
                bytes_per_env_step=bytes_per_env_step, env_steps_per_sec=round(n_envs / (ms * 1e-3), 1))


def cpu_baseline(n_envs, budget_s=10.0):
    """The oracle (oracle/navsim_oracle.c, scalar C, one thread) on the same workload, bounded in time."""
    from navbot_ppo_amd import maps
    from oracle import navsim_oracle as O
    sim = O.OracleSim(n_envs, max_episode_steps=500, auto_reset=True, seed=0)
    sim.set_map(maps.stage_1())
    sim.reset()
    rng = np.random.default_rng(0)
    a = np.stack([rng.uniform(0, 1, n_envs), rng.uniform(-1, 1, n_envs)], 1).astype(np.float32)
    sim.step(a)
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s:
        sim.step(a)
        steps += 1
    dt = time.perf_counter() - t0
    return dict(value=round(steps * n_envs / dt, 1), unit="env-steps/s", cores=1, kind="port",
                sample=f"{steps} steps of {n_envs} envs (stage_1, 10 beams, random actions), env step only, {dt:.1f} s",
                host_cpus=os.cpu_count())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--rollout", type=int, default=512)
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--policy", default="mlp64x2")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline legs")
    args = ap.parse_args()

    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    ctx = ppo.DistCtx()
    if ctx.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ctx.world}: launch with torch.distributed.run")
    n_local = args.envs_per_gpu
    n_total = n_local * ctx.world
    lo, _ = ctx.shard(n_total)
    env = VecEnv(n_local, map="stage_1", n_beams=10, max_episode_steps=500, auto_reset=True, seed=0, env_id_base=lo,
                 device=ctx.device)
    cfg = ppo.PPOConfig(rollout_len=args.rollout, max_episode_steps=500, n_updates_per_iteration=args.epochs,
                        policy=args.policy, use_graph=not args.no_graph, seed=0)
    trainer = ppo.PPOTrainer(env, cfg, ctx)

    for _ in range(args.warmup):
        trainer.iteration()
    ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    roll_t = upd_t = 0.0
    for _ in range(args.steps):
        lg = trainer.iteration()
        roll_t += lg["rollout_time"]
        upd_t += lg["update_time"]
    ctx.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ctx.device)
    ctx.all_reduce_max(dt)
    dt = float(dt.item())

    out = None
    if ctx.rank == 0:
        K = args.steps
        out = {
            "metric": "env_steps_per_sec", "value": round(K * args.rollout * n_total / dt, 1), "unit": "env-steps/s",
            "n_gpus": ctx.world, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "dtype_detail": "pose/goal angles/reward f64, ray-cast f32, PPO nets f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {n_local} envs/GPU, stage_1 (32 segments), 10 beams, "
                                   f"PPO {args.policy}, rollout={args.rollout}, {args.epochs} full-batch epochs, episode cap 500",
                       "n_envs_total": n_total, "n_envs_per_gpu": n_local, "rollout_len": args.rollout,
                       "epochs": args.epochs, "policy": args.policy, "parallelism": f"env-shard dp{ctx.world}",
                       "hip_graph_rollout": not args.no_graph},
            "rollout_only_env_steps_per_sec": round(K * args.rollout * n_total / roll_t, 1),
            "rollout_ms": round(roll_t / K * 1e3, 3), "update_ms": round(upd_t / K * 1e3, 3),
            "last_iter": {k: lg[k] for k in ("avg_ep_rews", "success_rate", "episodes", "actor_loss", "critic_loss", "approx_kl")},
        }
    if not args.no_extras and ctx.rank == 0:   # rank 0 only (the other ranks wait at the barrier below)
        del trainer
        torch.cuda.empty_cache()
        out["roofline"] = step_kernel_roofline(16384, "stage_2", per_env=True)
        out["roofline_timed_region"] = step_kernel_roofline(n_local, "stage_1", per_env=False)
        pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            t = json.load(open(pmc))
            out["roofline"]["traffic"] = t.get("cfg3_step_bytes_per_launch")
            out["roofline_timed_region"]["traffic"] = t.get("cfg2_step_bytes_per_launch")
        if ctx.world == 1:
            out["cpu_baseline"] = cpu_baseline(n_local)
    if ctx.rank == 0:
        print(json.dumps(out), flush=True)
    ctx.barrier()
    if ctx.enabled:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
