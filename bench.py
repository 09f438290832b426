#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: env-steps/sec at 4096 envs per GPU (PPO end to end).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 4096 envs per GPU, stage_1 map (32 segments), 10 beams, PPO with the
2x64 MLP heads, rollout T=512, episode cap 500, 50 full-batch update epochs, lr 3e-4, clip 0.2, gamma 0.99.
One "step" = one PPO iteration = T env steps of all envs (policy forward + sampling + env step, all T of them in ONE
persistent HIP launch) + the HIP return scan + the 50-epoch update (two launches per epoch; + one RCCL all-reduce of the
flat gradient per epoch when N > 1).  value = K * T * n_envs_total / wall, the reference's own
`perf/steps_per_sec` (project_ppo/src/ppo.py:855), all inputs resident in HBM, synthetic (random-init
policy, seeded goals).  Default: weak scaling, every GPU owns its own 4096-env shard (no data-path collective; the units
are independent envs).  `--scaling strong` keeps 4096 envs IN TOTAL (SURVEY.md 8d(i) read literally): shards of 4096 / N.

Besides the contract fields, rank 0 adds
  roofline            the ray-cast run of BASELINE configs[2] (16384 envs, per-env stage_2 segment buffers, S=128) through the entry
                      point the north star names: navsim_step (step_kernel), ONE launch per env step, 64 launches per graph replay:
                      algorithmic bytes (134 + 16*S per env-step) x envs / mean launch duration from HIP events.
                      (Rounds 1-2 reported this kernel here; round 3 had put the open-loop tape under this key.)
  roofline_closed_loop       the same step body with the mlp64 policy in the kernel (navsim_rollout_mlp64 at 16384 envs:
                      rollout_big_kernel, 256 steps per launch): what PPO.rollout does, the form a trainer can use
  roofline_open_loop_tape    the same step body, one launch per 256-step action tape (navsim_step_seq): replay / evaluation form only
  roofline_at_l3 (+ _closed_loop, _open_loop_tape)   the same three with S=1024 per env: 268 MB per step = the Infinity Cache size
  roofline_hbm (+ _closed_loop, _open_loop_tape)     the same three with S=2048 per env: 537 MB per step = 2x the Infinity Cache, the
                      HBM-bound regime; `frac` of the 8 TB/s spec, `frac_of_achievable_hbm` of the ~6.3 TB/s the guide calls achievable
  cfg4_ppo_shard / cfg5_ppo_shard / cfg5_ppo_shard_resmlp512   the same shards as PPO workloads end to end (rollout + 50 epochs) per GPU on
                      their own row formats: 42-D float32 rows, 16-D float16 rows; the last one with the reference's ACTIVE 512-wide nets
  cfg4_shard / cfg5_shard    one GPU's shard of BASELINE configs[3] (4096 envs, stage_4, 36 beams) and configs[4] (8192 envs, 2048-segment
                      house map, f16 observations, start / goal tables): us per step (launch per step / tape), env-steps/s
  traffic_source      which rocprofv3 --pmc file the `traffic` fields come from (--with-pmc-file: the same gpurun call as this run)
  roofline_timed_region   the persistent rollout kernel of the timed workload
  update_roofline     the update kernels of the timed workload (94 % of the timed region).  Split-bf16 pass: EXECUTED bf16 MFMA FLOPs of one
                      epoch / its duration against the dense bf16 MFMA peak (the unit it runs on), with the algorithmic float32 FLOP against
                      the f32-input MFMA peak beside it (`f32_equivalent`); f32 pass: algorithmic FLOP against the f32-input MFMA peak
  time_to_reward_s    PPO wall-clock until mean episode return >= +100 (ppo.py:833) from a fresh policy
  resmlp512           the same iteration with the reference's active 512-wide residual nets (fused f32-MFMA kernels of
                      csrc/ppo_resmlp512.hip) + `update_roofline`: MFMA FLOPs of one epoch / its duration vs the 157.3 TF peak
  env_n1_step_us      one step of the N = 1 drop-in class `Env` driven from Python like PPO.rollout drives the reference's
  cpu_baseline / _all_cores / _n1   the CPU oracle (scalar C port) on 1 core, on every host core, and one env per call
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


# ---- the same legs under rocprofv3 (one source for every roofline number).  HIP events around back-to-back
# launches overlap a kernel's tail with the next one's ramp and run un-profiled clocks: they sat 3-6 % above what a committed
# `rocprofv3 --kernel-trace --stats` summary reproduces.  So rank 0 profiles ITSELF: the extras block re-runs every roofline leg in a child
# process under `rocprofv3 --kernel-trace` (NAVBOT_BENCH_PLAIN=1: plain launches, a marker kernel between the legs) and takes each leg's
# per-kernel duration from the dispatch trace -- `frac`, `achieved`, `launch_us` come from THAT, the HIP-event figures stay beside them as
# `*_hip_events`.  If rocprofv3 cannot run, the HIP-event figures are printed and `time_source` says so.
PLAIN = os.environ.get("NAVBOT_BENCH_PLAIN") == "1"
_MANIFEST = []


def _marker():
    from navbot_ppo_amd.env import odometry
    torch.cuda.synchronize()
    z = torch.zeros(1, dtype=torch.float64, device="cuda")
    odometry(z, z, torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=torch.float64, device="cuda"), torch.zeros((1, 2), dtype=torch.float64, device="cuda"))
    torch.cuda.synchronize()


def _plain(tag, match, launch, n, per=None, graph=0):
    """PLAIN mode: a marker launch (navsim_odometry, used by no leg) in front of and behind n launches of the leg -- replayed from a hipGraph
    of `graph` launches where the HIP-event measurement replays one (the step kernels: back-to-back dispatches, as in a rollout graph);
    the manifest says which kernel names belong to the leg and how many units (launches / epochs) the summed duration is divided by."""
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(graph):
                launch()
        g.replay()
        _marker()
        for _ in range(max(1, n // graph)):
            g.replay()
        n = max(1, n // graph) * graph
    else:
        _marker()
        for _ in range(n):
            launch()
    _marker()
    _MANIFEST.append(dict(leg=tag, match=list(match), per=int(per if per is not None else n)))
    return float("nan")


def rocprof_legs(argv):
    """Runs `bench.py --kernel-trace-legs` under rocprofv3 --kernel-trace and returns {leg: {"us": mean kernel time per unit, "n": dispatches}}
    (None + reason if that is not possible here)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="navbot_rocprof_", dir="/tmp")
    env = dict(os.environ, NAVBOT_BENCH_PLAIN="1", TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NAVBOT_DIST_FORCE"):
        env.pop(k, None)
    cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
           "--kernel-trace-legs"] + list(argv)
    try:
        r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("MANIFEST ")]
        files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not lines or not files:
            return None, f"rocprofv3 run failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
        manifest = json.loads(lines[-1][len("MANIFEST "):])
        rows = list(csv.DictReader(open(files[0])))
        rows.sort(key=lambda x: int(x["Start_Timestamp"]))
        marks, segs = 0, {}   # leg k = the dispatches between marker 2 k and marker 2 k + 1 (warm-up launches sit outside)
        for x in rows:
            name = x["Kernel_Name"]
            if "odometry_kernel" in name:
                marks += 1
                continue
            if marks % 2 == 1:
                segs.setdefault(marks // 2, []).append((name, int(x["End_Timestamp"]) - int(x["Start_Timestamp"])))
        out = {}
        for k, m in enumerate(manifest):
            durs = [dur for name, dur in segs.get(k, []) if any(t in name for t in m["match"])]
            if durs:
                names = sorted({name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-70:]
                                for name, dur in segs.get(k, []) if any(t in name for t in m["match"])})
                out[m["leg"]] = dict(us=sum(durs) / 1e3 / m["per"], n=len(durs), min_us=min(durs) / 1e3, max_us=max(durs) / 1e3, per=m["per"], kernels=names)
        save = os.environ.get("NAVBOT_BENCH_SAVE_PROF")   # (tools/prof_r06.sh: the per-leg summary the line was computed from, for profiles/)
        if save:
            with open(save, "w") as f:
                f.write("leg,dispatches,units,mean_us_per_unit,min_dispatch_us,max_dispatch_us,kernels\n")
                for leg, v in out.items():
                    f.write(f"{leg},{v['n']},{v['per']},{v['us']:.3f},{v['min_us']:.3f},{v['max_us']:.3f},\"{' | '.join(v['kernels'])}\"\n")
        return out, f"rocprofv3 --kernel-trace of this command's own legs ({len(rows)} dispatches)"
    except Exception as e:   # noqa: BLE001 -- the bench line must still be printed
        return None, f"rocprofv3 run failed: {e!r}"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def apply_rocprof(leg, prof, key):
    """A leg's `frac` / `achieved` (+ launch_us / us_per_step / epoch time) from the rocprofv3 kernel time of the same leg; the HIP-event
    figures move to *_hip_events."""
    if not leg or not prof or key not in prof:
        if leg is not None:
            leg["time_source"] = "hip events (no rocprofv3 figure for this leg)"
        return leg
    us = prof[key]["us"]
    tkey = "launch_us" if "launch_us" in leg else "epoch_us" if "epoch_us" in leg else "epoch_ms" if "epoch_ms" in leg else None
    if tkey is None:
        return leg
    old_us = leg[tkey] * (1e3 if tkey == "epoch_ms" else 1.0)
    scale = old_us / us
    leg["frac_hip_events"], leg["achieved_hip_events"], leg[tkey + "_hip_events"] = leg["frac"], leg["achieved"], leg[tkey]
    leg["frac"] = round(leg["frac"] * scale, 5)
    leg["achieved"] = round(leg["achieved"] * scale, 2)
    leg[tkey] = round(us / (1e3 if tkey == "epoch_ms" else 1.0), 3)
    if "us_per_step" in leg and "steps_per_launch" in leg:
        leg["us_per_step"] = round(us / leg["steps_per_launch"], 3)
    if "env_steps_per_sec" in leg:
        leg["env_steps_per_sec"] = round(leg["env_steps_per_sec"] * scale, 1)
    for sub in ("f32_equivalent", "executed"):
        if isinstance(leg.get(sub), dict):
            leg[sub]["frac"] = round(leg[sub]["frac"] * scale, 4)
            leg[sub]["achieved"] = round(leg[sub]["achieved"] * scale, 2)
    if "frac_of_achievable_hbm" in leg:
        leg["frac_of_achievable_hbm"] = round(leg["frac_of_achievable_hbm"] * scale, 5)
    leg["time_source"] = f"rocprofv3 --kernel-trace, live ({prof[key]['n']} dispatches)"
    return leg


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


HBM_ACHIEVABLE_GBS = 6300.0  # same guide, HBM section: "8 TB/s peak (spec); ~6.3 TB/s achievable"


class CastWorkload:
    """One ray-cast workload (map + shard size + options), built once and shared by the legs that time it (one launch per step,
    one launch per action tape, closed loop): replicating a per-env map on the host is the slow part at S = 2048."""

    def __init__(self, n_envs, map_name, per_env, sides=None, n_beams=10, obs_f16=False, sampler=None, seed=0, house_segments=None):
        from navbot_ppo_amd import maps
        self.n_envs, self.map_name, self.per_env, self.B, self.obs_f16, self.seed = n_envs, map_name, per_env, n_beams, obs_f16, seed
        if house_segments:
            seg = maps.house(house_segments)
        else:
            seg = maps.stage_2(sides=sides) if sides else maps.by_name(map_name)
        self.S = int(seg.shape[0])
        self.rects = maps.goal_rects(map_name)
        self.sampler = None
        if sampler:
            st, g, lo, hi = maps.spawn_tables(sampler)
            self.sampler = maps.open_tables(seg, st, g) + (lo, hi)
        if per_env:
            self.seg = torch.from_numpy(maps.replicate_per_env(seg, n_envs, seed=seed)).cuda()
        else:
            self.seg = torch.from_numpy(np.ascontiguousarray(seg)).cuda()
        # reads 44 B + writes 26 + 4 (B + 6) B per env-step (f16 observations: 26 + 2 (B + 6)); B = 10, f32: 134 B;
        # + 16 S with a per-env map
        self.bytes_per_env_step = 70 + (2 if obs_f16 else 4) * (n_beams + 6) + (16 * self.S if per_env else 0)
        assert n_beams != 10 or obs_f16 or self.bytes_per_env_step == 134 + (16 * self.S if per_env else 0)

    def alg_bytes(self, steps):
        return steps * self.n_envs * self.bytes_per_env_step + (0 if self.per_env else 16 * self.S)

    def sim(self):
        from navbot_ppo_amd.env import NavSim
        sim = NavSim(self.n_envs, n_beams=self.B, max_episode_steps=500, auto_reset=True, seed=self.seed, obs_f16=self.obs_f16)
        sim.set_goal_rects(0, self.rects[0])
        sim.set_goal_rects(1, self.rects[1])
        sim.set_map(self.seg, per_env=self.per_env)
        if self.sampler:
            sim.set_spawn_sampler(*self.sampler)
        return sim

    def describe(self, steps=None):
        return (f"{self.n_envs} envs" + (f" x {steps} steps" if steps else "") + f", {self.map_name} ({self.S} segments, "
                f"{'per-env' if self.per_env else 'shared'} map), {self.B} beams" + (", f16 observations" if self.obs_f16 else ""))

    def leg(self, kernel, ms, steps, detail, **extra):
        alg = self.alg_bytes(steps)
        ach = alg / (ms * 1e-3) / 1e9
        d = dict(bound="hbm", bound_detail=detail, kernel=kernel, workload=self.describe(steps if steps > 1 else None),
                 achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                 traffic=None, launch_us=round(ms * 1e3, 3),
                 algorithmic_bytes_per_launch=int(alg), bytes_per_env_step=self.bytes_per_env_step,
                 env_steps_per_sec=round(steps * self.n_envs / (ms * 1e-3), 1))
        if self.per_env and self.n_envs * self.S * 16 > 1.25 * (256 << 20):
            # only where every segment byte of a step comes from HBM (the stream exceeds the Infinity Cache): of the ~6.3 TB/s the
            # guide calls achievable.  On a cache-fed working set the quotient is not a fraction of anything (it exceeded 1).
            d["frac_of_achievable_hbm"] = round(ach / HBM_ACHIEVABLE_GBS, 5)
        if steps > 1:
            d.update(steps_per_launch=steps, us_per_step=round(ms * 1e3 / steps, 3))
        d.update(extra)
        return d


def step_kernel_roofline(w, iters=640, detail="", tag=None):
    """HIP-event timing of navsim_step alone -- ONE launch per env step, the entry point a policy outside the kernel drives
    (Env.step, environment_new.py:272-310) -- with random actions resident in HBM."""
    sim = w.sim()
    io = sim.alloc_io()
    sim.reset(io.obs)
    g = torch.Generator(device="cuda").manual_seed(w.seed)
    acts = torch.rand((64, w.n_envs, 2), device="cuda", generator=g)
    acts[..., 1] = acts[..., 1] * 2 - 1
    k = [0]

    def launch():
        sim.step(acts[k[0] & 63], io.obs, io.reward, io.done, io.arrive, io.ended, io.ep_return, io.ep_length)
        k[0] += 1

    ms = _plain(tag, ["::step_kernel<"], launch, 320, graph=64) if PLAIN else _event_time_ms(launch, iters)
    sim.close()
    return w.leg("step_kernel<%d beams,%s> (navsim_step: one launch per step)" % (w.B, "per_env" if w.per_env else "shared"), ms, 1, detail)


def step_seq_roofline(w, T, reps=12, detail="", tag=None):
    """HIP-event timing of navsim_step_seq: T steps of a random action tape (resident in HBM) per launch, the env state on chip
    between the steps.  Algorithmic bytes per launch = T x n_envs x (134 + 16 S) (SURVEY.md 8(d) per env-step)."""
    sim = w.sim()
    n_envs = w.n_envs
    io = sim.alloc_io()
    sim.reset(io.obs)
    g = torch.Generator(device="cuda").manual_seed(w.seed)
    acts = torch.rand((T, n_envs, 2), device="cuda", generator=g)
    acts[..., 1] = acts[..., 1] * 2 - 1
    obs = torch.zeros((T, n_envs, sim.D), dtype=sim.obs_dtype, device="cuda")
    rew, epr = torch.zeros((T, n_envs), device="cuda"), torch.zeros((T, n_envs), device="cuda")
    done, arrive, ended = (torch.zeros((T, n_envs), dtype=torch.uint8, device="cuda") for _ in range(3))
    epl = torch.zeros((T, n_envs), dtype=torch.int32, device="cuda")
    launch = lambda: sim.step_seq(acts, obs, rew, done, arrive, ended, epr, epl)
    for _ in range(2):
        launch()
    torch.cuda.synchronize()
    if PLAIN:
        ms = _plain(tag, ["::steps_kernel<"], launch, min(reps, 6))
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    sim.close()
    return w.leg("steps_kernel<%d beams,%s> (navsim_step_seq: %d steps per launch, same step body as step_kernel)"
                 % (w.B, "per_env" if w.per_env else "shared", T), ms, T, detail)


def cpu_baseline(n_envs, procs, budget_s):
    """The oracle (oracle/navsim_oracle.c, scalar C) on the configs[1] env workload, bounded in time, in its own process
    (oracle/cpu_bench.py forks its worker pool from an interpreter that never touched HIP)."""
    import subprocess
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--procs", str(procs), "--envs", str(n_envs),
                          "--budget", str(budget_s)], cwd=REPO, capture_output=True, text=True, timeout=60 + 4 * budget_s)
    if out.returncode != 0:
        return {"error": out.stderr[-400:]}
    return json.loads(out.stdout.strip().splitlines()[-1])


def closed_loop_roofline(w, T, reps=8, detail="", tag=None):
    """HIP-event timing of navsim_rollout_mlp64 at a configs[2]-sized shard (rollout_big_kernel): the ray-cast run CLOSED-LOOP -- the
    16-64-64 actor chooses every action from the observation the previous step left on chip (PPO.rollout, ppo.py:505-594), T steps
    per launch.  Algorithmic bytes per launch = T x n_envs x (134 + 16 S), the env-step figure of SURVEY.md 8(d): the 12 bytes of
    action + log-prob the policy adds per env-step are not counted."""
    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv
    env = VecEnv(w.n_envs, map=w.seg, max_episode_steps=500, seed=w.seed)
    env.sim.set_goal_rects(0, w.rects[0])
    env.sim.set_goal_rects(1, w.rects[1])
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(policy="mlp64x2", rollout_len=T, max_episode_steps=500, seed=w.seed))
    assert tr.updater.fused_mlp64
    env.sim.reset(tr.obs_buf[0])
    for _ in range(2):
        tr._persistent_rollout()
    torch.cuda.synchronize()
    if PLAIN:
        ms = _plain(tag, ["::rollout_big_kernel<", "::rollout_kernel<"], tr._persistent_rollout, min(reps, 6))
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            tr._persistent_rollout()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    env.close()
    d = w.leg("rollout_big_kernel<64 envs, 16 waves, %s> (navsim_rollout_mlp64: %d steps per launch, policy phase + the step "
              "body of step_kernel)" % ("per_env" if w.per_env else "shared", T), ms, T, detail)
    d["workload"] += ", 16-64-64 policy in-kernel"
    return d


def rollout_kernel_leg(trainer, reps=6, tag=None):
    """The persistent rollout kernel of the timed workload alone (HIP events on its stream)."""
    T, N = trainer.cfg.rollout_len, trainer.env.N
    trainer.env.sim.reset(trainer.obs_buf[0])
    trainer._persistent_rollout()
    torch.cuda.synchronize()
    if PLAIN:
        ms = _plain(tag, ["::rollout_kernel<", "::rollout_big_kernel<"], trainer._persistent_rollout, reps)
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            trainer._persistent_rollout()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    alg = T * N * 134 + 16 * 32           # 134 B per env-step with a shared map (+ the map once)
    ach = alg / (ms * 1e-3) / 1e9
    return dict(bound="hbm", bound_detail="latency chain: one workgroup per CU runs T dependent steps (policy MFMA -> f64 motion -> "
                "cast -> rules); bytes are irrelevant at this size", kernel="rollout_kernel<16,8 waves>",
                workload=f"{N} envs x {T} steps, stage_1 (32 segments, shared map), 10 beams, 16-64-64 policy in-kernel",
                achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5), traffic=None,
                launch_us=round(ms * 1e3, 1), steps_per_launch=T, us_per_step=round(ms * 1e3 / T, 3), algorithmic_bytes_per_launch=int(alg),
                env_steps_per_sec=round(T * N / (ms * 1e-3), 1))


def time_to_reward(n_envs, target=100.0, max_iters=40, seeds=(0, 1, 2)):
    """BASELINE metric part (ii): PPO wall-clock until the iteration's mean episode return (avg_ep_rews, ppo.py:833) reaches
    +100, from a fresh policy on the configs[1] workload; the clock includes every launch from the first reset.  One run per
    seed (policy init, goal streams and action noise all follow it); `seconds` is the median, every run is listed."""
    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv
    runs = []
    for seed in seeds:
        env = VecEnv(n_envs, map="stage_1", max_episode_steps=500, seed=seed)
        tr = ppo.PPOTrainer(env, ppo.PPOConfig(policy="mlp64x2", seed=seed))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trace = []
        for it in range(max_iters):
            lg = tr.iteration()
            torch.cuda.synchronize()
            trace.append([round(time.perf_counter() - t0, 3), round(lg["avg_ep_rews"], 2), round(lg["success_rate"], 4)])
            if lg["avg_ep_rews"] >= target and it >= 1:
                break
        env.close()
        runs.append(dict(seed=seed, seconds=trace[-1][0], reached=bool(trace[-1][1] >= target), iterations=len(trace),
                         env_steps=tr.env_steps, trace_sec_meanreward_success=trace))
        del tr
    secs = sorted(r["seconds"] for r in runs)
    return dict(seconds=secs[len(secs) // 2], seconds_min=secs[0], seconds_max=secs[-1], reached=all(r["reached"] for r in runs),
                target=target, seeds=list(seeds), runs=runs)


MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF headline is 2:1 sparse)
MFMA_F32_PEAK_TF = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA, 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz


def resmlp512_update_roofline(tr, reps=5, tag=None):
    """The update kernels of the 512-wide nets alone: HIP events around whole epochs on the trainer's own buffers (the epoch is 8
    launches on one stream).  `achieved` / `frac` count the ALGORITHMIC float32 FLOP -- 131,072 MACs per sample and net (W1a 16x512,
    W2a 512x16, W1b 32x512, W2b 512x32, forward + both backward products) -- against the f32-input MFMA peak; `executed` adds the
    hidden layers the backward kernels recompute instead of storing (155,648 MACs)."""
    up = tr.updater
    if not up.fused_resmlp512:
        return None
    T, N, D = tr.cfg.rollout_len, tr.env.N, tr.env.D
    obs, acts = tr.obs_buf[:T].reshape(T * N, D), tr.act_buf.reshape(T * N, 2)
    logp, rtg = tr.logp_buf.reshape(T * N), tr.rtg_buf.reshape(T * N)
    adv = torch.randn(T * N, device=obs.device)
    st = torch.zeros(8, device=obs.device)
    up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
    torch.cuda.synchronize()
    if PLAIN:
        ms = _plain(tag, ["resmlp_"], lambda: up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st), reps)
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
    flop = 2 * 2 * 131072 * T * N
    flop_exec = 2 * 2 * 155648 * T * N
    return dict(bound="mfma", kernel="navppo_resmlp512_update_epoch (resmlp_fwd<16|32>, resmlp_bwd2s, resmlp_bwd<16>, 3 streaming kernels, reduce+Adam)",
                achieved=round(flop / ms / 1e9, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(flop / ms / 1e9 / MFMA_F32_PEAK_TF, 4),
                epoch_ms=round(ms, 3), flop_per_epoch=flop, traffic=None,
                executed=dict(achieved=round(flop_exec / ms / 1e9, 2), frac=round(flop_exec / ms / 1e9 / MFMA_F32_PEAK_TF, 4), flop_per_epoch=flop_exec,
                              note="incl. the hidden layers recomputed in the backward kernels (155,648 instead of 131,072 MACs per sample and net)"),
                detail="algorithmic float32 FLOP against the f32-input MFMA peak.  The products that fill a k-step of v_mfma_f32_16x16x32_bf16 "
                       "(H, Y of resmlp_fwd<32>; H^T, dH^T, dW2, dW1, Q of rb2's backward -- resmlp_bwd2s, one hand-placed instruction stream per "
                       "wave, 4 waves x 512 registers; H of resmlp_fwd<16>) run as float32 products from three-piece bf16 splits, the rest on "
                       "v_mfma_f32_16x16x4_f32 (f32 MFMA and VALU share the SIMD's FMA lanes, the loop sustains ~2.2 GHz)")


def resmlp512_leg(n_envs, rollout, epochs, steps=2, prof=None):
    """SURVEY 8(d) cfg 2 "reported alongside": the reference's ACTIVE nets (net_actor.py:56-144, net_critic.py:50-130) on the
    same workload: fused HIP update (csrc/ppo_resmlp512.hip), rollout in one persistent launch (navsim_rollout_resmlp512: the policy
    step of csrc/resmlp_policy.h in front of every env step; round 4 ran a hipGraph of policy launch + step launch per step)."""
    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv
    env = VecEnv(n_envs, map="stage_1", max_episode_steps=500, seed=0)
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=rollout, n_updates_per_iteration=epochs, policy="resmlp512", seed=0))
    tr.iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = u = 0.0
    for _ in range(steps):
        lg = tr.iteration()
        r += lg["rollout_time"]
        u += lg["update_time"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    up = tr.updater
    roof = apply_rocprof(resmlp512_update_roofline(tr), prof, "resmlp512.update_roofline")
    env.close()
    return dict(policy="resmlp512", value=round(steps * rollout * n_envs / dt, 1), unit="env-steps/s", steps=steps,
                ms_per_step=round(dt / steps * 1e3, 2), rollout_ms=round(r / steps * 1e3, 2), update_ms=round(u / steps * 1e3, 2),
                rollout=("persistent kernel (navsim_rollout_resmlp512: rollout_resmlp_kernel, 16 envs on 8 waves)"
                         if tr.uses_persistent_rollout else "hipGraph of per-step launches"),
                update="fused MFMA kernels (products that fill a bf16 k-step as bf16x3 float32 products, the rest on the f32-input MFMA)" if up.fused_resmlp512 else "PyTorch-ROCm", update_roofline=roof)


def ppo_shard_leg(n_envs, world, n_beams, obs_f16, sampler, rollout, epochs, detail, steps=2, policy="mlp64x2"):
    """One GPU's shard of a BASELINE 8-GPU configuration as a PPO workload, end to end: the same timed
    region as the driver line -- persistent HIP rollout on the shard's own observation format (42-D rows with 36 beams, float16
    rows), return scan, V0, advantage normalisation and all epochs of the fused D-64-64 update -- env-steps/s of ONE GPU."""
    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv
    env = VecEnv(n_envs, map=world, n_beams=n_beams, max_episode_steps=500, seed=0, obs_f16=obs_f16, sampler=sampler)
    tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=rollout, n_updates_per_iteration=epochs, policy=policy, seed=0))
    tr.iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = u = 0.0
    for _ in range(steps):
        lg = tr.iteration()
        r += lg["rollout_time"]
        u += lg["update_time"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    inf = env.sim.info()
    D = env.D
    if policy == "resmlp512":   # the reference's ACTIVE nets on the shard's own row type (round 6: float16 rows through the 512-wide kernels)
        out = dict(workload=f"{n_envs} envs, {world} ({env.sim.S} segments, shared map), {n_beams} beams, {D}-D "
                            f"{'float16' if obs_f16 else 'float32'} rows, PPO resmlp512 (NetActor / NetCritic), rollout={rollout}, {epochs} full-batch epochs",
                   value=round(steps * rollout * n_envs / dt, 1), unit="env-steps/s per GPU", steps=steps,
                   ms_per_step=round(dt / steps * 1e3, 3), rollout_ms=round(r / steps * 1e3, 3), update_ms=round(u / steps * 1e3, 3),
                   rollout_us_per_step=round(r / steps / rollout * 1e6, 2),
                   rollout="persistent kernel (navsim_rollout_resmlp512)" if tr.uses_persistent_rollout else "hipGraph of per-step launches",
                   update="navppo_resmlp512_update_epoch" if tr.updater.fused_resmlp512 else "PyTorch-ROCm",
                   update_roofline=resmlp512_update_roofline(tr, reps=3),
                   last_iter={k: lg[k] for k in ("avg_ep_rews", "success_rate", "episodes")}, bound_detail=detail)
        env.close()
        return out
    roof = mlp64_update_roofline(tr, reps=10)
    roof_f32 = mlp64_update_roofline(tr, reps=10, arith="f32") if roof and roof["arith"] != "f32" else None   # same buffers, same weights
    out = dict(workload=f"{n_envs} envs, {world} ({env.sim.S} segments, shared map), {n_beams} beams, {D}-D "
                        f"{'float16' if obs_f16 else 'float32'} rows, PPO mlp64x2 ({D}-64-64), rollout={rollout}, {epochs} full-batch epochs",
               value=round(steps * rollout * n_envs / dt, 1), unit="env-steps/s per GPU", steps=steps,
               ms_per_step=round(dt / steps * 1e3, 3), rollout_ms=round(r / steps * 1e3, 3), update_ms=round(u / steps * 1e3, 3),
               rollout_us_per_step=round(r / steps / rollout * 1e6, 2),
               rollout=("persistent kernel (navsim_rollout_mlp64: " + (f"rollout_big_kernel, {inf['rollout_epb']} envs on {inf['rollout_waves']} waves" if inf["rollout_kind"] == 2
                        else f"rollout_kernel, {inf['rollout_epb']} envs on 8 waves") + (", tile boxes" if inf["rollout_cast"] == 3 else "") + ")")
               if tr.updater.fused_mlp64 else "hipGraph of policy + navsim_step launches",
               update=("navppo_mlp64_bf16x3_update_epoch" if tr.updater.bf16x3 else "navppo_mlp64_update_epoch") if tr.updater.fused_mlp64
               else "PyTorch-ROCm", update_roofline=roof, update_roofline_f32=roof_f32,
               last_iter={k: lg[k] for k in ("avg_ep_rews", "success_rate", "episodes")}, bound_detail=detail)
    env.close()
    return out


def mlp64_update_roofline(tr, reps=40, arith=None, tag=None):
    """The update kernels of the TIMED workload alone (94 % of the timed region): HIP events around whole epochs of
    navppo_mlp64[_bf16x3]_update_epoch (pass kernel + reduce_adam) on the trainer's own rollout buffers.  arith: None = the arithmetic
    the trainer runs (PPOConfig.update_arith), "f32" / "bf16x3" = that one (same buffers, same weights: the two legs are like for like)."""
    up = tr.updater
    if not up.fused_mlp64:
        return None
    T, N, D = tr.cfg.rollout_len, tr.env.N, tr.env.D
    obs, acts = tr.obs_buf[:T].reshape(T * N, D), tr.act_buf.reshape(T * N, 2)
    logp, rtg = tr.logp_buf.reshape(T * N), tr.rtg_buf.reshape(T * N)
    adv = torch.randn(T * N, device=obs.device)
    st = torch.zeros(8, device=obs.device)
    was = up.bf16x3
    if arith is not None:
        up.bf16x3 = arith == "bf16x3"
    split_ms = None
    try:
        if up.bf16x3:   # the one-off split of the observations (once per update, not per epoch)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            up.prepare(obs)
            e0.record()
            up.prepare(obs)
            e1.record()
            torch.cuda.synchronize()
            split_ms = e0.elapsed_time(e1)
        for _ in range(10):   # the clock settles on the MFMA loop's level within a few epochs
            up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
        torch.cuda.synchronize()
        if PLAIN:
            ms = _plain(tag, ["::mlp64_pass_both", "::reduce_adam<"], lambda: up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st), min(reps, 20))
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                up._fused_epoch(obs, acts, logp, rtg, adv, 0.8, st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
        x3 = up.bf16x3
    finally:
        up.bf16x3 = was
    # MACs per sample and net: F1 64 D + F2 4096 + B2 4096 + G2 4096 + G1 64 D on MFMA (D = 16: 14,336), output units / their
    # gradients on the vector units (actor 384, critic 192): D = 16: 2 x (2 x 14,336 + 576) = 58,496 FLOP per sample
    flop = (2 * 2 * (12288 + 128 * D) + 2 * 576) * T * N
    f32_equiv = dict(achieved=round(flop / ms / 1e9, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", frac=round(flop / ms / 1e9 / MFMA_F32_PEAK_TF, 4),
                     flop_per_epoch=flop, note="ALGORITHMIC float32 FLOP of the epoch against the f32-input MFMA peak")
    if not x3:
        out = dict(bound="mfma", arith="f32", kernel="navppo_mlp64_update_epoch (mlp64_pass_both + reduce_adam)", **{k: f32_equiv[k] for k in
                   ("achieved", "peak", "unit", "frac")}, epoch_us=round(ms * 1e3, 1), flop_per_epoch=flop, samples=T * N, traffic=None)
    else:
        # The split pass runs on the bf16 matrix pipe: `achieved` / `peak` / `frac` are what THAT unit executes -- six bf16 piece products
        # per float32 product (v_mfma_f32_32x32x16_bf16 / 16x16x32; 16 columns: + the ones-products of db1) -- against the dense bf16 MFMA
        # peak.  The algorithmic float32 FLOP against the f32-input MFMA peak (the number earlier rounds printed as `frac`: a value
        # above 1 there means "beyond what the f32 MFMA can do", not a utilisation) is kept beside it as `f32_equivalent`.
        sched = D == 16   # the hand-placed stream of round 6 (csrc/ppo_mlp64_x3s.h); 42-column rows run round 5's pass
        mfma_flop = 6 * 2 * 2 * (12288 + 128 * (16 if D == 16 else 48)) * T * N   # (42-column rows are 48 columns on chip)
        if sched:
            mfma_flop += 2 * 12 * 16384 * ((T * N + 31) // 32)   # db1 = dH1^T 1: 12 v_mfma_f32_16x16x32_bf16 per tile and net
        out = dict(bound="mfma", arith="bf16x3",
                   kernel=("navppo_mlp64_bf16x3_update_epoch (mlp64_pass_both_x3s + reduce_adam)" if sched else
                           "navppo_mlp64_bf16x3_update_epoch (mlp64_pass_both_x3 + reduce_adam)"),
                   achieved=round(mfma_flop / ms / 1e9, 1), peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s (bf16 MFMA, executed)",
                   frac=round(mfma_flop / ms / 1e9 / MFMA_BF16_PEAK_TF, 4), epoch_us=round(ms * 1e3, 1), flop_per_epoch=mfma_flop, samples=T * N,
                   traffic=None, f32_equivalent=f32_equiv, split_obs_us_per_update=round(split_ms * 1e3, 1))
        if sched:
            out["detail"] = ("float32 products from operands split into three bf16 pieces, six piece products each on the bf16 MFMA, float32 "
                             "accumulate (float32-equivalent: tests/test_gpu_bf16x3.py).  Round 6 (csrc/ppo_mlp64_x3s.h): one hand-placed stream per "
                             "wave, 4 waves x 512 registers, every operand split once and transposed as bf16 pieces through LDS "
                             "(ds_read_b64_tr_b16); per 32-sample tile and net 156 + 36 MFMAs (5,568 cycles: SQ_VALU_MFMA_BUSY = 48 % of the "
                             "launch) beside ~1,200 vector and ~270 LDS instructions; the clock settles at ~2.0 GHz under this stream "
                             "(2.2 under round 5's, 2.4 under the f32 pass) -- profiles/r06_x3s_pmc.txt")
        else:
            out["detail"] = ("float32 products from operands split into three bf16 pieces, six piece products each on the bf16 MFMA, float32 "
                             "accumulate (float32-equivalent: tests/test_gpu_bf16x3.py).  Round 5's compiler-scheduled pass (8 waves at 16 columns; "
                             "42-column rows -- 48 on chip, 252 MFMAs per tile and net -- 4 waves x 512 registers): 180 MFMAs beside ~1,465 vector "
                             "instructions at 16 columns, SQ_VALU_MFMA_BUSY = 33 % of the launch -- profiles/r05_bf16x3_pmc.txt")
    if not x3:
        out["detail"] = ("f32-input MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32): 14,336 MFMA cycles + ~575 vector instructions per 32-sample "
                         "tile and net at D = 16; f32 MFMA and VALU share the SIMD's FMA lanes, the loop sustains ~2.2 GHz")
    return out


def env_n1_step_us(budget_s=1.5):
    """The N = 1 drop-in class `Env` (navbot_ppo_amd/env.py) stepped from Python exactly as PPO.rollout steps the reference's
    (ppo.py:541, 591-593): one launch + one stream wait per step, caller-side reset."""
    from navbot_ppo_amd.env import Env
    env = Env(is_training=True)
    env.reset()
    rng = np.random.default_rng(0)
    past = [0.0, 0.0]
    n, t0 = 0, None
    while True:
        a = [rng.uniform(0, 1), rng.uniform(-1, 1)]
        _, _, d, ar = env.step(a, past)
        past = a
        if d or ar:
            env.reset()
            past = [0.0, 0.0]
        n += 1
        if n == 200:
            t0, n0 = time.perf_counter(), n
        if t0 is not None and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    env.close()
    return round(dt / (n - n0) * 1e6, 2)


def dist_diagnostics(ctx, trainer, roll_ms, upd_ms, epochs=20):
    """N > 1: what a scaling run needs to be read in one shot -- every rank takes part, rank 0 reports.
      rollout_ms / update_ms    min and max over the ranks of the timed region's split (a straggler shows as max >> min)
      allreduce_us              the flat-gradient all-reduce alone (the ONE collective per epoch), HIP events around 200 of them
                                back to back on the rank's stream, max over ranks: wire + RCCL launch -- the term DESIGN section 7
                                could only assume (~25 us)
      epoch_us_with_allreduce / epoch_us_local   one multi-GPU epoch (fused passes -> all-reduce -> scale + Adam) against the same
                                epoch without the collective, 20 epochs each on the rank's own batch: their difference is what
                                the collective costs IN the epoch (launch + wire + the cross-stream hand-over), max over ranks"""
    import torch.distributed as dist
    dev = ctx.device
    t = torch.tensor([roll_ms, upd_ms], dtype=torch.float64, device=dev)
    lo, hi = t.clone(), t.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    up = trainer.updater
    g = up.fp.grad.clone()

    def timed(fn, reps):
        torch.cuda.synchronize()
        ctx.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for _ in range(10):
        ctx.all_reduce_sum(g)
    ar_us = timed(lambda: ctx.all_reduce_sum(g), 200)
    ep_with = ep_local = None
    if up.fused:
        T, N, D = trainer.cfg.rollout_len, trainer.env.N, trainer.env.D
        obs, acts = trainer.obs_buf[:T].reshape(T * N, D), trainer.act_buf.reshape(T * N, 2)
        logp, rtg = trainer.logp_buf.reshape(T * N), trainer.rtg_buf.reshape(T * N)
        adv = torch.randn(T * N, device=dev)
        flat, m_, v_, t_ = up.fp.flat.clone(), up._adam_m.clone(), up._adam_v.clone(), up._adam_t
        st = torch.zeros(8, device=dev)

        def epoch(collective):
            up._fused_loss_grad(obs, acts, logp, rtg, adv, 0.8, stats=st)
            if collective:
                ctx.all_reduce_sum(up.fp.grad)
            up._fused_adam(1.0 / ctx.world)

        for c in (True, False):
            epoch(c)
        ep_with = timed(lambda: epoch(True), epochs)
        ep_local = timed(lambda: epoch(False), epochs)
        up.fp.flat.copy_(flat), up._adam_m.copy_(m_), up._adam_v.copy_(v_)   # the diagnostic epochs leave no trace
        up._adam_t = t_
    d = torch.tensor([ar_us, ep_with or 0.0, ep_local or 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(d, op=dist.ReduceOp.MAX)
    ar_us, ep_with_m, ep_local_m = (float(x) for x in d.tolist())
    return dict(ranks=ctx.world, backend=ctx.backend, rollout_ms_min=round(float(lo[0]), 3), rollout_ms_max=round(float(hi[0]), 3),
                update_ms_min=round(float(lo[1]), 3), update_ms_max=round(float(hi[1]), 3), allreduce_us=round(ar_us, 2),
                allreduce_bytes=int(g.numel() * 4), epoch_us_with_allreduce=round(ep_with_m, 2) if up.fused else None,
                epoch_us_local=round(ep_local_m, 2) if up.fused else None,
                allreduce_cost_in_epoch_us=round(ep_with_m - ep_local_m, 2) if up.fused else None)


PMC_FILE = [None]   # --with-pmc-file: counters recorded in the SAME gpurun call as this bench run (tools/prof_all.sh)


def profiled_traffic(key):
    """HBM bytes per launch from rocprofv3 --pmc passes (tools/pmc_traffic.py): the file given with --with-pmc-file (recorded in
    the same call as this run), else the committed profiles/pmc_traffic.json; either is used only while it was recorded from THIS
    kernel source (sha256 of csrc/navsim.hip), else None."""
    import hashlib
    f = PMC_FILE[0] or os.path.join(REPO, "profiles", "pmc_traffic.json")
    src = os.path.join(REPO, "navbot_ppo_amd", "csrc", "navsim.hip")
    if not (os.path.exists(f) and os.path.exists(src)):
        return None
    d = json.load(open(f))
    if d.get("navsim_hip_sha256") != hashlib.sha256(open(src, "rb").read()).hexdigest():
        return None
    return d.get(key)


def traffic_source():
    f = PMC_FILE[0] or os.path.join(REPO, "profiles", "pmc_traffic.json")
    if not os.path.exists(f):
        return None
    d = json.load(open(f))
    return dict(file=os.path.relpath(f, REPO), run_id=d.get("run_id"), recorded_utc=d.get("recorded_utc"),
                same_call_as_this_run=bool(PMC_FILE[0]) and d.get("run_id") == os.environ.get("PROF_RUN_ID"))


def pmc_field(key):
    """VALU-issue figures of a leg from the same pmc file (tools/pmc_traffic.py records SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU /
    SQ_BUSY_CYCLES passes for the shard legs), or None."""
    return profiled_traffic(key)


def shard_valu(leg, key):
    """vector-issue fraction of the shard's step launch from the counters of the pmc file (SQ_ACTIVE_INST_VALU), or None"""
    v = pmc_field(key)
    leg["valu"] = v
    leg["valu_issue_frac_step"] = round(v["valu_busy_us_per_simd_at_2p4GHz"] / leg["step_us"], 4) if v else None


def shard_leg(w, T, detail, tag=None, prof=None):
    """One GPU's shard of a BASELINE 8-GPU configuration: us per step one launch per step and one launch per tape, env-steps/s."""
    a = apply_rocprof(step_kernel_roofline(w, iters=64 * 200, detail=detail, tag=f"{tag}.step"), prof, f"{tag}.step")
    b = apply_rocprof(step_seq_roofline(w, T, reps=6, detail=detail, tag=f"{tag}.tape"), prof, f"{tag}.tape")
    if PLAIN:
        return None
    return dict(time_source=a.get("time_source"), workload=w.describe(), step_us=a["launch_us"], tape_us_per_step=b["us_per_step"], env_steps_per_sec_step=a["env_steps_per_sec"],
                env_steps_per_sec_tape=b["env_steps_per_sec"], bytes_per_env_step=w.bytes_per_env_step,
                hbm_frac_step=a["frac"], hbm_frac_tape=b["frac"], bound_detail=detail,
                ray_segment_tests_per_sec_tape=round(w.n_envs * w.S * w.B / (b["us_per_step"] * 1e-6), 1))


def kernel_legs(trainer, prof):
    """Every roofline leg of the line, in a fixed order -- run with HIP events by the bench itself and, in the child process under
    rocprofv3 (PLAIN), as plain launches behind markers; `prof` (the child's per-leg kernel times) then replaces the event figures."""
    out = {}
    ap = lambda leg, key: apply_rocprof(leg, prof, key)
    # GPU legs first and long enough (several seconds in total) for an outside utilisation sampler to see them
    out["roofline_timed_region"] = ap(rollout_kernel_leg(trainer, tag="roofline_timed_region"), "roofline_timed_region") if trainer.updater.fused_mlp64 else None
    out["update_roofline"] = ap(mlp64_update_roofline(trainer, tag="update_roofline"), "update_roofline")   # the arithmetic the timed region ran
    out["update_roofline_f32"] = ap(mlp64_update_roofline(trainer, arith="f32", tag="update_roofline_f32"), "update_roofline_f32")   # the native f32-MFMA pass, same buffers
    if out["update_roofline"] and out["update_roofline"]["arith"] != "bf16x3":
        out["update_roofline_bf16x3"] = ap(mlp64_update_roofline(trainer, arith="bf16x3", tag="update_roofline_bf16x3"), "update_roofline_bf16x3")
    # The ray-cast run of BASELINE configs[2].  `roofline` is the entry point the north star names -- navsim_step, ONE launch per
    # env step, what a policy outside the kernel drives (Env.step, environment_new.py:272-310; like for like with rounds 1-2) --
    # and beside it the two persistent forms of the same step body: closed loop (the 16-64-64 policy in the kernel: PPO.rollout,
    # ppo.py:505-594) and an open-loop action tape (replay / evaluation form: the actions do not depend on the observations).
    l3 = "working set 35.7 MB per step sits in the 256 MiB Infinity Cache: L3-fed, VALU-issue bound at this size"
    w = CastWorkload(16384, "stage_2", per_env=True)
    out["roofline"] = ap(step_kernel_roofline(
        w, iters=64 * 1500, detail=l3 + "; one step per launch: + launch ramp of 256 x 16 waves, state round trips, graph-node boundary", tag="roofline"), "roofline")
    out["roofline_closed_loop"] = ap(closed_loop_roofline(
        w, T=256, detail=l3 + "; the 16-64-64 policy chooses every action in-kernel (PPO.rollout closed-loop): + the policy phase, "
                           "bound by the SIMDs' f32 MFMA pipes (64 envs x 10.2 kFLOP per CU and step = 1.07 us)", tag="roofline_closed_loop"), "roofline_closed_loop")
    out["roofline_open_loop_tape"] = ap(step_seq_roofline(
        w, T=256, detail=l3 + "; open loop: the 256 actions of a launch are known ahead (replay / evaluation), NOT what PPO.rollout does",
        tag="roofline_open_loop_tape"), "roofline_open_loop_tape")
    del w
    # the same three at S = 1024 per env: 16384 x 1024 x 16 B = exactly 256 MiB = the Infinity Cache size (AT the L3, not beyond it)
    at = "segment stream 268 MB per step = the 256 MiB Infinity Cache size: no L2 reuse, L3 hits possible"
    w = CastWorkload(16384, "stage_2", per_env=True, sides=248)
    out["roofline_at_l3"] = ap(step_kernel_roofline(w, iters=64 * 400, detail=at + "; one step per launch", tag="roofline_at_l3"), "roofline_at_l3")
    out["roofline_at_l3_closed_loop"] = ap(closed_loop_roofline(w, T=64, detail=at + "; closed loop (policy in the kernel)", tag="roofline_at_l3_closed_loop"),
                                           "roofline_at_l3_closed_loop")
    out["roofline_at_l3_open_loop_tape"] = ap(step_seq_roofline(w, T=64, detail=at + "; open-loop tape", tag="roofline_at_l3_open_loop_tape"),
                                              "roofline_at_l3_open_loop_tape")
    del w
    torch.cuda.empty_cache()
    # ... and at S = 2048 per env (the size SURVEY.md section 7 names for the HBM-bound regime): 512 MiB per step = 2x the Infinity
    # Cache, so every segment byte of a step comes from HBM; frac = of the 8 TB/s spec, frac_of_achievable_hbm = of the ~6.3 TB/s
    # the guide gives as achievable
    hb = "segment stream 537 MB per step = 2x the 256 MiB Infinity Cache: HBM-bound"
    w = CastWorkload(16384, "stage_2", per_env=True, sides=504)
    out["roofline_hbm"] = ap(step_kernel_roofline(w, iters=64 * 200, detail=hb + "; one step per launch", tag="roofline_hbm"), "roofline_hbm")
    out["roofline_hbm_closed_loop"] = ap(closed_loop_roofline(w, T=32, reps=6, detail=hb + "; closed loop (policy in the kernel)", tag="roofline_hbm_closed_loop"),
                                         "roofline_hbm_closed_loop")
    out["roofline_hbm_open_loop_tape"] = ap(step_seq_roofline(w, T=32, reps=6, detail=hb + "; open-loop tape", tag="roofline_hbm_open_loop_tape"),
                                            "roofline_hbm_open_loop_tape")
    del w
    torch.cuda.empty_cache()
    if not PLAIN:
        for k, t in (("roofline", "cfg3_step"), ("roofline_closed_loop", "cfg3_closed_loop"), ("roofline_open_loop_tape", "cfg3_seq"),
                     ("roofline_at_l3", "s1024_step"), ("roofline_at_l3_closed_loop", "s1024_closed_loop"), ("roofline_at_l3_open_loop_tape", "s1024_seq"),
                     ("roofline_hbm", "s2048_step"), ("roofline_hbm_closed_loop", "s2048_closed_loop"), ("roofline_hbm_open_loop_tape", "s2048_seq")):
            out[k]["traffic"] = profiled_traffic(t + "_bytes_per_launch")
    # one GPU's shard of the two 8-GPU configurations of BASELINE.json (shared maps: VALU-bound, SURVEY 8d caveat -- the byte
    # fraction is nominal there, the vector-issue fraction from the counters says how busy the SIMDs are)
    out["cfg4_shard"] = shard_leg(CastWorkload(4096, "stage_4", per_env=False, n_beams=36), 128,
                                  "BASELINE configs[3] per GPU: 32768 / 8 envs, stage_4 (64 segments, shared), 36 beams", "cfg4_shard", prof)
    out["cfg5_shard"] = shard_leg(CastWorkload(8192, "house", per_env=False, obs_f16=True, sampler="small_house", house_segments=2048), 64,
                                  "BASELINE configs[4] per GPU: 65536 / 8 envs, 2048-segment house map (shared, tile boxes), f16 "
                                  "observations, start / goal tables", "cfg5_shard", prof)
    if PLAIN:   # the update of the reference's ACTIVE nets (its own leg of the line, resmlp512_leg, takes the figure from here)
        from navbot_ppo_amd import ppo
        from navbot_ppo_amd.env import VecEnv
        env = VecEnv(trainer.env.N, map="stage_1", max_episode_steps=500, seed=0)
        tr = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=trainer.cfg.rollout_len, n_updates_per_iteration=1, policy="resmlp512", seed=0))
        tr.iteration()
        resmlp512_update_roofline(tr, reps=3, tag="resmlp512.update_roofline")
        env.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --envs-per-gpu envs on every GPU; strong: --envs-total envs split over the GPUs")
    ap.add_argument("--envs-total", type=int, default=4096, help="strong scaling: total envs (shards of envs-total / gpus)")
    ap.add_argument("--rollout", type=int, default=512)
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--policy", default="mlp64x2")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline legs")
    ap.add_argument("--with-pmc-file", default=None,
                    help="pmc_traffic.json recorded by tools/pmc_traffic.py in the same gpurun call (roofline.traffic comes from it)")
    ap.add_argument("--update-arith", choices=["bf16x3", "f32"], default="bf16x3",
                    help="arithmetic of the fused 16-64-64 update's matrix products (PPOConfig.update_arith)")
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="N > 1, mlp64x2: the two-stage per-net pipeline instead of one all-reduce of the flat gradient per epoch")
    ap.add_argument("--kernel-trace-legs", action="store_true",
                    help="(run by the bench itself under rocprofv3, NAVBOT_BENCH_PLAIN=1) only the roofline legs, as plain launches behind markers")
    args = ap.parse_args()
    PMC_FILE[0] = args.with_pmc_file
    if args.kernel_trace_legs:
        if not PLAIN:
            raise SystemExit("--kernel-trace-legs is run by bench.py itself (NAVBOT_BENCH_PLAIN=1)")
        from navbot_ppo_amd import ppo
        from navbot_ppo_amd.env import VecEnv
        env = VecEnv(args.envs_per_gpu, map="stage_1", n_beams=10, max_episode_steps=500, auto_reset=True, seed=0)
        trainer = ppo.PPOTrainer(env, ppo.PPOConfig(rollout_len=args.rollout, max_episode_steps=500, n_updates_per_iteration=1, policy=args.policy,
                                                    seed=0, update_arith=args.update_arith))
        trainer.iteration()   # the buffers the update legs run on
        kernel_legs(trainer, None)
        print("MANIFEST " + json.dumps(_MANIFEST), flush=True)
        return

    from navbot_ppo_amd import ppo
    from navbot_ppo_amd.env import VecEnv

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    ctx = ppo.DistCtx()
    if ctx.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ctx.world}: launch with torch.distributed.run")
    if ctx.world > 1 and not os.environ.get("NAVBOT_DIST_BACKEND"):
        # a scaling run must be a run over RCCL with one rank per GPU -- anything else is a different experiment.
        # (NAVBOT_DIST_BACKEND=gloo is the explicit override of the 1-GPU tests, where two ranks share a device.)
        n_rccl = torch.distributed.get_world_size() if ctx.backend == "nccl" else 0
        if n_rccl != args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: RCCL connected {n_rccl} ranks (backend {ctx.backend!r}); refusing to report a "
                             "scaling number that did not run over RCCL")
        if torch.cuda.device_count() < ctx.world:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} devices visible: one rank per GPU")
    if args.scaling == "strong":
        if args.envs_total % ctx.world:
            raise SystemExit(f"--envs-total {args.envs_total} is not divisible by {ctx.world} GPUs")
        n_local = args.envs_total // ctx.world
    else:
        n_local = args.envs_per_gpu
    n_total = n_local * ctx.world
    lo, _ = ctx.shard(n_total)
    env = VecEnv(n_local, map="stage_1", n_beams=10, max_episode_steps=500, auto_reset=True, seed=0, env_id_base=lo,
                 device=ctx.device)
    cfg = ppo.PPOConfig(rollout_len=args.rollout, max_episode_steps=500, n_updates_per_iteration=args.epochs,
                        policy=args.policy, use_graph=not args.no_graph, seed=0,
                        overlap_allreduce=args.overlap_allreduce, update_arith=args.update_arith)
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
    diag = dist_diagnostics(ctx, trainer, roll_t / args.steps * 1e3, upd_t / args.steps * 1e3) if ctx.enabled else None

    out = None
    if ctx.rank == 0:
        K = args.steps
        out = {
            "metric": "env_steps_per_sec", "value": round(K * args.rollout * n_total / dt, 1), "unit": "env-steps/s",
            "n_gpus": ctx.world, "steps": K, "warmup": args.warmup, "ms_per_step": round(dt / K * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32",
            "dtype_detail": "pose/goal angles/reward f64, ray-cast f32, PPO nets f32" + (
                "; the update's matrix products are float32 products evaluated from three-piece bf16 splits on the bf16 MFMA (six piece "
                "products, float32 accumulate: 'bf16x3', float32-equivalent against float64 -- tests/test_gpu_bf16x3.py; "
                "--update-arith f32 selects the f32-input MFMA)" if trainer.updater.bf16x3 else "; update on the f32-input MFMA"),
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {n_local} envs/GPU, stage_1 (32 segments), 10 beams, "
                                   f"PPO {args.policy}, rollout={args.rollout}, {args.epochs} full-batch epochs, episode cap 500",
                       "n_envs_total": n_total, "n_envs_per_gpu": n_local, "rollout_len": args.rollout,
                       "epochs": args.epochs, "policy": args.policy, "parallelism": f"env-shard dp{ctx.world}",
                       "update_arith": "bf16x3" if trainer.updater.bf16x3 else "f32",
                       "rollout": "persistent kernel (navsim_rollout_mlp64)" if trainer.updater.fused_mlp64 and cfg.persistent_rollout
                       else ("hipGraph of per-step launches" if not args.no_graph else "per-step launches")},
            "dist_backend": ctx.backend, "rccl_version": ctx.rccl_version,
            "rccl_ranks": (torch.distributed.get_world_size() if ctx.backend == "nccl" else 0),
            "rollout_only_env_steps_per_sec": round(K * args.rollout * n_total / roll_t, 1),
            "rollout_ms": round(roll_t / K * 1e3, 3), "update_ms": round(upd_t / K * 1e3, 3), "dist": diag,
            "last_iter": {k: lg[k] for k in ("avg_ep_rews", "success_rate", "episodes", "actor_loss", "critic_loss", "approx_kl")},
        }
    if not args.no_extras and ctx.rank == 0:   # rank 0 only (the other ranks wait at the barrier below)
        # GPU legs first and long enough (several seconds in total) for an outside utilisation sampler to see them
        prof, prof_src = rocprof_legs(["--envs-per-gpu", str(n_local), "--rollout", str(args.rollout), "--epochs", str(args.epochs), "--policy", args.policy,
                                       "--update-arith", args.update_arith])
        out["roofline_time_source"] = prof_src
        out.update(kernel_legs(trainer, prof))
        del trainer
        torch.cuda.empty_cache()
        out["traffic_source"] = traffic_source()
        shard_valu(out["cfg4_shard"], "cfg4_valu")
        shard_valu(out["cfg5_shard"], "cfg5_valu")
        # ... and the same shards as PPO workloads on their own observation formats, end to end per GPU
        out["cfg4_ppo_shard"] = ppo_shard_leg(4096, "stage_4", 36, False, None, args.rollout, args.epochs,
                                              "BASELINE configs[3] per GPU: 32768 / 8 envs, stage_4, 36 beams -> 42-D rows")
        out["cfg5_ppo_shard"] = ppo_shard_leg(8192, "house", 10, True, "small_house", args.rollout, args.epochs,
                                              "BASELINE configs[4] per GPU: 65536 / 8 envs, 2048-segment house map, start / goal tables, "
                                              "float16 observation buffers")
        # ... and configs[4]'s shard with the policy the reference actually runs (round 6: the 512-wide kernels read float16 rows)
        out["cfg5_ppo_shard_resmlp512"] = ppo_shard_leg(8192, "house", 10, True, "small_house", args.rollout, args.epochs,
                                                        "BASELINE configs[4] per GPU with the reference's ACTIVE nets (net_actor.py:56-144): float16 "
                                                        "observation buffers read by navsim_rollout_resmlp512 / navppo_resmlp512_*", steps=1,
                                                        policy="resmlp512")
        if ctx.world == 1:
            ttr = time_to_reward(n_local)
            out["time_to_reward_s"] = ttr["seconds"]
            out["time_to_reward"] = ttr
            out["resmlp512"] = resmlp512_leg(n_local, args.rollout, args.epochs, prof=prof)
            cores = os.cpu_count() or 1
            out["env_n1_step_us"] = env_n1_step_us()
            out["cpu_baseline"] = cpu_baseline(n_local, 1, 8.0)
            # the reference's own Python arithmetic cannot travel to this box: its survey-time figure, for scale
            out["cpu_baseline"]["reference_python_us_per_step"] = 29.3
            out["cpu_baseline"]["reference_python_note"] = ("reference Env.getOdometry + Env.step arithmetic with stubbed ROS and a "
                                                            "canned scan, 1 core of the survey container (Xeon 2.1 GHz), no simulator; "
                                                            "container-only number, not measured on this box")
            out["cpu_baseline_all_cores"] = cpu_baseline(n_local, cores, 6.0)
            out["cpu_baseline_n1"] = cpu_baseline(1, 1, 3.0)
            out["cpu_baseline_n1"]["gpu_env_n1_step_us"] = out["env_n1_step_us"]
    if ctx.rank == 0:
        print(json.dumps(out), flush=True)
    ctx.barrier()
    if ctx.enabled:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
