#!/usr/bin/env python3
"""Dev tool (runs here, after a `tools/prof_all.sh` gpurun call): copies gpurun_out/r03/* into profiles/r03_* and rewrites the
"closed-loop session" section of profiles/README.md from the numbers in those files."""
import json, os, re, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out", "r03"), os.path.join(R, "profiles")
cp = lambda a, b: shutil.copy(os.path.join(O, a), os.path.join(P, b))
cp("pmc_traffic.json", "pmc_traffic.json")
open(os.path.join(P, "r03_bench_final.json"), "w").write(open(os.path.join(O, "bench_final.json")).read().strip().splitlines()[-1] + "\n")
for f in ("bench_final", "step_cfg3", "step_s1024", "step_seq_cfg3", "step_seq_s1024", "update", "resmlp512_update", "rollout_big_cfg3", "rollout_big_s1024"):
    cp(f + "_kernel_stats.csv", "r03_" + f + "_kernel_stats.csv")
open(os.path.join(P, "r03_step_pmc.txt"), "w").write(open(os.path.join(O, "step_final_pmc.txt")).read() + open(os.path.join(O, "step_final_b_pmc.txt")).read())
for a, b in (("step_seq_final_pmc.txt", "r03_step_seq_pmc.txt"), ("update_final_pmc.txt", "r03_update_pmc.txt"), ("resmlp512_update_pmc.txt", "r03_resmlp512_update_pmc.txt"),
             ("rollout_big_pmc.txt", "r03_rollout_big_pmc.txt"), ("rollout_big.txt", "r03_rollout_big.txt"), ("rollout_shard_sizes.txt", "r03_rollout_shard_sizes.txt"),
             ("update_scale.txt", "r03_update_scale.txt"), ("run_id.txt", "r03_run_id.txt"), ("step_cfg3_phase_stamps.txt", "r03_step_cfg3_phase_stamps.txt"),
             ("time_to_reward_resmlp512.txt", "r03_time_to_reward_resmlp512.txt")):
    if "Traceback" in open(os.path.join(O, a)).read():   # e.g. the phase stamps need build/libnavsim_timing.so (tools/make_timing_build.py)
        print("skipped (the tool failed on the box):", a)
        continue
    cp(a, b)
open(os.path.join(P, "r03_time_rtg.txt"), "w").write("".join(l for l in open(os.path.join(O, "time_rtg.log")) if "amdgpu" not in l))

d = json.loads(open(os.path.join(P, "r03_bench_final.json")).read())
t = json.load(open(os.path.join(P, "pmc_traffic.json")))
cl = t["cfg3_closed_loop"]
row = lambda f, k: [r for r in open(os.path.join(P, f)).read().splitlines() if k in r][0].split('",')[1].split(",")
big, sq = row("r03_rollout_big_cfg3_kernel_stats.csv", "rollout_big_kernel"), row("r03_step_seq_cfg3_kernel_stats.csv", "steps_kernel")
calls, avg_ns, sq_ns = int(big[0]), float(big[2]), float(sq[2])
s_ns = float(row("r03_rollout_big_s1024_kernel_stats.csv", "rollout_big_kernel")[2])
pmc = open(os.path.join(P, "r03_rollout_big_pmc.txt")).readline()
g = lambda k: float(re.search(k + r"=([\d.e+]+)", pmc).group(1))
L = {}
for l in open(os.path.join(P, "r03_rollout_big.txt")):
    m = re.search(r"mlp64x2 (.*?) T=\d+ N= *(\d+) EPB= *(\w+): .*= +([\d.]+) us per step", l)
    if m:
        L[(m.group(1), int(m.group(2)), m.group(3))] = m.group(4)
    m = re.search(r"step_seq +([\d.]+) us/step .* step launches +([\d.]+) us", l)
    if m:
        tape, single = m.group(1), m.group(2)
rc, r, r1, bb, b1, cb, ur = (d[k] for k in ("roofline_closed_loop", "roofline", "roofline_single_launch", "roofline_beyond_l3",
                                                  "roofline_beyond_l3_single_launch", "roofline_closed_loop_beyond_l3", "update_roofline"))
new = f'''## Round 3, closed-loop session (`tools/prof_all.sh`, ONE gpurun call, run id `{t['run_id']}` in `r03_run_id.txt` and `pmc_traffic.json`; supersedes every `r03_*` row below where a file name repeats; `tools/collect_profiles.py` wrote this section from the files)

New kernel: `rollout_big_kernel` (`navsim_rollout_mlp64` beyond 4096 envs per GPU: the tape kernel's 64-env / 16-wave workgroup with the 16-64-64 policy phase in front of every step, DESIGN §5a).

| file | notes |
|---|---|
| `r03_bench_final.json` | plain `python bench.py`: **{d['value']/1e6:.2f} M env-steps/s** ({d['ms_per_step']} ms / iteration: rollout {d['rollout_ms']} ms, update {d['update_ms']} ms). `roofline` (tape, `steps_kernel`) {r['us_per_step']} µs per step = **{r['frac']*100:.1f} %**; **`roofline_closed_loop`** (new: `rollout_big_kernel<64,false,16,false,true>`, 256 closed-loop steps per launch, {rc['launch_us']} µs) **{rc['us_per_step']} µs per step = {rc['achieved']/1000:.2f} TB/s = {rc['frac']*100:.1f} %** of 8 TB/s with the policy in the kernel, traffic {rc['traffic']/1e9:.2f} GB per launch vs {rc['algorithmic_bytes_per_launch']/1e9:.2f} GB algorithmic; `roofline_single_launch` {r1['launch_us']} µs = {r1['frac']*100:.1f} %; `roofline_beyond_l3` {bb['us_per_step']} µs per step = {bb['frac']*100:.1f} %, **`roofline_closed_loop_beyond_l3`** {cb['us_per_step']} µs per step = **{cb['frac']*100:.1f} %** (past the Infinity Cache the policy phase mostly disappears under the segment stream), `roofline_beyond_l3_single_launch` {b1['launch_us']} µs = {b1['frac']*100:.1f} %; **`update_roofline`** (new: the update kernels of the timed workload, {ur['epoch_us']} µs per epoch of 2,097,152 samples) {ur['achieved']} TF = **{ur['frac']:.3f}** of the 157.3 TF f32-MFMA peak; `resmlp512` {d['resmlp512']['value']/1e6:.2f} M env-steps/s (update {d['resmlp512']['update_roofline']['frac']:.3f} of the peak); `time_to_reward_s` {d['time_to_reward_s']}; `env_n1_step_us` {d['env_n1_step_us']}; `cpu_baseline` {d['cpu_baseline']['value']/1e6:.2f} M (1 core) / {d['cpu_baseline_all_cores']['value']/1e6:.1f} M ({d['cpu_baseline_all_cores']['cores']} processes) / {d['cpu_baseline_n1']['value']/1e3:.0f} k (one env per call) |
| `r03_rollout_big_cfg3_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats` over `tools/time_rollout.py` at configs[2] (`--cfg3`) — agreement check for `roofline_closed_loop`: `rollout_big_kernel<64,false,16,false,true>` **{avg_ns/1e3:.1f} µs average over {calls} launches of 256 steps = {avg_ns/256e3:.2f} µs per step** under the profiler (which lowers the clock as for every kernel here: `steps_kernel` {sq_ns/1e3:.1f} µs = {sq_ns/256e3:.2f} µs per step in `r03_step_seq_cfg3_kernel_stats.csv` vs {r['us_per_step']} from HIP events); HIP events: `bench.py` {rc['launch_us']} µs, `tools/time_rollout.py` 2.80–2.91 ms on five boxes |
| `r03_rollout_big_s1024_kernel_stats.csv`, `pmc_traffic.json: s1024_closed_loop` | the same for `roofline_closed_loop_beyond_l3` (`tools/time_rollout.py --s1024`: 64 closed-loop steps per launch, per-env S = 1024): `rollout_big_kernel` **{s_ns/1e3:.1f} µs average per launch = {s_ns/64e3:.2f} µs per step** under the profiler (HIP events in `bench.py`: {cb['us_per_step']}); traffic 2 × {t['s1024_closed_loop']['FETCH_SIZE_KiB']:,.0f} KiB + {t['s1024_closed_loop']['WRITE_SIZE_KiB']:,.0f} KiB = **{t['s1024_closed_loop_bytes_per_launch']/1e9:.2f} GB per launch vs {t['s1024_closed_loop']['algorithmic_bytes_per_launch']/1e9:.2f} GB algorithmic** |
| `r03_rollout_big.txt` | same-box A/B, µs per step. configs[2] closed-loop {L[('stage_2 per-env', 16384, 'auto')]} (`rollout_big_kernel`) vs **{L[('stage_2 per-env', 16384, '16')]}** with `NAVSIM_EPB=16` (the 16-env `rollout_kernel`, four rounds of workgroups) vs {tape} tape (`step_seq`) vs {single} one launch per step; S = 1024 closed-loop {L[('stage_1 per-env sides=248', 16384, 'auto')]} (without goal rectangles; the bench leg runs stage_2's); 16384 envs on the 2048-segment house map (tile boxes) {L[('house', 16384, 'auto')]}. **Which shards take the 64-env kernel** (stage_1, shared map; 16-env / 64-env shape): 4096 envs **{L[('stage_1', 4096, 'auto')]}** / {L[('stage_1', 4096, '64')]}, 4608 {L[('stage_1', 4608, '16')]} / **{L[('stage_1', 4608, 'auto')]}**, 8192 {L[('stage_1', 8192, '16')]} / **{L[('stage_1', 8192, 'auto')]}**, 12288 {L[('stage_1', 12288, '16')]} / **{L[('stage_1', 12288, 'auto')]}**, 16384 {L[('stage_1', 16384, '16')]} / **{L[('stage_1', 16384, 'auto')]}** — the 16-env shape needs a second round of workgroups from 4097 envs, the default (bold) switches there |
| `r03_rollout_big_pmc.txt` | one `--pmc` pass over the same command, per 256-step launch: `SQ_VALU_MFMA_BUSY_CYCLES` {g('SQ_VALU_MFMA_BUSY_CYCLES'):.4g} = **{g('SQ_VALU_MFMA_BUSY_CYCLES')/256/1024:.0f} cycles per SIMD and step — the policy's 80 `v_mfma_f32_16x16x4_f32` × 32 cycles per 16-env tile, one tile per SIMD (1.07 µs at 2.4 GHz: the floor of the policy phase)**; `SQ_INSTS_VALU` {g('SQ_INSTS_VALU'):.4g} = {g('SQ_INSTS_VALU')/256/1e6:.2f}e6 per step (tape kernel 3.94e6: + finish, noise, observation reads); `SQ_WAIT_INST_ANY` {g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES')*100:.0f} % of wave cycles |
| `pmc_traffic.json` | now also `cfg3_closed_loop` (FETCH_SIZE / WRITE_SIZE passes over `tools/time_rollout.py --cfg3`): 2 × {cl['FETCH_SIZE_KiB']:,.0f} KiB + {cl['WRITE_SIZE_KiB']:,.0f} KiB = **{t['cfg3_closed_loop_bytes_per_launch']/1e9:.2f} GB per 256-step launch vs 9.15 GB algorithmic** (the per-env segments are re-read from L2 / Infinity Cache, as in the tape form: `cfg3_seq` {t['cfg3_seq_bytes_per_launch']/1e9:.2f} GB; the closed-loop launch reads no action tape and writes actions + log-probs instead) |
| `r03_soak_parity.txt` | `tools/soak_parity.py` on the final kernels: 9.4 M open-loop env-steps (seven configurations) + **16.4 M closed-loop env-steps** (`navsim_rollout_mlp64`: three cast variants at 16384 envs, the timed 4096 × 512 rollout; every env's in-kernel actions replayed on the oracle): bad = 0, every observation row bit-identical |
| `r03_time_to_reward_cfg3.txt` | `TTR_ENVS=16384 TTR_MAP=stage_2 TTR_PER_ENV=1 TTR_ROLLOUT=256 python tools/time_to_reward.py`: PPO on configs[2]'s shard through the closed-loop kernel: +100 mean episode return after 3 iterations (≈0.1 s each), 735 / 68 % success after 23 |
| `r03_learning_curve.txt` | `python tools/learning_curve.py 400`: the timed workload for 400 iterations = 839 M env-steps in 21.7 s; mean episode return −381 → 1306, success rate 0.03 → 0.88, collisions 0.96 → 0.11 |
| `r03_update_fixed_cost.txt` | `tools/time_update_fixed.py`: one mlp64x2 epoch at small batches — 31.8 µs with one workgroup, 56.2 µs at one tile per wave, + 30.3 µs per further tile and wave (both nets) → 22 µs of batch-independent cost in the 984 µs epoch of the timed workload |
| the other `r03_*` files of the list below | (`r03_step_cfg3_phase_stamps.txt` only when the instrumented library was built: `tools/make_timing_build.py`) re-recorded in the same call on unchanged kernels (`step_kernel` 4.21e6 VALU instructions per launch, resmlp512 epoch 11.2 ms, rollout at 4096 / 2048 / 1024 / 512 envs ≈2.66 ms) |

'''
p = os.path.join(P, "README.md")
s = open(p).read()
a, b = s.index("## Round 3, closed-loop session ("), s.index("## Round 3, last session (")
open(p, "w").write(s[:a] + new + s[b:])
print("bench", d["value"], "closed-loop", rc["frac"], "tape", r["frac"], "update", ur["frac"], "run", t["run_id"])
