#!/usr/bin/env python3
"""Dev tool (runs here, after a `tools/prof_all.sh` gpurun call): copies gpurun_out/<round>/* into profiles/<round>_* and prints the
table of rocprofv3 kernel averages against the bench line's HIP-event figures (the numbers profiles/README.md quotes)."""
import csv, glob, json, os, re, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("ROUND", "r04")
O, P = os.path.join(R, "gpurun_out", RND), os.path.join(R, "profiles")
for f in sorted(glob.glob(os.path.join(O, "*_kernel_stats.csv")) + glob.glob(os.path.join(O, "*_pmc.txt"))):
    shutil.copy(f, os.path.join(P, f"{RND}_" + os.path.basename(f)))
for a in ("run_id.txt", "hbm_legs_hip_events.txt", "rollout_shard_sizes.txt", "update_scale.txt", "time_rtg.txt",
          # round 5: the update passes of both arithmetics / of the 42-column rows, the split pass's error table and microbenchmarks, the PPO shards
          "update_arith_hip_events.txt", "update_wide_hip_events.txt", "bf16x3_error.txt", "bf16_mfma_valu_overlap.txt",
          "bf16_mfma_fillers.txt", "bf16_split_ops.txt", "ppo_cfg4.json", "ppo_cfg5.json", "selected_instantiations.txt",
          "house_rollout_shapes.txt", "soak_parity.txt", "soak_parity_seed1.txt", "soak_parity_seed2.txt", "rollout_resmlp_times.txt", "rollout_resmlp_phases.txt"):
    if os.path.exists(os.path.join(O, a)):
        shutil.copy(os.path.join(O, a), os.path.join(P, f"{RND}_{a}"))
shutil.copy(os.path.join(O, "pmc_traffic.json"), os.path.join(P, "pmc_traffic.json"))
line = [l for l in open(os.path.join(O, "bench_final.json")).read().splitlines() if l.startswith("{")][-1]
open(os.path.join(P, f"{RND}_bench_final.json"), "w").write(line + "\n")
d = json.loads(line)

def avg_us(name, kernel):
    for r in csv.DictReader(open(os.path.join(P, f"{RND}_{name}_kernel_stats.csv"))):
        if kernel in r["Name"]:
            return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    return None, 0

rows = [("roofline", "step_cfg3", "step_kernel", 1), ("roofline_closed_loop", "rollout_big_cfg3", "rollout_big_kernel", 256),
        ("roofline_open_loop_tape", "step_seq_cfg3", "steps_kernel", 256), ("roofline_at_l3", "step_s1024", "step_kernel", 1),
        ("roofline_at_l3_closed_loop", "rollout_big_s1024", "rollout_big_kernel", 64), ("roofline_at_l3_open_loop_tape", "step_seq_s1024", "steps_kernel", 64),
        ("roofline_hbm", "step_s2048", "step_kernel", 1), ("roofline_hbm_closed_loop", "rollout_big_s2048", "rollout_big_kernel", 32),
        ("roofline_hbm_open_loop_tape", "step_seq_s2048", "steps_kernel", 32)]
print(f"{'leg':32s} {'kernel':20s} {'bytes/launch':>13s} {'rocprof avg us':>15s} {'frac(rocprof)':>13s} {'bench us':>10s} {'frac(bench)':>11s} {'traffic/alg':>11s}")
for key, name, kern, T in rows:
    leg = d[key]
    us, calls = avg_us(name, kern)
    alg = leg["algorithmic_bytes_per_launch"]
    fr = alg / (us * 1e-6) / 8e12
    tr = leg["traffic"] / alg if leg.get("traffic") else float("nan")
    print(f"{key:32s} {kern:20s} {alg/1e6:11.1f}MB {us:15.2f} {fr:13.3f} {leg['launch_us']:10.2f} {leg['frac']:11.3f} {tr:11.3f}")
for key, name in (("cfg4_shard", "cfg4"), ("cfg5_shard", "cfg5")):
    a, _ = avg_us("step_" + name, "step_kernel")
    b, _ = avg_us("step_seq_" + name, "steps_kernel")
    T = 128 if name == "cfg4" else 64
    print(f"{key}: step {a:.2f} us (bench {d[key]['step_us']}), tape {b / T:.2f} us/step (bench {d[key]['tape_us_per_step']}), "
          f"VALU issue fraction of the step launch {d[key].get('valu_issue_frac_step')}")
print("value", d["value"], "ms/iter", d["ms_per_step"], "update frac", d["update_roofline"]["frac"], "resmlp512", d["resmlp512"]["value"],
      d["resmlp512"]["update_roofline"]["frac"], "traffic_source", d.get("traffic_source"))
