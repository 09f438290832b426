#!/usr/bin/env python3
"""Dev tool (runs here, after a `bash tools/prof_r06.sh` gpurun call): copies the round's summaries from gpurun_out/r06/ into profiles/r06_*
(and the refreshed profiles/pmc_traffic.json) and prints the headline figures profiles/README.md quotes."""
import json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out", "r06"), os.path.join(R, "profiles")
for a in ("run_id.txt", "bench_legs.csv", "update_arith_kernel_stats.csv", "update_arith_hip_events.txt", "resmlp_update_kernel_stats.csv", "step_cfg3_kernel_stats.csv",
          "resmlp_update_hip_events.txt", "resmlp_epoch_timeline.txt", "x3s_pmc.txt", "b2s_pmc.txt", "mfma16_stream.txt", "bf16x3_error_kinkfree.txt",
          "ppo_cfg4.json", "ppo_cfg5.json"):
    if os.path.exists(os.path.join(O, a)):
        shutil.copy(os.path.join(O, a), os.path.join(P, "r06_" + a))
    else:
        print("missing:", a)
shutil.copy(os.path.join(O, "pmc_traffic.json"), os.path.join(P, "pmc_traffic.json"))
line = [l for l in open(os.path.join(O, "bench_final.json")).read().splitlines() if l.startswith("{")][-1]
open(os.path.join(P, "r06_bench_final.json"), "w").write(line + "\n")
d = json.loads(line)
print(f"headline {d['value'] / 1e6:.2f} M env-steps/s, {d['ms_per_step']:.2f} ms per iteration (rollout {d['rollout_ms']}, update {d['update_ms']})")
print(f"roofline {d['roofline']['frac']:.4f} ({d['roofline']['launch_us']} us, traffic {d['roofline']['traffic']}) source: {d['roofline'].get('time_source')}")
u = d["update_roofline"]
print(f"update_roofline {u['frac']:.4f} of {u['peak']} {u['unit']}, epoch {u['epoch_us']} us")
r = d["resmlp512"]
print(f"resmlp512 {r['value'] / 1e6:.2f} M env-steps/s, epoch {r['update_roofline']['epoch_ms']} ms = {r['update_roofline']['frac']} algorithmic, {r['update_roofline']['executed']['frac']} executed")
for k in ("cfg4_ppo_shard", "cfg5_ppo_shard", "cfg5_ppo_shard_resmlp512"):
    print(k, f"{d[k]['value'] / 1e6:.2f} M env-steps/s per GPU")
