#!/usr/bin/env python3
"""HBM-side traffic of the step kernels per launch (run on the GPU box through gpurun):
rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (kernel-trace only), as MI355X_MICROARCH.md prescribes,
for configs[2] (16384 envs, per-env S=128), S=1024 (the Infinity Cache size) and S=2048 (2x the Infinity Cache), each as one launch
per step (step_kernel), one launch per tape (steps_kernel) and closed loop (rollout_big_kernel); and the vector-issue counters of
the shards of BASELINE configs[3] / configs[4] (shared maps: VALU-bound, SURVEY 8d caveat).
Writes gpurun_out/<round>/pmc_traffic.json, tagged with the sha256 of csrc/navsim.hip so that bench.py only quotes it for the
kernel it was measured on.
Corrections (the guide's HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
(16 B / lane) coalesced stream -- the segment stream is >= 94 % of these kernels' reads -- so reads = 2 x FETCH_SIZE."""
import csv, datetime, glob, hashlib, json, os, socket, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TMPDIR"] = "/tmp"


def counters(ctrs, flag, tool="tools/time_step.py", kernel="step_kernel"):
    """one rocprofv3 --pmc pass; mean per launch of every counter in `ctrs` over the steady-state launches of `kernel`"""
    d = f"/tmp/pmc_tr_{ctrs.replace(' ', '_')[:40]}_{kernel}_{flag.strip('-').replace('=', '')}"
    subprocess.run(["rm", "-rf", d])
    subprocess.run(["rocprofv3", "--pmc"] + ctrs.split() + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                    sys.executable, os.path.join(R, tool), flag], cwd="/tmp", stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, check=False)
    f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    out = {}
    for c in ctrs.split():
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if kernel + "<" in r["Kernel_Name"] and r["Counter_Name"] == c]
        vals = vals[len(vals) // 4:]   # steady state
        out[c] = sum(vals) / len(vals)
    return out


def counter(ctr, flag, tool="tools/time_step.py", kernel="step_kernel"):
    return counters(ctr, flag, tool, kernel)[ctr]


RND = os.environ.get("ROUND", "r04")
out = {"recorded_utc": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"), "recorded_on": socket.gethostname(),
       "run_id": os.environ.get("PROF_RUN_ID", ""),   # tools/prof_all.sh: the same id is written next to the bench line of that call
       "navsim_hip_sha256": hashlib.sha256(open(os.path.join(R, "navbot_ppo_amd/csrc/navsim.hip"), "rb").read()).hexdigest(),
       "_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; KiB; reads = 2 x FETCH_SIZE (gfx950 wide-stream correction)"}
SIZES = (("cfg3", "--cfg3", 128, 256), ("s1024", "--s=1024", 1024, 64), ("s2048", "--s=2048", 2048, 32))
for key, flag, S, T in SIZES:
    alg = 16384 * (134 + 16 * S)
    f, w = counter("FETCH_SIZE", flag), counter("WRITE_SIZE", flag)
    out[key + "_step"] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "algorithmic_bytes_per_launch": alg}
    out[key + "_step_bytes_per_launch"] = int((2 * f + w) * 1024)
    # navsim_step_seq (steps_kernel): one launch = T steps of tools/time_step_seq.py's tape
    f, w = (counter(c, flag, "tools/time_step_seq.py", "steps_kernel") for c in ("FETCH_SIZE", "WRITE_SIZE"))
    out[key + "_seq"] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "steps_per_launch": T,
                         "algorithmic_bytes_per_launch": T * alg}
    out[key + "_seq_bytes_per_launch"] = int((2 * f + w) * 1024)
    # navsim_rollout_mlp64 (rollout_big_kernel): one launch = T closed-loop steps of tools/time_rollout.py
    rflag = {"cfg3": "--cfg3", "s1024": "--s1024", "s2048": "--s2048"}[key]
    f, w = (counter(c, rflag, "tools/time_rollout.py", "rollout_big_kernel") for c in ("FETCH_SIZE", "WRITE_SIZE"))
    out[key + "_closed_loop"] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "steps_per_launch": T,
                                 "algorithmic_bytes_per_launch": T * alg}
    out[key + "_closed_loop_bytes_per_launch"] = int((2 * f + w) * 1024)
# TCC hit / miss of the S = 1024 and S = 2048 step launches (how much of the stream the L2 serves: none expected)
for key, flag in (("s1024", "--s=1024"), ("s2048", "--s=2048")):
    try:
        out[key + "_step_tcc"] = {k: round(v, 1) for k, v in counters("TCC_HIT_sum TCC_MISS_sum", flag).items()}
    except Exception as e:   # the counters may not fit one pass on this box
        out[key + "_step_tcc"] = {"error": str(e)[:200]}
# one GPU's shard of BASELINE configs[3] (4096 envs, stage_4, 36 beams) and configs[4] (8192 envs, house map 2048 segments, f16):
for key, flag in (("cfg4", "--cfg4"), ("cfg5", "--cfg5")):
    c = counters("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE", flag)
    out[key + "_valu"] = {k: round(v, 1) for k, v in c.items()}
    # SQ_ACTIVE_INST_VALU counts quad-cycles: x 4 = SIMD cycles spent issuing vector instructions, summed over the 1024 SIMDs
    out[key + "_valu"]["valu_busy_us_per_simd_at_2p4GHz"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / 2400.0, 3)
    out[key + "_valu"]["note"] = ("per step_kernel launch; vector-issue fraction of a launch = valu_busy_us_per_simd / the launch's duration "
                                  "(bench.py divides by the duration it measures)")
os.makedirs(os.path.join(R, "gpurun_out", RND), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", RND, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
