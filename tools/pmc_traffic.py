#!/usr/bin/env python3
"""HBM-side traffic of the step kernel per launch (run on the GPU box through gpurun):
rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (kernel-trace only), as MI355X_MICROARCH.md prescribes,
for configs[2] (16384 envs, per-env S=128) and the beyond-L3 point (S=1024).  Writes gpurun_out/<round>/pmc_traffic.json, tagged
with the sha256 of csrc/navsim.hip so that bench.py only quotes it for the kernel it was measured on.
Corrections (the guide's HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
(16 B / lane) coalesced stream -- the segment stream is >= 94 % of this kernel's reads -- so reads = 2 x FETCH_SIZE."""
import csv, glob, hashlib, json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TMPDIR"] = "/tmp"

def counter(ctr, flag, tool="tools/time_step.py", kernel="step_kernel"):
    d = f"/tmp/pmc_tr_{ctr}_{kernel}_{flag.strip('-').replace('=', '')}"
    subprocess.run(["rm", "-rf", d])
    subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                    sys.executable, os.path.join(R, tool), flag], cwd="/tmp", stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, check=False)
    f = glob.glob(d + "/**/p_counter_collection.csv", recursive=True)
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if kernel + "<" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    vals = vals[len(vals) // 4:]   # steady state
    return sum(vals) / len(vals)

import datetime, socket
RND = os.environ.get("ROUND", "r03")
out = {"recorded_utc": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"), "recorded_on": socket.gethostname(),
       "run_id": os.environ.get("PROF_RUN_ID", ""),   # tools/prof_all.sh: the same id is written next to the bench line of that call
       "navsim_hip_sha256": hashlib.sha256(open(os.path.join(R, "navbot_ppo_amd/csrc/navsim.hip"), "rb").read()).hexdigest(),
       "_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; KiB; reads = 2 x FETCH_SIZE (gfx950 wide-stream correction)"}
for key, flag, alg in (("cfg3", "--cfg3", 16384 * (134 + 16 * 128)), ("s1024", "--s=1024", 16384 * (134 + 16 * 1024))):
    f, w = counter("FETCH_SIZE", flag), counter("WRITE_SIZE", flag)
    out[key + "_step"] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "algorithmic_bytes_per_launch": alg}
    out[key + "_step_bytes_per_launch"] = int((2 * f + w) * 1024)
# navsim_step_seq (steps_kernel): one launch = T steps of tools/time_step_seq.py's tape (256 at configs[2], 64 at S=1024)
for key, flag, T, S in (("cfg3", "--cfg3", int(os.environ.get("TS_T", "256")), 128), ("s1024", "--s=1024", 64, 1024)):
    f, w = (counter(c, flag, "tools/time_step_seq.py", "steps_kernel") for c in ("FETCH_SIZE", "WRITE_SIZE"))
    out[key + "_seq"] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "steps_per_launch": T,
                         "algorithmic_bytes_per_launch": T * 16384 * (134 + 16 * S)}
    out[key + "_seq_bytes_per_launch"] = int((2 * f + w) * 1024)
# navsim_rollout_mlp64 at configs[2] (rollout_big_kernel): one launch = 256 closed-loop steps of tools/time_rollout.py --cfg3
f, w = (counter(c, "--cfg3", "tools/time_rollout.py", "rollout_big_kernel") for c in ("FETCH_SIZE", "WRITE_SIZE"))
out["cfg3_closed_loop"] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "steps_per_launch": 256,
                           "algorithmic_bytes_per_launch": 256 * 16384 * (134 + 16 * 128)}
out["cfg3_closed_loop_bytes_per_launch"] = int((2 * f + w) * 1024)
f, w = (counter(c, "--s1024", "tools/time_rollout.py", "rollout_big_kernel") for c in ("FETCH_SIZE", "WRITE_SIZE"))
out["s1024_closed_loop"] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "steps_per_launch": 64,
                            "algorithmic_bytes_per_launch": 64 * 16384 * (134 + 16 * 1024)}
out["s1024_closed_loop_bytes_per_launch"] = int((2 * f + w) * 1024)
os.makedirs(os.path.join(R, "gpurun_out", RND), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", RND, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
