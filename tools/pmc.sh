#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <name> "<counters>" -- <command...>
# one rocprofv3 --pmc pass (kernel-trace only), aggregated per kernel into gpurun_out/${ROUND:-r04}/<name>_pmc.txt
name="$1"; ctrs="$2"; shift; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/pmc_$name
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- "$@" > /tmp/pmc_$name.log 2>&1
mkdir -p gpurun_out/${ROUND:-r04}
python3 - "$name" "${ROUND:-r04}" <<'PY'
import csv, sys, collections, glob
name, rnd = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/pmc_{name}/**/p_counter_collection.csv", recursive=True)
if not f:
    print(open(f"/tmp/pmc_{name}.log").read()[-3000:]); sys.exit(1)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"], r["Counter_Name"])
    if r["Counter_Name"] == list(agg[k].keys())[0] and key not in seen:
        calls[k] += 1; seen.add(key)
with open(f"gpurun_out/{rnd}/{name}_pmc.txt", "w") as out:
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:8]:
        line = f"{k:60s} calls={calls[k]:5d} " + " ".join(f"{c}={v / max(calls[k], 1):.4g}" for c, v in d.items())
        print(line); out.write(line + "\n")
PY
