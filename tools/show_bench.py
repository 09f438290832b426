#!/usr/bin/env python3
"""Dev tool: the headline fields of a bench.py JSON line (file argument) on a few lines."""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "rollout_ms", d["rollout_ms"], "update_ms", d["update_ms"])
for k in ("update_roofline", "update_roofline_f32", "cfg4_ppo_shard", "cfg5_ppo_shard"):
    if d.get(k):
        v = dict(d[k])
        print(k, json.dumps(v)[:1400])
for k in ("roofline", "roofline_closed_loop", "roofline_hbm", "cfg4_shard", "cfg5_shard", "resmlp512"):
    if d.get(k):
        v = d[k]
        print(k, {q: v[q] for q in ("frac", "launch_us", "us_per_step", "step_us", "tape_us_per_step", "value", "rollout_ms", "update_ms") if q in v})
