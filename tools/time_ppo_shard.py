#!/usr/bin/env python3
"""Dev tool: one GPU's shard of BASELINE configs[3] / configs[4] as a PPO workload (bench.py: cfg4_ppo_shard / cfg5_ppo_shard) on its own --
for rocprofv3 --kernel-trace --stats (tools/prof_r05.sh).  usage: python tools/time_ppo_shard.py cfg4|cfg5 [iterations]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if which == "cfg4":
    out = bench.ppo_shard_leg(4096, "stage_4", 36, False, None, 512, 50, "configs[3] per GPU", steps=steps)
else:
    out = bench.ppo_shard_leg(8192, "house", 10, True, "small_house", 512, 50, "configs[4] per GPU", steps=steps)
print(json.dumps(out))
