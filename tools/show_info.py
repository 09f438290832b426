#!/usr/bin/env python3
"""Dev tool: navsim_get_info for the BASELINE configurations (shards and full single-GPU sizes): the (envs, waves, cast variant) each entry point
launches and the registers / scratch / LDS of the selected step and tape instantiations as the loaded code object reports them
(profiles/r05_selected_instantiations.txt).  usage: python tools/show_info.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navbot_ppo_amd.env import VecEnv
cases = [("configs[1]: 4096 envs, stage_1", dict(n_envs=4096)),
         ("configs[2]: 16384 envs, per-env stage_2", dict(n_envs=16384, map="stage_2", per_env_map=True)),
         ("configs[3] shard: 4096 envs, stage_4, 36 beams", dict(n_envs=4096, map="stage_4", n_beams=36)),
         ("configs[3] full: 32768 envs, stage_4, 36 beams", dict(n_envs=32768, map="stage_4", n_beams=36)),
         ("configs[4] shard: 8192 envs, house, f16, tables", dict(n_envs=8192, map="house", sampler="small_house", obs_f16=True)),
         ("configs[4] full: 65536 envs, house, f16, tables", dict(n_envs=65536, map="house", sampler="small_house", obs_f16=True)),
         ("house, 2048 envs", dict(n_envs=2048, map="house", sampler="small_house"))]
keys = ("step_epb", "step_waves", "step_cast", "step_vgprs", "step_scratch_bytes", "step_lds_bytes", "seq_epb", "seq_waves", "seq_cast", "seq_vgprs",
        "seq_scratch_bytes", "seq_lds_bytes", "rollout_kind", "rollout_epb", "rollout_waves", "rollout_cast")
print("cast: 0 = 64-segment passes, 1 = 128-segment passes, 2 = + non-temporal loads, 3 = tile boxes; rollout_kind: 1 = rollout_kernel, 2 = rollout_big_kernel")
for name, kw in cases:
    e = VecEnv(**kw)
    i = e.sim.info()
    e.close()
    print(f"{name}:\n   " + "  ".join(f"{k}={i[k]}" for k in keys))
