# Round 5: everything under profiles/r05_* in ONE gpurun call (same box).  usage: bash tools/prof_r05.sh
export ROUND=r05
mkdir -p gpurun_out/r05
bash tools/prof_all.sh > gpurun_out/r05/prof_all.log 2>&1
O=gpurun_out/r05
# the update pass on both arithmetics (f32-input MFMA / split bf16), the 42-column pass, and their counters
tools/prof_stats.sh update_arith -- python tools/time_update_arith.py > $O/update_arith_rocprof.log 2>&1
python tools/time_update_arith.py 2>&1 | grep -v amdgpu > $O/update_arith_hip_events.txt
tools/prof_stats.sh update_wide -- python tools/time_update_wide.py > $O/update_wide_rocprof.log 2>&1
python tools/time_update_wide.py 2>&1 | grep -v amdgpu > $O/update_wide_hip_events.txt
bash tools/pmc_x3.sh > $O/bf16x3_pmc_raw.txt 2>&1
cat $O/x3_a_pmc.txt $O/x3_b_pmc.txt > $O/bf16x3_pmc.txt
python tools/bf16x3_error.py 2>&1 | grep -v amdgpu > $O/bf16x3_error.txt
# one GPU's shard of configs[3] / configs[4] as PPO workloads
tools/prof_stats.sh ppo_cfg4 -- python tools/time_ppo_shard.py cfg4 > $O/ppo_cfg4_rocprof.log 2>&1
tools/prof_stats.sh ppo_cfg5 -- python tools/time_ppo_shard.py cfg5 > $O/ppo_cfg5_rocprof.log 2>&1
python tools/time_ppo_shard.py cfg4 2>/dev/null | tail -1 > $O/ppo_cfg4.json
python tools/time_ppo_shard.py cfg5 2>/dev/null | tail -1 > $O/ppo_cfg5.json
./build/bf16_overlap > $O/bf16_mfma_valu_overlap.txt 2>&1
./build/bf16_fillers > $O/bf16_mfma_fillers.txt 2>&1
./build/bf16_split_ops > $O/bf16_split_ops.txt 2>&1
python tools/show_info.py 2>&1 | grep -v amdgpu > $O/selected_instantiations.txt
python tools/time_house_rollout.py 2>&1 | grep -v amdgpu > $O/house_rollout_shapes.txt
# the closed-loop rollout of the 512-wide actor: one persistent launch against the hipGraph of per-step launches, its kernel average, its phases
python tools/time_rollout_resmlp.py 2>&1 | grep -v amdgpu > $O/rollout_resmlp_times.txt
tools/prof_stats.sh rollout_resmlp -- python tools/time_rollout_resmlp.py > $O/rollout_resmlp_rocprof.log 2>&1
[ -f build/libnavsim_resmlp_phases.so ] && python tools/resmlp_rollout_phases.py 2>&1 | grep -v amdgpu > $O/rollout_resmlp_phases.txt
tail -c 600 $O/bench_final.json; echo; cat $O/update_arith_hip_events.txt $O/update_wide_hip_events.txt
