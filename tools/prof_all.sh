# Everything under profiles/r03_* in ONE gpurun call (same box): usage  bash tools/prof_all.sh
export ROUND=${ROUND:-r03}
export PROF_RUN_ID="$(date -u +%Y%m%dT%H%M%SZ)-$(hostname)"
O=gpurun_out/$ROUND
mkdir -p $O
echo "$PROF_RUN_ID" > $O/run_id.txt
# traffic first: bench.py quotes profiles/pmc_traffic.json only while it carries the sha256 of the current csrc/navsim.hip
python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1 && cp $O/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > $O/bench_final.json 2> $O/bench_final.err
tools/prof_stats.sh bench_final -- python bench.py > $O/bench_final_under_rocprof.log 2>&1
tools/prof_stats.sh step_cfg3 -- python tools/time_step.py --cfg3 > $O/step_cfg3_rocprof.log 2>&1
tools/prof_stats.sh step_s1024 -- python tools/time_step.py --s=1024 > $O/step_s1024_rocprof.log 2>&1
tools/prof_stats.sh step_seq_cfg3 -- python tools/time_step_seq.py --cfg3 > $O/step_seq_cfg3_rocprof.log 2>&1
tools/prof_stats.sh step_seq_s1024 -- python tools/time_step_seq.py --s=1024 > $O/step_seq_s1024_rocprof.log 2>&1
TR_SIZES=16384 TR_MAP=stage_2 TR_PER_ENV=1 TR_T=256 tools/prof_stats.sh rollout_big_cfg3 -- python tools/time_rollout.py > $O/rollout_big_cfg3_rocprof.log 2>&1
tools/prof_stats.sh rollout_big_s1024 -- python tools/time_rollout.py --s1024 > $O/rollout_big_s1024_rocprof.log 2>&1
tools/prof_stats.sh update -- python tools/time_update.py navbot_ppo_amd/libnavsim.so > $O/update_rocprof.log 2>&1
tools/prof_stats.sh resmlp512_update -- python tools/time_update_resmlp.py 2097152 5 > $O/resmlp512_update_rocprof.log 2>&1
tools/pmc.sh step_final "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" -- python tools/time_step.py --cfg3 > /dev/null 2>&1
tools/pmc.sh step_final_b "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE" -- python tools/time_step.py --cfg3 > /dev/null 2>&1
tools/pmc.sh step_seq_final "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" -- python tools/time_step_seq.py --cfg3 > /dev/null 2>&1
TR_SIZES=16384 TR_MAP=stage_2 TR_PER_ENV=1 TR_T=256 tools/pmc.sh rollout_big "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS" -- python tools/time_rollout.py > /dev/null 2>&1
tools/pmc.sh update_final "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" -- python tools/time_update.py navbot_ppo_amd/libnavsim.so > /dev/null 2>&1
tools/pmc.sh resmlp512_update "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" -- python tools/time_update_resmlp.py 2097152 2 > /dev/null 2>&1
python tools/time_rollout.py > $O/rollout_shard_sizes.txt 2>&1
(TR_SIZES=16384 TR_MAP=stage_2 TR_PER_ENV=1 TR_T=256 python tools/time_rollout.py; NAVSIM_EPB=16 TR_SIZES=16384 TR_MAP=stage_2 TR_PER_ENV=1 TR_T=256 python tools/time_rollout.py
 TR_SIZES=16384 TR_PER_ENV=1 TR_T=64 TR_SIDES=248 python tools/time_rollout.py; TR_SIZES=16384 TR_T=256 python tools/time_rollout.py; NAVSIM_EPB=16 TR_SIZES=16384 TR_T=256 python tools/time_rollout.py
 TR_SIZES=16384 TR_MAP=house TR_T=64 python tools/time_rollout.py; python tools/time_step_seq.py --cfg3
 TR_SIZES=4096,4608,8192,12288 TR_T=256 python tools/time_rollout.py; NAVSIM_EPB=16 TR_SIZES=4608,8192,12288 TR_T=256 python tools/time_rollout.py; NAVSIM_EPB=64 TR_SIZES=4096 TR_T=256 python tools/time_rollout.py) 2>&1 | grep -v amdgpu > $O/rollout_big.txt
python tools/time_update_scale.py > $O/update_scale.txt 2>&1
python tools/time_rtg.py > $O/time_rtg.log 2>&1
python tools/phase_timing.py build/libnavsim_timing.so > $O/step_cfg3_phase_stamps.txt 2>&1
python tools/time_to_reward.py resmlp512 > $O/time_to_reward_resmlp512.txt 2>&1
tail -c 2500 $O/bench_final.json; echo; tail -3 $O/step_cfg3_rocprof.log; tail -3 $O/step_s1024_rocprof.log; head -8 $O/resmlp512_update_rocprof.log
cat $O/step_final_pmc.txt | head -1; cat $O/resmlp512_update_pmc.txt | head -5; cat $O/rollout_shard_sizes.txt $O/update_scale.txt | grep -v amdgpu
