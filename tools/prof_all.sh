# Everything under profiles/r04_* in ONE gpurun call (same box): usage  bash tools/prof_all.sh
export ROUND=${ROUND:-r04}
export PROF_RUN_ID="$(date -u +%Y%m%dT%H%M%SZ)-$(hostname)"
O=gpurun_out/$ROUND
mkdir -p $O
echo "$PROF_RUN_ID" > $O/run_id.txt
# counters first: bench.py --with-pmc-file quotes THIS call's file (same kernels: the file carries the sha256 of csrc/navsim.hip)
python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1
python bench.py --with-pmc-file $O/pmc_traffic.json > $O/bench_final.json 2> $O/bench_final.err
tools/prof_stats.sh bench_final -- python bench.py --with-pmc-file $O/pmc_traffic.json > $O/bench_final_under_rocprof.log 2>&1
# the three forms of the step body at configs[2], at the Infinity Cache size and at 2x it
for s in cfg3 s=1024 s=2048; do n=${s/=/}; n=${n/cfg3/cfg3}
  tools/prof_stats.sh step_$n -- python tools/time_step.py --$s > $O/step_${n}_rocprof.log 2>&1
  tools/prof_stats.sh step_seq_$n -- python tools/time_step_seq.py --$s > $O/step_seq_${n}_rocprof.log 2>&1
done
tools/prof_stats.sh rollout_big_cfg3 -- python tools/time_rollout.py --cfg3 > $O/rollout_big_cfg3_rocprof.log 2>&1
tools/prof_stats.sh rollout_big_s1024 -- python tools/time_rollout.py --s1024 > $O/rollout_big_s1024_rocprof.log 2>&1
tools/prof_stats.sh rollout_big_s2048 -- python tools/time_rollout.py --s2048 > $O/rollout_big_s2048_rocprof.log 2>&1
# the same S = 1024 / 2048 legs WITHOUT the profiler in the same call (HIP events): is the gap to the rocprof average clock or cache?
(python tools/time_step.py --s=1024 --s=2048; python tools/time_step_seq.py --s=1024 --s=2048; python tools/time_rollout.py --s1024; python tools/time_rollout.py --s2048) 2>&1 | grep -v amdgpu > $O/hbm_legs_hip_events.txt
# one GPU's shard of BASELINE configs[3] / configs[4]
tools/prof_stats.sh step_cfg4 -- python tools/time_step.py --cfg4 > $O/step_cfg4_rocprof.log 2>&1
tools/prof_stats.sh step_cfg5 -- python tools/time_step.py --cfg5 > $O/step_cfg5_rocprof.log 2>&1
tools/prof_stats.sh step_seq_cfg4 -- python tools/time_step_seq.py --cfg4 > $O/step_seq_cfg4_rocprof.log 2>&1
tools/prof_stats.sh step_seq_cfg5 -- python tools/time_step_seq.py --cfg5 > $O/step_seq_cfg5_rocprof.log 2>&1
tools/prof_stats.sh update -- python tools/time_update.py navbot_ppo_amd/libnavsim.so > $O/update_rocprof.log 2>&1
tools/prof_stats.sh resmlp512_update -- python tools/time_update_resmlp.py 2097152 5 > $O/resmlp512_update_rocprof.log 2>&1
C1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU"
tools/pmc.sh step "$C1" -- python tools/time_step.py --cfg3 > /dev/null 2>&1
tools/pmc.sh step_seq "$C1" -- python tools/time_step_seq.py --cfg3 > /dev/null 2>&1
tools/pmc.sh step_cfg4 "$C1" -- python tools/time_step.py --cfg4 > /dev/null 2>&1
tools/pmc.sh step_cfg5 "$C1" -- python tools/time_step.py --cfg5 > /dev/null 2>&1
tools/pmc.sh rollout_big "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS" -- python tools/time_rollout.py --cfg3 > /dev/null 2>&1
tools/pmc.sh rollout16 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS" -- env TR_SIZES=4096 python tools/time_rollout.py > /dev/null 2>&1
C2="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
tools/pmc.sh update "$C2" -- python tools/time_update.py navbot_ppo_amd/libnavsim.so > /dev/null 2>&1
tools/pmc.sh resmlp512_update "$C2" -- python tools/time_update_resmlp.py 2097152 2 > /dev/null 2>&1
python tools/time_rollout.py 2>&1 | grep -v amdgpu > $O/rollout_shard_sizes.txt
python tools/time_update_scale.py 2>&1 | grep -v amdgpu > $O/update_scale.txt
python tools/time_rtg.py 2>&1 | grep -v amdgpu > $O/time_rtg.txt
tail -c 1500 $O/bench_final.json; echo; cat $O/hbm_legs_hip_events.txt; for f in $O/step_cfg3_kernel_stats.csv $O/step_s2048_kernel_stats.csv $O/step_seq_s2048_kernel_stats.csv; do head -3 $f; done
