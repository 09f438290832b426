mkdir -p gpurun_out/r02
# traffic first: bench.py quotes profiles/pmc_traffic.json only while it carries the sha256 of the current csrc/navsim.hip
python tools/pmc_traffic.py > gpurun_out/r02/pmc_traffic.log 2>&1 && cp gpurun_out/r02/pmc_traffic.json profiles/pmc_traffic.json
python tools/time_rtg.py > gpurun_out/r02/time_rtg.log 2>&1
python bench.py > gpurun_out/r02/bench_final.json 2> gpurun_out/r02/bench_final.err
tools/prof_stats.sh bench_final -- python bench.py > gpurun_out/r02/bench_final_under_rocprof.log 2>&1
tools/prof_stats.sh step_cfg3 -- python tools/time_step.py --cfg3 > gpurun_out/r02/step_cfg3_rocprof.log 2>&1
tools/prof_stats.sh step_s1024 -- python tools/time_step.py --s=1024 > gpurun_out/r02/step_s1024_rocprof.log 2>&1
tools/prof_stats.sh update -- python tools/time_update.py navbot_ppo_amd/libnavsim.so > gpurun_out/r02/update_rocprof.log 2>&1
tools/pmc.sh step_final "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" -- python tools/time_step.py --cfg3 > /dev/null 2>&1
tools/pmc.sh update_final "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" -- python tools/time_update.py navbot_ppo_amd/libnavsim.so > /dev/null 2>&1
tail -c 3000 gpurun_out/r02/bench_final.json; echo; tail -2 gpurun_out/r02/time_rtg.log; tail -3 gpurun_out/r02/step_cfg3_rocprof.log; tail -3 gpurun_out/r02/step_s1024_rocprof.log; tail -6 gpurun_out/r02/update_rocprof.log; cat gpurun_out/r02/step_final_pmc.txt | head -2; cat gpurun_out/r02/update_final_pmc.txt | head -3
