# Round 6: everything under profiles/r06_* (and the refreshed profiles/pmc_traffic.json) in ONE gpurun call (same box).
# usage: bash tools/prof_r06.sh      then: python tools/collect_profiles.py r06   (copies the summaries into profiles/)
export ROUND=r06
export PROF_RUN_ID="$(date -u +%Y%m%dT%H%M%SZ)-$(hostname)"
O=gpurun_out/r06
mkdir -p $O
echo "$PROF_RUN_ID" > $O/run_id.txt
# counters first: bench.py --with-pmc-file quotes THIS call's file (the file carries the sha256 of csrc/navsim.hip)
python tools/pmc_traffic.py > $O/pmc_traffic.log 2>&1
NAVBOT_BENCH_SAVE_PROF=$O/bench_legs.csv python bench.py --with-pmc-file $O/pmc_traffic.json > $O/bench_final.json 2> $O/bench_final.err
# the step kernel at configs[2] on its own (the stand-alone cross-check of the line's `roofline` leg: same kernel, graph replays of 64)
tools/prof_stats.sh step_cfg3 -- python tools/time_step.py --cfg3 > $O/step_cfg3_rocprof.log 2>&1
# the update pass of the timed workload (both arithmetics) and of the 512-wide nets: kernel averages, HIP events, counters, epoch timeline
tools/prof_stats.sh update_arith -- python tools/time_update_arith.py > $O/update_arith_rocprof.log 2>&1
python tools/time_update_arith.py 2>&1 | grep -v amdgpu > $O/update_arith_hip_events.txt
tools/prof_stats.sh resmlp_update -- python tools/time_update_resmlp.py > $O/resmlp_update_rocprof.log 2>&1
python tools/time_update_resmlp.py 2>&1 | grep -v amdgpu > $O/resmlp_update_hip_events.txt
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_gaps -o p -- python tools/time_update_resmlp.py > /tmp/tr_gaps.log 2>&1; python tools/epoch_gaps.py $(find /tmp/tr_gaps -name "*kernel_trace.csv" | head -1) > $O/resmlp_epoch_timeline.txt 2>&1)
bash tools/pmc_x3s.sh > $O/x3s_pmc_raw.txt 2>&1
cat $O/x3s_a_pmc.txt $O/x3s_b_pmc.txt $O/x3s_c_pmc.txt > $O/x3s_pmc.txt
bash tools/pmc_b2s.sh > $O/b2s_pmc_raw.txt 2>&1
cat $O/b2s_a_pmc.txt $O/b2s_b_pmc.txt $O/b2s_c_pmc.txt | grep bwd2s > $O/b2s_pmc.txt
[ -x build/mfma16_stream ] && ./build/mfma16_stream > $O/mfma16_stream.txt 2>&1
python tools/bf16x3_error_kinkfree.py 2>&1 | grep -v amdgpu > $O/bf16x3_error_kinkfree.txt
# one GPU's shard of configs[3] / configs[4] as PPO workloads
python tools/time_ppo_shard.py cfg4 2>/dev/null | tail -1 > $O/ppo_cfg4.json
python tools/time_ppo_shard.py cfg5 2>/dev/null | tail -1 > $O/ppo_cfg5.json
tail -c 400 $O/bench_final.json; echo; cat $O/update_arith_hip_events.txt $O/resmlp_update_hip_events.txt $O/resmlp_epoch_timeline.txt
