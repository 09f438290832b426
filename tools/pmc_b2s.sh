# Round 6: counters of resmlp_bwd2s (and its neighbours) in three --pmc passes -> gpurun_out/r06/b2s_{a,b,c}_pmc.txt
cd $GRAFT_REPO_ROOT
export ROUND=r06
bash tools/pmc.sh b2s_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS" -- python tools/time_update_resmlp.py 2>&1 | grep -i "resmlp_bwd"
bash tools/pmc.sh b2s_b "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY" -- python tools/time_update_resmlp.py 2>&1 | grep -i "resmlp_bwd"
bash tools/pmc.sh b2s_c "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" -- python tools/time_update_resmlp.py 2>&1 | grep -i "resmlp_bwd"
