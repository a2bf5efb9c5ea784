cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/attn_pmc/g$i -- python3 tools/probes/train_host/attn_once.py > gpurun_out/attn_pmc_$i.log 2>&1
  f=$(ls -t gpurun_out/attn_pmc/g$i/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/summarize_pmc.py "$f" | grep "k_attn" 
done
rm -rf gpurun_out/attn_pmc
