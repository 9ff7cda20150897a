# SQ / LDS counters of one op (rocprofv3 --pmc passes, 8 SQ counters each): OP, SHAPE, KERN from the environment
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_sq
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  REPS=2 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq/${OP}_$i -- python $R/tools/prof_one.py $OP $SHAPE > $R/gpurun_out/pmc_sq/${OP}_$i.log 2>&1
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmc_sq/${OP}_*/*/*counter_collection.csv" "$KERN"
