# PMC passes of the split form of the masked spatial stencil (fused moment 0), 256 x 2048^2 + uint8 mask
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_split
mkdir -p $O
TAG=${1:-split}
i=0
for cset in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
            "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  REPS=2 timeout 200 rocprofv3 --pmc $cset --kernel-trace --output-format csv -d $O/${TAG}_$i -- python $R/tools/prof_one.py spmfma_mom 256 2048 2048 > $O/${TAG}_$i.log 2>&1
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmc_split/${TAG}_*/*/*counter_collection.csv" "spatial_split" | tee gpurun_out/pmc_split/${TAG}_summary.txt
