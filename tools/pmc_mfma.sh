# SQ counters of the MFMA-denominator spatial kernel against the grouped ring kernel (256 x 2048^2 + uint8 mask)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mfma
mkdir -p $O
for spec in "spconv_mask spatial_sep_grouped" "spmfma_mom spatial_sep_mfma" "spmfma_store spatial_sep_mfma"; do
  set -- $spec
  i=0
  for cset in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
              "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
              "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
              "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    REPS=2 timeout 200 rocprofv3 --pmc $cset --kernel-trace --output-format csv -d $O/$1_$i -- python $R/tools/prof_one.py $1 256 2048 2048 > $O/$1_$i.log 2>&1
  done
  echo "== $1"
  cd $R && python tools/pmc_summary.py "gpurun_out/pmc_mfma/$1_*/*/*counter_collection.csv" "$2"; cd /tmp
done
