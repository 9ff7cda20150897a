# FETCH_SIZE + L2 hit counters of the split kernel (fused moment 0), 256 x 2048^2 + uint8 mask
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_split
mkdir -p $O
TAG=${1:-fetch}
i=0
for cset in "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  REPS=2 timeout 200 rocprofv3 --pmc $cset --kernel-trace --output-format csv -d $O/${TAG}_$i -- python $R/tools/prof_one.py spmfma_mom 256 2048 2048 > $O/${TAG}_$i.log 2>&1
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmc_split/${TAG}_*/*/*counter_collection.csv" "spatial_split" | tee gpurun_out/pmc_split/${TAG}_summary.txt
