# round 4: the matrix-core spatial stencil, both forms, against the grouped ring kernel: timings (512 x 2048^2 + uint8 mask) and SQ counters
# (256 x 2048^2) -> gpurun_out/mfma_r04/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mfma_r04
mkdir -p $O
{
  echo "# python tools/bench_mfma.py 512   (512 x 2048 x 2048 float32 + 80 %-valid uint8 mask, Gaussian2DKernel(FWHM 8 px) = 29 x 29 taps; HIP events, median of 5)"
  echo "## SPC_SPATIAL_MFMA_FORM=1: numerator on the vector ALU (band kernel, LDS staging), denominator on v_mfma_f32_16x16x32_f16"
  SPC_SPATIAL_MFMA_FORM=1 python $R/tools/bench_mfma.py 512 2>&1 | tail -4
  echo "## default (form 2): numerator on v_mfma_f32_16x16x4_f32 as well, wave-private regions"
  python $R/tools/bench_mfma.py 512 2>&1 | tail -4
} > $O/masked_spatial_mfma.log
for spec in "spconv_mask spatial_sep_grouped 0" "spmfma_mom spatial_sep_mfma 1" "spmfma_mom spatial_mfma2 2"; do
  set -- $spec
  i=0
  for cset in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
              "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
              "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
              "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    SPC_SPATIAL_MFMA_FORM=$3 REPS=2 timeout 200 rocprofv3 --pmc $cset --kernel-trace --output-format csv -d $O/$1_$3_$i -- python $R/tools/prof_one.py $1 256 2048 2048 > $O/$1_$3_$i.log 2>&1
  done
  echo "== $1 (form $3): kernel $2, 256 x 2048^2 + uint8 mask; FETCH_SIZE / WRITE_SIZE in KiB (FETCH_SIZE x 2 on gfx950)"
  cd $R && python tools/pmc_summary.py "gpurun_out/mfma_r04/$1_$3_*/*/*counter_collection.csv" "$2"; cd /tmp
done > $O/masked_spatial_mfma_pmc.txt 2>&1
cat $O/masked_spatial_mfma.log; cat $O/masked_spatial_mfma_pmc.txt
