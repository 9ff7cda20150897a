# round-2 counter passes for the kernels that changed: all-valid and masked spatial stencil, masked spectral stencil
# (table denominators), register-resident selection.  Summaries land in gpurun_out/pmc_sq/*.summary.txt
R=$GRAFT_REPO_ROOT
for spec in "spconv:spatial_sep_fast" "spconv_mask:spatial_sep_kernel" "sconv_mask:spectral_conv_kernel" "median:select_reg"; do
  export OP=${spec%%:*} KERN=${spec##*:} SHAPE="1024 1024 1024"
  bash $R/tools/pmc_sq.sh > $R/gpurun_out/pmc_sq_${OP}.summary.txt 2>&1
done
