cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_w -- python $GRAFT_REPO_ROOT/tools/bench_wide_ops.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/prof_w/*/*kernel_stats.csv | head -1)
head -12 $f | cut -c1-200
