# rocprofv3 evidence for bench.py, round 3 (run on the GPU box through gpurun): kernel trace + stats of the DEFAULT bench
# command (configs[1], the north-star record, the configs[2..4] records), then FETCH_SIZE / WRITE_SIZE in separate --pmc
# passes (MI355X_MICROARCH.md), summarised by tools/prof_bench_summary_r03.py.  Everything lands in gpurun_out/prof_r03/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2> $O/bench_under_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-north-star --no-configs > $O/pmc_$c.log 2>&1
done
cd $R && python tools/prof_bench_summary_r03.py $O
