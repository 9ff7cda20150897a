# rocprofv3 evidence for bench.py, round 5 (run on the GPU box through gpurun): kernel trace + stats of the DEFAULT bench
# command (configs[1], the north-star record, the configs[2..4] records, the next rows), then FETCH_SIZE / WRITE_SIZE in
# separate --pmc passes OF THE SAME COMMAND (MI355X_MICROARCH.md: one counter per pass, FETCH_SIZE x 2 on gfx950), summarised
# per bench record by tools/prof_bench_summary_r05.py.  Everything lands in gpurun_out/prof_r05/.
# (gpurun MERGES gpurun_out/ back: remove the local gpurun_out/prof_r05 before a new run, or the summary re-run locally mixes runs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.log 2> $O/bench_under_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2> $O/pmc_$c.err
done
cd $R && python tools/prof_bench_summary_r05.py $O
