# rocprofv3 evidence for bench.py (run on the GPU box through gpurun): kernel trace + stats of the DEFAULT bench command
# (the north-star headline, configs[1], the configs[2..4] records, the SURVEY 8(f) rows, the float64 rows), then FETCH_SIZE /
# WRITE_SIZE in separate --pmc passes OF THE SAME COMMAND (MI355X_MICROARCH.md: one counter per pass, FETCH_SIZE x 2 on
# gfx950), summarised per bench record by tools/prof_bench_summary.py.  Everything lands in gpurun_out/prof_$TAG/ (TAG
# defaults to r06).  The summary names the sha256 of the library the counters were taken on: bench.py quotes them only for
# that library.
# (gpurun MERGES gpurun_out/ back: remove the local gpurun_out/prof_$TAG before a new run, or the summary re-run locally mixes runs)
TAG=${TAG:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --records-file $O/bench_records.json > $O/bench_under_rocprof.log 2> $O/bench_under_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strip-terms --records-file $O/bench_records_pmc_$c.json > $O/pmc_$c.log 2> $O/pmc_$c.err
done
sha256sum $R/spectral_cube_amd/libspcube_hip.so | cut -d' ' -f1 > $O/library_sha256.txt
cd $R && python tools/prof_bench_summary.py $O
