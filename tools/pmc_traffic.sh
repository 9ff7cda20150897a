# HBM traffic of one op: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (together they exceed the counter
# hardware); gfx950: FETCH_SIZE counts 64-byte units at half rate (x2), see MI355X_MICROARCH.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_tr
for c in FETCH_SIZE WRITE_SIZE; do
  REPS=2 timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tr/${OP}_$c -- python $R/tools/prof_one.py $OP $SHAPE > $R/gpurun_out/pmc_tr/${OP}_$c.log 2>&1
done
cd $R && python tools/pmc_summary.py "gpurun_out/pmc_tr/${OP}_*/*/*counter_collection.csv" "$KERN"
