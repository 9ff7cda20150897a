# rocprofv3 kernel stats of the order-statistic kernels at 1024^3 (tools/bench_select.py), round 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_sel
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tools/bench_select.py > $O/bench_select_under_rocprof.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python - "$f" > $O/r03_order_statistics_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Name"] for k in ("select", "sigma_clip", "clip_", "stats_axis", "fill_masked"))]
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
for r in keep:
    w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]])
PY
cat $O/r03_order_statistics_kernel_stats.csv | cut -c1-200
