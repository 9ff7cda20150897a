"""Summarise the rocprofv3 output of tools/prof_bench_r03.sh: per-(kernel, grid) statistics from the kernel trace of the
default bench command (one process runs configs[1], the north-star record and the configs[2..4] records, so the plain
kernel_stats average mixes workloads of the same kernel), and the PMC traffic JSON bench.py reads."""
import collections, csv, glob, json, os, re, sys
O = sys.argv[1]
stats = glob.glob(os.path.join(O, "stats", "*", "*kernel_stats.csv"))
if stats:
    with open(os.path.join(O, "bench_kernel_stats.csv"), "w") as fh:
        fh.write(open(stats[0]).read())
trace = glob.glob(os.path.join(O, "stats", "*", "*kernel_trace.csv"))
if trace:
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(trace[0])):
        name = re.sub(r"\(anonymous namespace\)::|spc_\w+::|void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*\)$", "", name)
        grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
        groups[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = sorted(((sum(v), k, v) for k, v in groups.items()), reverse=True)
    with open(os.path.join(O, "bench_kernel_stats_by_workload.csv"), "w") as fh:
        fh.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline (round 3): durations from the\n"
                 "# kernel trace, grouped by (kernel, grid threads) = by workload; warm-up and verification launches included\n")
        fh.write("kernel,grid_threads,calls,average_ns,min_ns,max_ns,total_ms\n")
        for tot, (name, grid), v in rows:
            if tot < 2e5:
                continue
            fh.write('"%s",%d,%d,%.1f,%d,%d,%.3f\n' % (name, grid, len(v), sum(v) / len(v), min(v), max(v), tot / 1e6))
    for tot, (name, grid), v in rows[:14]:
        print("%-80s grid %9d calls %3d avg %10.1f us" % (name[:80], grid, len(v), sum(v) / len(v) / 1e3))
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(O, "pmc_" + c, "*", "*counter_collection.csv"))
    vals, meta = [], {}
    for f in files:
        for r in csv.DictReader(open(f)):
            if "moments_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c and int(r["Grid_Size"]) == 1048576:
                vals.append(float(r["Counter_Value"]))
                meta = {"vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"], "lds": r["LDS_Block_Size"], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"]}
    if vals:
        raw[c] = dict(launches=len(vals), mean_kb=sum(vals) / len(vals), min_kb=min(vals), max_kb=max(vals), **meta)
if len(raw) == 2:
    fetch = raw["FETCH_SIZE"]["mean_kb"] * 1024 * 2
    write = raw["WRITE_SIZE"]["mean_kb"] * 1024
    out = {"kernel": "moments_kernel<4,4,8,true,false,true>", "workload": "1024x1024x1024 fp32 + uint8 mask, moment0+1+2",
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-north-star --no-configs",
           "raw": raw, "fetch_bytes_per_launch_corrected_x2": fetch, "write_bytes_per_launch": write,
           "hbm_traffic_bytes_per_launch": fetch + write, "algorithmic_bytes_per_launch": 1024 ** 3 * 5 + 1024 ** 2 * 24,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE used as reported"}
    json.dump(out, open(os.path.join(O, "moments_c2_pmc.json"), "w"), indent=1)
    print("traffic / algorithmic = %.5f" % (out["hbm_traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]))
