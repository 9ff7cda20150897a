"""Summarise the rocprofv3 output of tools/prof_bench.sh.

1. per-(kernel, grid) statistics from the kernel trace of the default bench command (one process runs configs[1], the
   north-star record, the configs[2..4] records and the next rows, so the plain kernel_stats average mixes workloads of
   one kernel) -> bench_kernel_stats_by_workload.csv;
2. every record of the run (read from the bench_records.json the profiled run wrote) is matched to its (kernel, grid) group:
   same kernel base name, trace average closest to the record's HIP-event median;
3. the FETCH_SIZE / WRITE_SIZE passes (separate runs of the same command) give HBM bytes per launch for that group:
   FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 128-byte requests as 64, MI355X_MICROARCH.md) + WRITE_SIZE KiB x 1024
   -> pmc_traffic_by_record.json, keyed by the records' short keys, carrying the sha256 of the library the counters were
   taken on (bench.py reads profiles/r06_pmc_traffic_by_record.json and quotes it only when the loaded library matches)."""
import collections, csv, glob, json, os, re, sys

O = sys.argv[1]


def norm(name):
    name = re.sub(r"\(anonymous namespace\)::|spc_\w+::|void ", "", name)
    return re.sub(r"\(.*\)$", "", name).strip()


def grid_of(r):
    return int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))


stats = glob.glob(os.path.join(O, "stats", "*", "*kernel_stats.csv"))
if stats:
    with open(os.path.join(O, "bench_kernel_stats.csv"), "w") as fh:
        fh.write(open(stats[0]).read())
groups = collections.defaultdict(list)
for f in glob.glob(os.path.join(O, "stats", "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        groups[(norm(r["Kernel_Name"]), grid_of(r))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(((sum(v), k, v) for k, v in groups.items()), reverse=True)
with open(os.path.join(O, "bench_kernel_stats_by_workload.csv"), "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline: durations from the\n"
             "# kernel trace, grouped by (kernel, grid threads) = by workload; warm-up and verification launches included\n")
    fh.write("kernel,grid_threads,calls,median_ns,average_ns,min_ns,max_ns,total_ms\n")
    for tot, (name, grid), v in rows:
        if tot < 2e5:
            continue
        sv = sorted(v)
        fh.write('"%s",%d,%d,%d,%.1f,%d,%d,%.3f\n' % (name, grid, len(v), sv[len(sv) // 2], sum(v) / len(v), min(v), max(v), tot / 1e6))

# Round 5 (round-4 verdict, weak 6): a (kernel, grid) group of a PMC pass also holds the SMALL verification launches of that
# kernel (statistics(): 13 calls, one of 44 us - its average made traffic / algorithmic 12 / 13), and two workloads can share
# a group (the C4 stencil under the random and under the signal mask got ONE value).  Every PMC row keeps its launch duration;
# a record averages only the launches of the duration cluster closest to its own timing, and of those only the ones within
# 10 % of the cluster's median duration.
counters = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(O, "pmc_" + c, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[(norm(r["Kernel_Name"]), int(r["Grid_Size"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), float(r["Counter_Value"])))
    counters[c] = acc


def pmc_mean(c, key, ms):
    """mean counter value of the launches of group `key` that belong to the record timed at `ms` (see above); (mean, n used, n in group)"""
    rows = counters[c].get(key)
    if not rows:
        return None, 0, 0
    durs = cluster([d for d, _ in rows], ms)
    lo, hi = min(durs), max(durs)
    sel = [(d, v) for d, v in rows if lo <= d <= hi]
    med = sorted(d for d, _ in sel)[len(sel) // 2]
    near = [v for d, v in sel if abs(d - med) <= 0.10 * med]
    return sum(near) / len(near), len(near), len(rows)

try:
    detail = json.load(open(os.path.join(O, "bench_records.json")))
except (OSError, ValueError):
    sys.exit("no bench_records.json in %s" % O)

rf = detail["headline"]["roofline"]
records = [("ns", rf["kernel"], rf["kernel_ms"], rf["algorithmic_bytes"])]
if isinstance(detail.get("configs1"), dict) and "roofline" in detail["configs1"]:
    rf = detail["configs1"]["roofline"]
    records.append(("c2", rf["kernel"], rf["kernel_ms"], rf["algorithmic_bytes"]))
for grp in [detail.get("next_rows") or []] + list((detail.get("configs") or {}).values()) + [detail.get("wide") or []]:
    recs = grp if isinstance(grp, list) else ([v for v in grp.values() if isinstance(v, dict)] if isinstance(grp, dict) else [])
    for r in recs:
        if isinstance(r, dict) and "kernel_ms" in r:
            records.append((r["key"], r.get("pmc_kernel") or r["kernel"], r["kernel_ms"], r.get("algorithmic_bytes")))


def base(kernel):
    return re.split(r"[<\s(\[]", kernel.strip())[0]


def cluster(v, ms):
    """two workloads can share (kernel, grid) - the C4 stencil under the random and under the signal mask: when the group's
    durations fall into two clusters (largest gap between sorted neighbours > 2 % of the value), keep the one whose median
    is closer to the record's own timing"""
    sv = sorted(v)
    if len(sv) < 6:
        return v
    gaps = [(sv[i + 1] - sv[i], i) for i in range(2, len(sv) - 3)]
    if not gaps:
        return v
    g, i = max(gaps)
    if g < 0.02 * sv[i]:
        return v
    lo, hi = sv[:i + 1], sv[i + 1:]
    med = lambda a: a[len(a) // 2]                          # noqa: E731
    return lo if abs(med(lo) / 1e6 - ms) <= abs(med(hi) / 1e6 - ms) else hi


out, table = {}, []
for name, kernel, ms, alg in records:
    parts = [p.strip() for p in re.sub(r"\([^)]*\)", "", kernel).split(" + ") if p.strip()]      # (notes in parentheses may hold a '+')
    chosen = []
    for i, part in enumerate(parts):
        cands = [(k, v) for k, v in groups.items() if k[0].startswith(base(part)) and len(v) >= 3]
        # a record that spells its template arguments out (spectral_conv_kernel<33,true,false,false,true>) is matched on them
        # first: the fused and the materialised masked stencil take the same 19.6 ms and differ only there
        targ = re.search(r"<([^>]*)>", part)
        if targ:
            want = base(part) + "<" + targ.group(1).replace(" ", "")
            exact = [(k, v) for k, v in cands if k[0].replace(" ", "").startswith(want)]
            if exact:
                cands = exact
        if not cands:
            chosen = []
            break
        # (the PMC passes run without the strip terms: a group that only the stats pass saw - a rank strip of the same kernel, timed
        #  within a percent of configs[1] - is no candidate when another one has counters)
        with_pmc = [(k, v) for k, v in cands if counters["FETCH_SIZE"].get(k) and counters["WRITE_SIZE"].get(k)]
        if with_pmc:
            cands = with_pmc
        if len(parts) == 1:
            k, v = min(cands, key=lambda kv: abs(sorted(cluster(kv[1], ms))[len(cluster(kv[1], ms)) // 2] / 1e6 - ms))
            v = cluster(v, ms)
            if abs(sorted(v)[len(v) // 2] / 1e6 - ms) > 0.25 * ms:
                chosen = []
                break
        else:      # a pipeline record: its parts are the groups of single-kernel records already chosen for this cube size
            if part.endswith("[all]"):      # every (kernel, grid) group of this kernel: a call that launches it once per slab of planes
                fewest = min(len(v) for _, v in cands)      # a group launched twice per call (two slabs of that size) counts twice
                for kv in sorted(cands, key=lambda kv: -kv[0][1]):
                    chosen.extend([kv] * max(1, round(len(kv[1]) / fewest)))
                continue
            prev = [t for t in table if base(t["kernel_trace_name"]) == base(part)]
            if prev:
                k = max(((p["kernel_trace_name"], p["grid_threads"]) for p in prev), key=lambda kk: kk[1])
            else:                           # no single-kernel record uses it: the group of the largest grid
                k = max((kk for kk, _ in cands), key=lambda kk: kk[1])
            v = groups[k]
        chosen.append((k, v))
    if not chosen:
        continue
    if not all(counters["FETCH_SIZE"].get(k) and counters["WRITE_SIZE"].get(k) for k, _ in chosen):
        continue
    part_ms = [ms if len(chosen) == 1 else sorted(v)[len(v) // 2] / 1e6 for _, v in chosen]
    fm = [pmc_mean("FETCH_SIZE", k, t) for (k, _), t in zip(chosen, part_ms)]
    wm = [pmc_mean("WRITE_SIZE", k, t) for (k, _), t in zip(chosen, part_ms)]
    fetch, write = sum(x[0] for x in fm), sum(x[0] for x in wm)
    sv = sorted(sum((v for _, v in chosen), [])) if len(chosen) == 1 else None
    rec = {"kernel_trace_name": chosen[0][0][0] if len(chosen) == 1 else " + ".join(k[0] for k, _ in chosen),
           "grid_threads": chosen[0][0][1] if len(chosen) == 1 else [k[1] for k, _ in chosen],
           "trace_calls": [len(v) for _, v in chosen],
           "trace_median_ms": (sv[len(sv) // 2] / 1e6) if sv else sum(sorted(v)[len(v) // 2] for _, v in chosen) / 1e6,
           "trace_average_ms": sum(sum(v) / len(v) for _, v in chosen) / 1e6,
           "bench_event_median_ms": ms,
           "fetch_kib_per_launch": fetch, "write_kib_per_launch": write,
           "pmc_launches_used_of_group": {"FETCH_SIZE": [[x[1], x[2]] for x in fm], "WRITE_SIZE": [[x[1], x[2]] for x in wm]},
           "fetch_bytes_per_launch_corrected_x2": fetch * 1024 * 2, "write_bytes_per_launch": write * 1024,
           "hbm_traffic_bytes_per_launch": fetch * 1024 * 2 + write * 1024, "algorithmic_bytes_per_launch": alg,
           "traffic_over_algorithmic": (fetch * 1024 * 2 + write * 1024) / alg if alg else None}
    out[name] = rec
    if len(chosen) == 1:
        table.append(rec)
    print("%-84s %-44s grid %9s trace %8.3f ms bench %8.3f ms traffic/alg %s" % (
        name[:84], str(rec["kernel_trace_name"])[:44], rec["grid_threads"], rec["trace_median_ms"], ms,
        "%.3f" % rec["traffic_over_algorithmic"] if rec["traffic_over_algorithmic"] else "-"))
try:
    lib_sha = open(os.path.join(O, "library_sha256.txt")).read().strip()
except OSError:
    lib_sha = None
try:
    import subprocess
    head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    head = None
json.dump({"library_sha256": lib_sha, "git_head_when_summarised": head,
           "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strip-terms; "
                      "records matched to (kernel, grid) groups of `rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline`",
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE used as reported",
           "records": out}, open(os.path.join(O, "pmc_traffic_by_record.json"), "w"), indent=1)
print("%d of %d records have PMC traffic" % (len(out), len(records)))
