# HBM traffic of the stencil / stream kernels, FETCH_SIZE and WRITE_SIZE in separate --pmc passes (gfx950: FETCH_SIZE KiB x 2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_all
mkdir -p $O
for spec in "spconv 256 2048 2048 spatial_sep" "sconv 1024 1024 1024 spectral_conv" "sconv_mask 1024 1024 1024 spectral_conv" "spconv_mask 256 2048 2048 spatial_sep" "median 1024 1024 1024 select_reg"; do
  set -- $spec
  for c in FETCH_SIZE WRITE_SIZE; do
    REPS=2 timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$1_$c -- python $R/tools/prof_one.py $1 $2 $3 $4 > $O/$1_$c.log 2>&1
  done
  python - "$O" $1 $2 $3 $4 $5 <<'PY'
import csv, glob, sys
O, op, nz, ny, nx, kern = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
vox = nz * ny * nx
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("%s/%s_%s/*/*counter_collection.csv" % (O, op, c)):
        rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == c]
        if rows:
            by = {}
            for r in rows:
                by.setdefault(r["Kernel_Name"][:70], []).append(float(r["Counter_Value"]))
            res[c] = {k: sum(v) / len(v) for k, v in by.items()}
for k in res.get("FETCH_SIZE", {}):
    rd = res["FETCH_SIZE"][k] * 1024 * 2 / 1e9
    wr = res.get("WRITE_SIZE", {}).get(k, float("nan")) * 1024 / 1e9
    print("%-12s %-72s read %.2f GB (x%.2f of the cube)  written %.2f GB (x%.2f)" % (op, k, rd, rd / (vox * 4 / 1e9), wr, wr / (vox * 4 / 1e9)))
PY
done
