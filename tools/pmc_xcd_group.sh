# HBM read traffic of the clip kernel's one-iteration form with and without the XCD grouping of its tiles (FETCH_SIZE alone in a
# --pmc pass; gfx950: KiB x 2, MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_xcd
mkdir -p $O
cat > /tmp/one_sigma.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests")); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "oracle"))
import numpy as np
from spectral_cube_amd import ops, synth
from spectral_cube_amd.device import DeviceArray, synchronize
from test_gpu_fullsize import _replicate_rows
shape = (1024, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
cube = DeviceArray(shape, np.float32); _replicate_rows(cube, tile, 4)
os.environ["SPC_SIGMA_BT"] = "256"
for _ in range(3):
    ops.sigma_clip_axis0(cube, sigma=3.0, cenfunc="mean", maxiters=1)
synchronize()
PY
for g in 0 16; do
  SPC_XCD_GROUP=$g timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/g$g -- python /tmp/one_sigma.py > $O/g$g.log 2>&1
  f=$(find $O/g$g -name "*counter_collection.csv" | head -1)
  python - "$f" $g <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sigma_clip_reg_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
v = [float(r["Counter_Value"]) for r in rows]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print("SPC_XCD_GROUP=%s: sigma_clip_reg_kernel x%d  FETCH_SIZE %.0f KiB-units -> %.2f GB read (x2 gfx950 correction) for 4.29 GB of cube | %.2f ms under the counter pass" % (
    sys.argv[2], len(v), sum(v) / len(v), sum(v) / len(v) * 1024 * 2 / 1e9, sum(d) / len(d)))
PY
done
