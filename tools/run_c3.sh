mkdir -p gpurun_out/r3d
python -m pytest tests -m gpu -x -q -k "spectral or smooth or fused or conv" 2>&1 | tail -4
python bench.py --configs-only C3 --no-north-star --no-cpu-baseline --steps 3 --warmup 1 2> gpurun_out/r3d/c3.err | tail -1 > gpurun_out/r3d/c3.json
python - <<PY
import json
d=json.load(open("gpurun_out/r3d/c3.json"))
for r in d["configs"]["C3"]: print("%-70s %8.3f ms frac %.3f err %s" % (r["name"], r["kernel_ms"], r["frac"], r["verify"].get("max_scaled_err")))
PY
