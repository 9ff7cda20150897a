"""Scratch: is the one-kernel sigma clip (and the median) bit-reproducible from launch to launch?  The same 1024^3 input
(the full-size test's tile, replicated along y) is clipped N times; every result is reduced on the device to
(count, min, max, sum, sumsq) and, per y period, compared against period 0 through the sum map along z - any
difference is a scheduling dependence."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from test_gpu_fullsize import _replicate_rows

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
shape, ty = (1024, 1024, 1024), 8
rng = np.random.default_rng(77)
tile = rng.standard_normal((shape[0], ty, shape[2])).astype(np.float32)
tile[rng.random(tile.shape) < 0.02] *= 15.0
tile[:, 2, 16:24] = np.nan
tmask = rng.random(tile.shape) < 0.9
cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
_replicate_rows(cube, tile, 4)
_replicate_rows(mask, tmask.astype(np.uint8), 1)
ms = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
ref = None
bad = 0
for it in range(N):
    out = ops.sigma_clip_axis0(cube, sigma=3.0, mask=ms)
    st = ops.stats_global(out)
    sm = ops.stats_axis(out, 0, want=("count", "sum"))
    cnt, s = sm["count"].get(), sm["sum"].get()
    per = [(np.array_equal(cnt[k * ty:(k + 1) * ty], cnt[:ty]) and np.array_equal(s[k * ty:(k + 1) * ty], s[:ty], equal_nan=True)) for k in range(shape[1] // ty)]
    key = (st["npts"], st["sum"], st["sumsq"], st["min"], st["max"])
    if ref is None:
        ref = key
        print("reference record", key, "periods equal", sum(per), "of", len(per), flush=True)
    ok = key == ref and all(per)
    if not ok:
        bad += 1
        print("run %d: differs: stats equal %s, periods unlike period 0: %s" % (it, key == ref, [k for k, p in enumerate(per) if not p][:8]), flush=True)
    del out
med_ref = None
for it in range(N):
    med = ops.percentile_axis0(cube, 50.0, mask=ms).get()
    per = all(np.array_equal(med[k * ty:(k + 1) * ty], med[:ty], equal_nan=True) for k in range(shape[1] // ty))
    if med_ref is None:
        med_ref = med
    if not (per and np.array_equal(med, med_ref, equal_nan=True)):
        bad += 1
        print("median run %d differs" % it, flush=True)
print("%d sigma clips + %d medians of the same input: %d differing results" % (N, N, bad))
