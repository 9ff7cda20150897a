"""Scratch/report: the float64 moment kernels (spc_moments_f64 / spc_moment_order_f64) at 512 x 1024 x 1024 float64 (4.3 GB):
kernel times by events, GB/s against the 8 bytes per voxel each pass reads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize
nz, ny, nx = (int(v) for v in (sys.argv[1:4] or (512, 1024, 1024)))
rng = np.random.default_rng(0)
tile = 1000.0 + rng.standard_normal((nz, 8, nx))
cube = DeviceArray.from_numpy(np.tile(tile, (1, ny // 8, 1)), 0)
mask = DeviceArray.from_numpy(np.tile((rng.random((nz, 8, nx)) < 0.8).astype(np.uint8), (1, ny // 8, 1)), 0)
cen = DeviceArray.from_numpy(np.arange(nz, dtype=np.float64) - nz // 2, 0)


def timeit(fn, n=5):
    fn(); synchronize(); ts = []
    for _ in range(n):
        e0, e1 = Event(), Event(); e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))


gb = nz * ny * nx * 8 / 1e9
for label, spec in (("isfinite", ops.MaskSpec(_lib.MASK_FINITE)), ("isfinite & uint8 array", ops.MaskSpec(_lib.MASK_FINITE | _lib.MASK_ARRAY, array=mask))):
    extra = nz * ny * nx / 1e9 if spec.array is not None else 0.0
    t1 = timeit(lambda: ops.moments_f64(cube, cen, mask=spec, want=("m0", "m1")))
    t2 = timeit(lambda: ops.moments_f64(cube, cen, mask=spec, want=("m0", "m1", "m2")))
    t3 = timeit(lambda: ops.moments_f64(cube, cen, mask=spec, want=("m0", "argmax", "vmax")))
    print("%-24s m0 + m1 (one pass) %.3f ms = %.2f TB/s | m0 + m1 + m2 (two passes) %.3f ms = %.2f TB/s | m0 + argmax + max %.3f ms" % (
        label, t1, (gb + extra) / t1, t2, 2 * (gb + extra) / t2, t3), flush=True)
