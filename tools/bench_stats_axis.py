"""Scratch: the per-axis statistics kernels at 1024^3 + uint8 mask (kernel times by HIP events, median of 10)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize

shape = (1024, 1024, 1024)
rng = np.random.default_rng(3)
tile = rng.standard_normal((shape[0], 8, shape[2])).astype(np.float32)
tm = (rng.random(tile.shape) < 0.8).astype(np.uint8)
cube = DeviceArray.from_numpy(np.tile(tile, (1, shape[1] // 8, 1)))
mask = DeviceArray.from_numpy(np.tile(tm, (1, shape[1] // 8, 1)))
ms = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)


def timeit(fn, n=10):
    for _ in range(2): fn()
    synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = Event(), Event()
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))


vox = float(np.prod(shape))
for axis in (0, 1, 2):
    out = ops.stats_axis(cube, axis, mask=ms)
    t = timeit(lambda: ops.stats_axis(cube, axis, mask=ms, out=out))
    t0 = timeit(lambda: ops.stats_axis(cube, axis, out=out))
    print("stats_axis %d: uint8 mask %.3f ms (%.2f TB/s = %.3f of 8) | no mask %.3f ms (%.2f TB/s)" % (axis, t, vox * 5 / t / 1e9, vox * 5 / t / 8e9, t0, vox * 4 / t0 / 1e9))
t = timeit(lambda: ops.stats_planes(cube, mask=ms))
print("stats_planes (axis 1, 2): %.3f ms wall" % t)
t = timeit(lambda: ops.stats_global(cube, mask=ms))
print("stats_global: %.3f ms wall" % t)
