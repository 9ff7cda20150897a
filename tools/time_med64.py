"""scratch: float64 median along z at 512 (or argv[1]) x 1024 x 1024 + uint8 mask, per form of the kernel (SPC_SELECT64 = 0 | 1 | 2)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (nz, 1024, 1024)
rng = np.random.default_rng(1)
tile = 1000.0 + rng.standard_normal((shape[0], 8, shape[2]))
tm = (rng.random(tile.shape) < 0.8).astype(np.uint8)
cube = DeviceArray.from_numpy(np.tile(tile, (1, shape[1] // 8, 1)))
mask = DeviceArray.from_numpy(np.tile(tm, (1, shape[1] // 8, 1)))
ms = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
def timeit(fn, n=5):
    fn(); synchronize(); ts = []
    for _ in range(n):
        e0, e1 = Event(), Event(); e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))
t = timeit(lambda: ops.percentile_axis0_f64(cube, 50.0, mask=ms))
print("SPC_SELECT64=%s nz=%d: median f64 %.3f ms = %.0f GB/s algorithmic" % (os.environ.get("SPC_SELECT64", "default"), nz, t, nz * 1024 * 1024 * 9 / t / 1e6))
if nz <= 1024:
    keep = {}
    def clip():
        keep["r"] = None
        keep["r"] = ops.sigma_clip_axis0_f64(cube, sigma=3.0, mask=ms)
    t = timeit(clip, n=3)
    print("SPC_SELECT64=%s nz=%d: sigma_clip f64 %.3f ms = %.0f GB/s algorithmic" % (os.environ.get("SPC_SELECT64", "default"), nz, t, nz * 1024 * 1024 * 17 / t / 1e6))
