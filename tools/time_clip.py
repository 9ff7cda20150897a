"""sigma_clip_spectrally at 1024^3 + uint8 mask (80 % valid, random) by iteration count, with and without the sorted neighbourhood of
the first median (SPC_CLIP_NEAR, read per call): where the clip kernel's time goes"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize
shape = (1024, 1024, 1024)
rng = np.random.default_rng(3)
tile = (rng.standard_normal((1024, 8, 1024)) + 2.0).astype(np.float32)
tm = (rng.random(tile.shape) < 0.8).astype(np.uint8)
cube = DeviceArray.from_numpy(np.tile(tile, (1, 128, 1)))
mask = DeviceArray.from_numpy(np.tile(tm, (1, 128, 1)))
ms = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
def timeit(fn, n=3):
    fn(); synchronize(); ts = []
    for _ in range(n):
        e0, e1 = Event(), Event(); e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))
keep = {}
for cen in ("median", "mean"):
    for near in ("1", "0"):
        os.environ["SPC_CLIP_NEAR"] = near
        row = []
        for it in (1, 2, 3, 5, None):
            def run():
                keep["r"] = None
                keep["r"] = ops.sigma_clip_axis0(cube, sigma=3.0, maxiters=it, cenfunc=cen, mask=ms)
            row.append("%s: %.2f" % (it, timeit(run)))
        print("cenfunc=%s SPC_CLIP_NEAR=%s  maxiters " % (cen, near) + "  ".join(row) + "  ms", flush=True)
