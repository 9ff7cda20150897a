"""float64 spectral_smooth at 512 (or argv[1]) x 1024 x 1024 + uint8 mask: the ring form against the runs of 16 (SPC_SPECTRAL64_RING
= 1 | 0, read per call), timed and compared bit for bit"""
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


def gauss(n, sigma):
    x = np.arange(n) - n // 2
    g = np.exp(-0.5 * (x / sigma) ** 2)
    return g / g.sum()


def timeit(fn, n=5):
    fn(); synchronize(); ts = []
    for _ in range(n):
        e0, e1 = Event(), Event(); e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))


out = DeviceArray(shape, np.float64)
for taps, sigma in ((33, 4.0), (17, 2.0), (9, 1.0)):
    k1 = gauss(taps, sigma)
    got = {}
    for form in ("1", "0"):
        os.environ["SPC_SPECTRAL64_RING"] = form
        t = timeit(lambda: ops.spectral_conv_f64(cube, k1, mask=ms, out=out))
        print("SPC_SPECTRAL64_RING=%s nz=%d taps=%d: %.3f ms = %.0f GB/s algorithmic (17 B/voxel)" % (form, nz, taps, t, nz * 1024 * 1024 * 17 / t / 1e6), flush=True)
        got[form] = np.concatenate([out.planes(z, z + 1).get()[0][:24] for z in (0, 1, 15, 16, 17, nz // 2, nz - 18, nz - 2, nz - 1)])
    a, b = got["1"], got["0"]
    print("taps %d: ring == runs of 16 bit for bit: %s (NaN patterns equal: %s, max rel diff %.3g)" % (
        taps, np.array_equal(a, b, equal_nan=True), np.array_equal(np.isnan(a), np.isnan(b)), np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))))
