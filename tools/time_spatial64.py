"""float64 spatial_smooth at 512 (or argv[1]) x 1024 x 1024 + uint8 mask: the one-kernel ring form against the two-pass forms
(SPC_SPATIAL64_RING is read once per process: the script re-runs itself with SPC_SPATIAL64_RING=0 for the two-pass time and for
the samples both forms are compared on, bit for bit)"""
import sys, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize

nz = int(sys.argv[1]) if len(sys.argv) > 1 else 512
child = len(sys.argv) > 2 and sys.argv[2] == "child"
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
tag = "SPC_SPATIAL64_RING=%s" % os.environ.get("SPC_SPATIAL64_RING", "default")
samples = {}
for taps, sigma in ((29, 8 / 2.3548200450309493), (15, 2.0), (33, 4.0)):
    k2 = np.outer(gauss(taps, sigma), gauss(taps, sigma))
    t = timeit(lambda: ops.spatial_conv_f64(cube, k2, mask=ms, out=out))
    print("%s nz=%d taps=%d: %.3f ms = %.0f GB/s algorithmic (17 B/voxel)" % (tag, nz, taps, t, nz * 1024 * 1024 * 17 / t / 1e6), flush=True)
    samples["t%d" % taps] = np.concatenate([out.planes(z, z + 1).get()[0][rows] for z in (0, nz - 1) for rows in (slice(0, 40), slice(500, 540), slice(-40, None))])
np.savez("/tmp/spatial64_%s.npz" % ("two_pass" if child else "ring"), **samples)
if not child:
    env = dict(os.environ, SPC_SPATIAL64_RING="0")
    subprocess.run([sys.executable, os.path.abspath(__file__), str(nz), "child"], env=env, check=True)
    a, b = np.load("/tmp/spatial64_ring.npz"), np.load("/tmp/spatial64_two_pass.npz")
    for key in a.files:
        same = np.array_equal(a[key], b[key], equal_nan=True)
        diff = np.nanmax(np.abs(a[key] - b[key]) / np.maximum(np.abs(b[key]), 1e-300))
        print("%s: ring == two-pass bit for bit: %s (max rel diff %.3g, NaN patterns equal: %s)" % (key, same, diff, np.array_equal(np.isnan(a[key]), np.isnan(b[key]))))
