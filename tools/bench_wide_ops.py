"""Scratch: the float64 operators at 512 x 1024 x 1024 float64 (4.3 GB) + uint8 mask, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import ops, _lib, Gaussian1DKernel, Gaussian2DKernel
from spectral_cube_amd.device import DeviceArray, Event, synchronize
shape = (512, 1024, 1024)
rng = np.random.default_rng(1)
tile = 1000.0 + rng.standard_normal((shape[0], 8, shape[2]))
tm = (rng.random(tile.shape) < 0.8).astype(np.uint8)
cube = DeviceArray.from_numpy(np.tile(tile, (1, shape[1] // 8, 1)))
mask = DeviceArray.from_numpy(np.tile(tm, (1, shape[1] // 8, 1)))
ms = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
out = DeviceArray(shape, np.float64)
def timeit(fn, n=5):
    fn(); synchronize(); ts = []
    for _ in range(n):
        e0, e1 = Event(), Event(); e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))
vox = shape[0] * shape[1] * shape[2]
def rep(name, ms_, bpv):
    print("%-58s %8.3f ms  %6.0f GB/s algorithmic (%d B/voxel)" % (name, ms_, vox * bpv / ms_ / 1e6, bpv), flush=True)
k1, k2 = Gaussian1DKernel(4).array, Gaussian2DKernel(8 / 2.3548).array
rep("statistics() one pass, uint8 mask", timeit(lambda: ops.stats_global_f64(cube, mask=ms)), 9)
rep("sum / max along z, uint8 mask", timeit(lambda: ops.stats_axis_f64(cube, 0, mask=ms, want=("count", "sum", "max"))), 9)
rep("spectral_smooth 33 taps, uint8 mask", timeit(lambda: ops.spectral_conv_f64(cube, k1, mask=ms, out=out)), 17)
rep("spectral_smooth 33 taps, no mask", timeit(lambda: ops.spectral_conv_f64(cube, k1, out=out)), 16)
rep("spatial_smooth 29 x 29 (outer product), uint8 mask", timeit(lambda: ops.spatial_conv_f64(cube, k2, mask=ms, out=out), n=3), 17)
v = np.arange(shape[0]) * 1.0
lo, t, inv, _, _, fill = ops.lerp_plan(v, np.linspace(v[0], v[-1], shape[0]))
rep("spectral_interpolate 512 -> 512 channels", timeit(lambda: ops.spectral_lerp_f64(cube, lo, t, inv, fill, out=out)), 16)
rep("median along z (sorted rays), uint8 mask", timeit(lambda: ops.percentile_axis0_f64(cube, 50.0, mask=ms)), 9)
rep("sigma_clip_spectrally(3) defaults, uint8 mask", timeit(lambda: ops.sigma_clip_axis0_f64(cube, sigma=3.0, mask=ms), n=3), 17)
