"""Scratch: non-separable 2-D stencil (rotated elliptical Gaussians: the convolve_to kernels) on 256 x 2048^2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize
shape = (256, 2048, 2048)
nz, ny, nx = shape
rng = np.random.default_rng(0)
cube = DeviceArray(shape, np.float32)
plane = rng.standard_normal((ny, nx)).astype(np.float32)
for z in range(nz):
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(cube.ptr + z * plane.nbytes), plane.ctypes.data_as(C.c_void_p), plane.nbytes, None)
out = DeviceArray(shape, np.float32)
def timeit(fn, n=3):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
for kk in (9, 13, 25, 41):
    yy, xx = np.mgrid[-(kk // 2):kk // 2 + 1, -(kk // 2):kk // 2 + 1]
    s = kk / 8.0
    kn = np.exp(-0.5 * (((xx + 0.5 * yy) / s) ** 2 + (yy / (0.6 * s)) ** 2))
    ms = timeit(lambda: ops.spatial_conv(cube, kn, out=out))
    print("%2d x %2d non-separable: %8.3f ms  %6.1f GFMA/s-equivalent %.1f%% of 8 TB/s" % (kk, kk, ms, nz * ny * nx * kk * kk / ms / 1e6, nz * ny * nx * 8 / ms / 1e6 / 80), flush=True)
