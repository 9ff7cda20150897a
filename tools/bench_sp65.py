"""Scratch: 65-tap separable spatial stencil, all-valid and masked (python tools/bench_sp65.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize
shape = (256, 2048, 2048)
nz, ny, nx = shape
rng = np.random.default_rng(0)
cube = DeviceArray(shape, np.float32); maskc = DeviceArray(shape, np.uint8)
plane = rng.standard_normal((ny, nx)).astype(np.float32); mp = (rng.random((ny, nx)) > 0.2).astype(np.uint8)
for z in range(nz):
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(cube.ptr + z * plane.nbytes), plane.ctypes.data_as(C.c_void_p), plane.nbytes, None)
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(maskc.ptr + z * mp.nbytes), mp.ctypes.data_as(C.c_void_p), mp.nbytes, None)
out = DeviceArray(shape, np.float32)
def timeit(fn, n=3):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
for taps in (41, 65):
    h = taps // 2
    g = np.exp(-0.5 * (np.arange(-h, h + 1) / (taps / 8.0)) ** 2); g /= g.sum(); k2 = np.outer(g, g)
    print("%d taps all-valid %.3f ms | u8 mask %.3f ms" % (taps, timeit(lambda: ops.spatial_conv(cube, k2, out=out)),
          timeit(lambda: ops.spatial_conv(cube, k2, out=out, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc)))), flush=True)
