"""Scratch: 29-tap separable spatial stencil timing only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize
shape = tuple(int(s) for s in (sys.argv[1:4] or (512, 2048, 2048)))
nz, ny, nx = shape
rng = np.random.default_rng(0)
cube = DeviceArray(shape, np.float32); maskc = DeviceArray(shape, np.uint8)
plane = rng.standard_normal((ny, nx)).astype(np.float32); mp = (rng.random((ny, nx)) > 0.2).astype(np.uint8)
for z in range(nz):
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(cube.ptr + z * plane.nbytes), (plane + np.float32(z % 7)).ctypes.data_as(C.c_void_p), plane.nbytes, None)
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(maskc.ptr + z * mp.nbytes), mp.ctypes.data_as(C.c_void_p), mp.nbytes, None)
out = DeviceArray(shape, np.float32)
g = np.exp(-0.5 * (np.arange(-14, 15) / 3.397) ** 2); g /= g.sum(); k2 = np.outer(g, g)
def timeit(fn, n=5):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
vox = nz * ny * nx
ms = timeit(lambda: ops.spatial_conv(cube, k2, out=out))
print("%s all-valid %8.3f ms  %.1f%% of 8TB/s" % (os.environ.get("SPC_HIP_LIBRARY", "default"), ms, vox * 8 / ms / 1e6 / 80), flush=True)
if "--mask" in sys.argv:
    ms = timeit(lambda: ops.spatial_conv(cube, k2, out=out, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc)))
    print("   u8 mask %8.3f ms  %.1f%% of 8TB/s" % (ms, vox * 9 / ms / 1e6 / 80), flush=True)
