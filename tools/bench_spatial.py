"""Scratch probe: spatial stencil timings (all-valid fast pass / general masked kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize

def timeit(fn, n=5, warm=1):
    for _ in range(warm): fn()
    synchronize()
    e0, e1 = Event(), Event()
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_ms(e1) / n

shape = tuple(int(s) for s in (sys.argv[1:4] or (1024, 1024, 1024)))
nz, ny, nx = shape
rng = np.random.default_rng(0)
plane = rng.standard_normal((ny, nx)).astype(np.float32)
cube = DeviceArray(shape, np.float32)
maskc = DeviceArray(shape, np.uint8)
mp = (rng.random((ny, nx)) > 0.2).astype(np.uint8)
for z in range(nz):
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(cube.ptr + z * plane.nbytes), (plane + np.float32(z % 7)).ctypes.data_as(C.c_void_p), plane.nbytes, None)
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(maskc.ptr + z * mp.nbytes), mp.ctypes.data_as(C.c_void_p), mp.nbytes, None)
vox = nz * ny * nx
out = DeviceArray(shape, np.float32)
for sig, half in ((3.397, 14), (1.0, 4), (2.0, 8), (4.0, 16)):
    g = np.exp(-0.5 * (np.arange(-half, half + 1) / sig) ** 2); g /= g.sum()
    k2 = np.outer(g, g)
    ms = timeit(lambda: ops.spatial_conv(cube, k2, out=out))
    print("spatial %2d taps all-valid   %8.3f ms %7.1f GB/s" % (2 * half + 1, ms, vox * 8 / ms / 1e6), flush=True)
    ms = timeit(lambda: ops.spatial_conv(cube, k2, out=out, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc)))
    print("spatial %2d taps u8 mask     %8.3f ms %7.1f GB/s" % (2 * half + 1, ms, vox * 9 / ms / 1e6), flush=True)
    ms = timeit(lambda: ops.spatial_conv(cube, k2, out=out, mask=ops.MaskSpec(_lib.MASK_GT, 0.5)))
    print("spatial %2d taps GT mask     %8.3f ms %7.1f GB/s" % (2 * half + 1, ms, vox * 8 / ms / 1e6), flush=True)
if "--nonsep" in sys.argv:
    yy, xx = np.mgrid[-7:8, -7:8]
    th = np.deg2rad(30.0)
    u, v = xx * np.cos(th) + yy * np.sin(th), -xx * np.sin(th) + yy * np.cos(th)
    k = np.exp(-0.5 * ((u / 3.0) ** 2 + (v / 1.5) ** 2)); k /= k.sum()       # rotated elliptical Gaussian (convolve_to-like)
    small = DeviceArray((64, ny, nx), np.float32)
    _lib.call("spc_memcpy_d2d", 0, C.c_void_p(small.ptr), C.c_void_p(cube.ptr), small.nbytes, None)
    so = DeviceArray(small.shape, np.float32)
    ms = timeit(lambda: ops.spatial_conv(small, k, out=so), n=2, warm=1)
    print("non-separable 15x15, 64 planes of %dx%d: %8.3f ms  -> %.1f ms per 1024 planes" % (ny, nx, ms, ms * 16))
    sm_ = DeviceArray((64, ny, nx), np.uint8)
    _lib.call("spc_memcpy_d2d", 0, C.c_void_p(sm_.ptr), C.c_void_p(maskc.ptr), sm_.nbytes, None)
    ms = timeit(lambda: ops.spatial_conv(small, k, out=so, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=sm_)), n=2, warm=1)
    print("non-separable 15x15 + u8 mask:            %8.3f ms  -> %.1f ms per 1024 planes" % (ms, ms * 16))
if "--wide" in sys.argv:
    for nt in (41, 65, 81):
        g = np.exp(-0.5 * (np.arange(-(nt // 2), nt // 2 + 1) / (nt / 8.0)) ** 2); g /= g.sum()
        ms = timeit(lambda: ops.spectral_conv(cube, g, out=out), n=2, warm=1)
        print("spectral %2d taps (generic kernel) %8.3f ms %7.1f GB/s" % (nt, ms, vox * 8 / ms / 1e6), flush=True)
    for nt in (41, 65):
        g = np.exp(-0.5 * (np.arange(-(nt // 2), nt // 2 + 1) / (nt / 8.0)) ** 2); g /= g.sum()
        ms = timeit(lambda: ops.spatial_conv(cube, np.outer(g, g), out=out), n=2, warm=1)
        print("spatial %2d taps separable      %8.3f ms %7.1f GB/s" % (nt, ms, vox * 8 / ms / 1e6), flush=True)
if "--vwide" in sys.argv:
    g = np.exp(-0.5 * (np.arange(-40, 41) / 10.0) ** 2); g /= g.sum()
    small = DeviceArray((128, ny, nx), np.float32)
    _lib.call("spc_memcpy_d2d", 0, C.c_void_p(small.ptr), C.c_void_p(cube.ptr), small.nbytes, None)
    so = DeviceArray(small.shape, np.float32)
    ms = timeit(lambda: ops.spatial_conv(small, np.outer(g, g), out=so), n=2, warm=1)
    print("spatial 81 taps separable, 128 planes: %8.3f ms -> %.1f ms per 1024 planes" % (ms, ms * 8), flush=True)
