"""Scratch: run ONE op a few times (for rocprofv3 counter passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, synchronize
op = sys.argv[1]
shape = tuple(int(s) for s in sys.argv[2:5]) if len(sys.argv) > 4 else (1024, 1024, 1024)
reps = int(os.environ.get("REPS", 3))
nz, ny, nx = shape
rng = np.random.default_rng(0)
cube = DeviceArray(shape, np.float32)
for z in range(nz):
    plane = rng.standard_normal((ny, nx), dtype=np.float32)
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(cube.ptr + z * plane.nbytes), plane.ctypes.data_as(C.c_void_p), plane.nbytes, None)
cen = DeviceArray.from_numpy((np.arange(nz) - nz // 2) * 500.0)
g = np.exp(-0.5 * (np.arange(-16, 17) / 4.0) ** 2); g /= g.sum()
g29 = np.exp(-0.5 * (np.arange(-14, 15) / 3.397) ** 2); g29 /= g29.sum()
out = DeviceArray(shape, np.float32)
for _ in range(reps):
    if op == "sconv": ops.spectral_conv(cube, g, out=out)
    elif op == "sconv_fused": ops.spectral_conv_moments(cube, g, cen)
    elif op == "spconv": ops.spatial_conv(cube, np.outer(g29, g29), out=out)
    elif op == "moments": ops.moments(cube, cen)
    elif op == "bilinear":
        yy, xx = np.mgrid[0:ny, 0:nx].astype(np.float64)
        a = np.deg2rad(30.0)
        xs = np.cos(a) * (xx - nx / 2) - np.sin(a) * (yy - ny / 2) + nx / 2
        ys = np.sin(a) * (xx - nx / 2) + np.cos(a) * (yy - ny / 2) + ny / 2
        ops.resample_bilinear(cube, xs, ys)
    elif op == "stats": ops.stats_global(cube)
    elif op == "median": ops.percentile_axis0(cube, 50.0)
    elif op == "sconv_mask":
        if _ == 0:
            mp = (rng.random((ny, nx)) > 0.3).astype(np.uint8)
            maskc = DeviceArray(shape, np.uint8)
            for z in range(nz):
                _lib.call("spc_memcpy_h2d", 0, C.c_void_p(maskc.ptr + z * mp.nbytes), np.roll(mp, z).ctypes.data_as(C.c_void_p), mp.nbytes, None)
        ops.spectral_conv(cube, g, out=out, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc))
    elif op == "spconv_mask":
        if _ == 0:
            mp = (rng.random((ny, nx)) > 0.2).astype(np.uint8)
            maskc = DeviceArray(shape, np.uint8)
            for z in range(nz):
                _lib.call("spc_memcpy_h2d", 0, C.c_void_p(maskc.ptr + z * mp.nbytes), mp.ctypes.data_as(C.c_void_p), mp.nbytes, None)
        ops.spatial_conv(cube, np.outer(g29, g29), out=out, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc))
    elif op in ("spmfma_store", "spmfma_mom"):
        if _ == 0:
            mp = (rng.random((ny, nx)) > 0.2).astype(np.uint8)
            maskc = DeviceArray(shape, np.uint8)
            for z in range(nz):
                _lib.call("spc_memcpy_h2d", 0, C.c_void_p(maskc.ptr + z * mp.nbytes), mp.ctypes.data_as(C.c_void_p), mp.nbytes, None)
            m0 = DeviceArray((ny, nx), np.float64)
        if op == "spmfma_store":
            ops.spatial_conv_mfma(cube, np.outer(g29, g29), mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc), out=out)
        else:
            ops.spatial_conv_mfma(cube, np.outer(g29, g29), mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc), want_cube=False, want_m0=True, m0=m0)
synchronize()
print("done", op)
