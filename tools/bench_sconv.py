"""Scratch: timings of the spectral stencil variants (python tools/bench_sconv.py [nz ny nx])."""
import os, sys
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
for z in range(nz):
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(cube.ptr + z * plane.nbytes), (plane + np.float32(z % 7)).ctypes.data_as(C.c_void_p), plane.nbytes, None)
maskc = DeviceArray(shape, np.uint8)
mp = (rng.random((ny, nx)) > 0.3).astype(np.uint8)
for z in range(nz):
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(maskc.ptr + z * mp.nbytes), np.roll(mp, z).ctypes.data_as(C.c_void_p), mp.nbytes, None)
# sparse NaNs (what blanked pixels of real cubes look like to the general kernel): density 1e-4 per voxel
nanc = DeviceArray(shape, np.float32)
npl = plane.copy()
npl[rng.random((ny, nx)) < 1e-4] = np.nan
for z in range(nz):
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(nanc.ptr + z * plane.nbytes), (np.roll(npl, 37 * z) + np.float32(z % 7)).ctypes.data_as(C.c_void_p), plane.nbytes, None)
vox = nz * ny * nx
out = DeviceArray(shape, np.float32)
cen_h = (np.arange(nz) - nz // 2) * 500.0
cen = DeviceArray.from_numpy(cen_h)


def report(name, ms, bpv):
    print("%-52s %8.3f ms  %9.1f Mvox/s  %5.1f%% of 8TB/s" % (name, ms, vox / ms / 1e3, vox * bpv / ms / 1e6 / 80), flush=True)


for taps, sig in ((33, 4.0), (17, 2.0), (9, 1.0)):
    h = taps // 2
    g = np.exp(-0.5 * (np.arange(-h, h + 1) / sig) ** 2); g /= g.sum()
    for env in ({"SPC_CONV_VEC": "1"}, {"SPC_CONV_VEC": "2"}, {"SPC_CONV_FAST": "0"}):
        for k in ("SPC_CONV_VEC", "SPC_CONV_FAST"): os.environ.pop(k, None)
        os.environ.update(env)
        report("sconv %d taps all-valid %s" % (taps, env), timeit(lambda: ops.spectral_conv(cube, g, out=out)), 8)
    for k in ("SPC_CONV_VEC", "SPC_CONV_FAST"): os.environ.pop(k, None)
    report("sconv %d taps u8 mask (general)" % taps, timeit(lambda: ops.spectral_conv(cube, g, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc), out=out)), 9)
    report("sconv %d taps NaNs at 1e-4 (fast pass + general)" % taps, timeit(lambda: ops.spectral_conv(nanc, g, out=out)), 8)
    report("sconv %d taps fused moments (algebraic)" % taps, timeit(lambda: ops.spectral_conv_moments(cube, g, cen, cen_host=cen_h, want=("m0", "m1", "m2"))), 4)
    os.environ["SPC_FUSE_ALGEBRAIC"] = "0"
    report("sconv %d taps fused moments (stencil)" % taps, timeit(lambda: ops.spectral_conv_moments(cube, g, cen, cen_host=cen_h, want=("m0", "m1", "m2"))), 4)
    os.environ.pop("SPC_FUSE_ALGEBRAIC")
    report("sconv %d taps fused moments+argmax" % taps, timeit(lambda: ops.spectral_conv_moments(cube, g, cen, cen_host=cen_h, want=("m0", "m1", "m2", "argmax"))), 4)
    report("sconv %d taps fused moments u8 mask" % taps, timeit(lambda: ops.spectral_conv_moments(cube, g, cen, cen_host=cen_h, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc), want=("m0", "m1", "m2"))), 5)
for taps in (41, 49, 65):
    h = taps // 2
    g = np.exp(-0.5 * (np.arange(-h, h + 1) / (taps / 8.0)) ** 2)
    report("sconv %d taps all-valid (ring pass)" % taps, timeit(lambda: ops.spectral_conv(cube, g, out=out), n=3), 8)
    os.environ["SPC_CONV_FAST"] = "0"
    report("sconv %d taps all-valid (runs-of-16 kernel)" % taps, timeit(lambda: ops.spectral_conv(cube, g, out=out), n=2), 8)
    os.environ.pop("SPC_CONV_FAST")
    report("sconv %d taps NaNs at 1e-4 (ring pass + runs of 16)" % taps, timeit(lambda: ops.spectral_conv(nanc, g, out=out), n=2), 8)
g = np.exp(-0.5 * (np.arange(-40, 41) / 10.0) ** 2)
report("sconv 81 taps (runs-of-16 kernel)", timeit(lambda: ops.spectral_conv(cube, g, out=out), n=2), 8)
