"""Scratch throughput probe (not the bench contract): device-resident timings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize

def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    synchronize()
    e0, e1 = Event(), Event()
    e0.record(); 
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_ms(e1) / n

shape = tuple(int(s) for s in (sys.argv[1:4] or (1024, 1024, 1024)))
nz, ny, nx = shape
print("shape", shape, flush=True)
rng = np.random.default_rng(0)
plane = rng.standard_normal((ny, nx)).astype(np.float32)
cube = DeviceArray(shape, np.float32)
import ctypes as C
for z in range(nz):   # fill plane by plane (cheap host side)
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(cube.ptr + z * plane.nbytes), (plane + np.float32(z % 7)).ctypes.data_as(C.c_void_p), plane.nbytes, None)
mask = DeviceArray.from_numpy((rng.random((ny, nx)) > 0.3).astype(np.uint8).repeat(1))
maskc = DeviceArray(shape, np.uint8)
mp = (rng.random((ny, nx)) > 0.3).astype(np.uint8)
for z in range(nz):
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(maskc.ptr + z * mp.nbytes), mp.ctypes.data_as(C.c_void_p), mp.nbytes, None)
cen = DeviceArray.from_numpy((np.arange(nz) - nz // 2) * 500.0)
vox = nz * ny * nx
def report(name, ms, bytes_per_vox):
    print("%-34s %8.3f ms  %9.1f Mvox/s  %7.1f GB/s  %5.1f%% of 8TB/s" % (name, ms, vox / ms / 1e3, vox * bytes_per_vox / ms / 1e6, vox * bytes_per_vox / ms / 1e6 / 80), flush=True)
ws = DeviceArray((max(1, _lib.load().spc_moments_workspace_bytes(nz, ny, nx)),), np.uint8)
for env in ({}, {"SPC_MOMENTS_ZW": "1"}, {"SPC_MOMENTS_NT": "0"}, {"SPC_MOMENTS_ZW": "1", "SPC_MOMENTS_NT": "0"}):
    for k in ("SPC_MOMENTS_ZW", "SPC_MOMENTS_NT"): os.environ.pop(k, None)
    os.environ.update(env)
    ms = timeit(lambda: ops.moments(cube, cen, mask=None, workspace=ws))
    report("moments012 nomask %s" % env, ms, 4)
    ms = timeit(lambda: ops.moments(cube, cen, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc), workspace=ws))
    report("moments012 u8mask %s" % env, ms, 5)
for k in ("SPC_MOMENTS_ZW", "SPC_MOMENTS_NT"): os.environ.pop(k, None)
ms = timeit(lambda: ops.moments(cube, cen, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=maskc), workspace=ws, want=("m0","m1","m2","argmax")))
report("moments012+argmax u8mask", ms, 5)
ms = timeit(lambda: ops.moments(cube, cen, mask=ops.MaskSpec(_lib.MASK_GT|_lib.MASK_FINITE, 0.5), workspace=ws))
report("moments012 predicate mask", ms, 4)
if "--conv" in sys.argv:
    from math import exp
    g = np.exp(-0.5 * (np.arange(-16, 17) / 4.0) ** 2); g /= g.sum()
    out = DeviceArray(shape, np.float32)
    ms = timeit(lambda: ops.spectral_conv(cube, g, out=out), n=3, warm=1)
    report("spectral_conv 33 taps", ms, 8)
    cen_h = (np.arange(nz) - nz // 2) * 500.0
    ms = timeit(lambda: ops.spectral_conv_moments(cube, g, cen, cen_host=cen_h), n=3, warm=1)
    report("spectral_conv->moments fused", ms, 4)
    g29 = np.exp(-0.5 * (np.arange(-14, 15) / 3.397) ** 2); g29 /= g29.sum()
    ms = timeit(lambda: ops.spatial_conv(cube, np.outer(g29, g29), out=out), n=3, warm=1)
    report("spatial_conv 29x29 sep", ms, 8)
    x = np.arange(nz) * 1.0
    lo, t, inv, _, _, fill = ops.lerp_plan(x, np.linspace(0, nz - 1, 2 * nz))
    out2 = DeviceArray((2 * nz, ny, nx), np.float32)
    ms = timeit(lambda: ops.spectral_lerp(cube, lo, t, inv, fill, out=out2), n=3, warm=1)
    report("spectral_lerp x2 (12B/in-vox)", ms, 12)
