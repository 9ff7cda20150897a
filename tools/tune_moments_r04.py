"""Scratch: the moment kernel's issue budget (round 4): sweep SPC_MOMENTS_{ZW,U,XCD} at a shape, event-timed."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize

def timeit(fn, n=int(os.environ.get("TUNE_N", "20")), warm=3):
    for _ in range(warm): fn()
    synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = Event(), Event()
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))

shape = tuple(int(s) for s in (sys.argv[1:4] or (1024, 1024, 1024)))
nz, ny, nx = shape
rng = np.random.default_rng(0)
cube = DeviceArray(shape, np.float32)
maskc = DeviceArray(shape, np.uint8)
plane = rng.standard_normal((ny, nx), dtype=np.float32)
for z in range(nz):
    pl = np.roll(plane, z * 7919, axis=1)
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(cube.ptr + z * pl.nbytes), pl.ctypes.data_as(C.c_void_p), pl.nbytes, None)
    mp = (pl > -0.5).astype(np.uint8)
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(maskc.ptr + z * mp.nbytes), mp.ctypes.data_as(C.c_void_p), mp.nbytes, None)
cen = DeviceArray.from_numpy((np.arange(nz) - nz // 2) * 500.0)
vox = nz * ny * nx
ws = DeviceArray((max(1, _lib.load().spc_moments_workspace_bytes(nz, ny, nx)),), np.uint8)
out = {k: DeviceArray((ny, nx), np.float64) for k in ("m0", "m1", "m2")}
out["argmax"] = DeviceArray((ny, nx), np.int64)
marr = ops.MaskSpec(_lib.MASK_ARRAY, array=maskc)
mthr = ops.MaskSpec(_lib.MASK_ARRAY | _lib.MASK_GT, array=maskc, thr_lo=-1.0)
mfin = ops.MaskSpec(_lib.MASK_FINITE)
ZWS = tuple(int(v) for v in os.environ.get("TUNE_ZW", "4,8,1").split(","))
US = tuple(int(v) for v in os.environ.get("TUNE_U", "4,8,2").split(","))
for zw, u, xcd in itertools.product(ZWS, US, (1, 0)):
    os.environ.update(SPC_MOMENTS_ZW=str(zw), SPC_MOMENTS_U=str(u), SPC_MOMENTS_XCD=str(xcd), SPC_MOMENTS_NSPLIT="1")
    t_n = timeit(lambda: ops.moments(cube, cen, mask=None, workspace=ws, out=out))
    t_f = timeit(lambda: ops.moments(cube, cen, mask=mfin, workspace=ws, out=out))
    t_m = timeit(lambda: ops.moments(cube, cen, mask=marr, workspace=ws, out=out))
    t_t = timeit(lambda: ops.moments(cube, cen, mask=mthr, workspace=ws, out=out))
    t_e = timeit(lambda: ops.moments(cube, cen, mask=marr, workspace=ws, out=out, want=("m0", "m1", "m2", "argmax")))
    print("zw=%d u=%d xcd=%d  nomask %.3f ms %.2f TB/s | finite %.3f | u8mask %.3f ms %.2f TB/s | u8 + thr %.3f | +argmax %.3f ms" % (
        zw, u, xcd, t_n, vox * 4 / t_n / 1e9, t_f, t_m, vox * 5 / t_m / 1e9, t_t, t_e), flush=True)
