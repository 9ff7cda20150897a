"""Scratch: sweep the SPC_MOMENTS_* tuning hooks on the GPU (event-timed)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize

def timeit(fn, n=20, warm=3):
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
cube = DeviceArray(shape, np.float32)
maskc = DeviceArray(shape, np.uint8)
for z in range(nz):
    plane = rng.standard_normal((ny, nx), dtype=np.float32)
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(cube.ptr + z * plane.nbytes), plane.ctypes.data_as(C.c_void_p), plane.nbytes, None)
    mp = (plane > -0.5).astype(np.uint8)
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(maskc.ptr + z * mp.nbytes), mp.ctypes.data_as(C.c_void_p), mp.nbytes, None)
cen = DeviceArray.from_numpy((np.arange(nz) - nz // 2) * 500.0)
vox = nz * ny * nx
os.environ["SPC_MOMENTS_VEC"] = "4"
ws = DeviceArray((max(1, _lib.load().spc_moments_workspace_bytes(nz, ny, nx)),), np.uint8)
out = {k: DeviceArray((ny, nx), np.float64) for k in ("m0", "m1", "m2")}
out["argmax"] = DeviceArray((ny, nx), np.int64)
marr = ops.MaskSpec(_lib.MASK_ARRAY, array=maskc)
res = []
for vec, zw, u, nt in itertools.product((4, 2, 1), (1, 2, 4), (2, 4, 8), (1, 0)):
    os.environ.update(SPC_MOMENTS_VEC=str(vec), SPC_MOMENTS_ZW=str(zw), SPC_MOMENTS_U=str(u), SPC_MOMENTS_NT=str(nt), SPC_MOMENTS_NSPLIT="1")
    t_n = timeit(lambda: ops.moments(cube, cen, mask=None, workspace=ws, out=out))
    t_m = timeit(lambda: ops.moments(cube, cen, mask=marr, workspace=ws, out=out))
    t_e = timeit(lambda: ops.moments(cube, cen, mask=marr, workspace=ws, out=out, want=("m0", "m1", "m2", "argmax")))
    res.append((t_m, vec, zw, u, nt, t_n, t_e))
    print("vec=%d zw=%d u=%d nt=%d  nomask %.3f ms %.2f TB/s | u8mask %.3f ms %.2f TB/s | +argmax %.3f ms" % (
        vec, zw, u, nt, t_n, vox * 4 / t_n / 1e9, t_m, vox * 5 / t_m / 1e9, t_e), flush=True)
res.sort()
print("best (u8mask):", res[:5])
