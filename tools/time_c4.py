"""scratch: the masked C4 records alone (fused moment0, fused moment 0/1/2, cube -> cube) at nz planes"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray, Event, synchronize
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 512
which = sys.argv[2] if len(sys.argv) > 2 else "f0,f012,mat"
ny = nx = 2048
rng = np.random.default_rng(3)
tile = rng.standard_normal((2, ny, nx), dtype=np.float32) + 2.0
tm = (rng.random((2, ny, nx), dtype=np.float32) > 0.2).view(np.uint8)
if len(sys.argv) > 3 and sys.argv[3] == "sig":      # the bench's coherent SIGNAL mask (35 % valid) instead of the random one
    yy_, xx_ = np.mgrid[0:ny, 0:nx]
    tm = np.stack([(np.sin(xx_ / 37.0) * np.cos(yy_ / 53.0) > 0.2), (np.sin(xx_ / 41.0 + 1.0) * np.cos(yy_ / 47.0) > 0.2)]).view(np.uint8)
cube, mask = DeviceArray((nz, ny, nx), np.float32), DeviceArray((nz, ny, nx), np.uint8)
for dev, host in ((cube, tile), (mask, tm)):
    import ctypes as C
    plane = ny * nx * dev.dtype.itemsize
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(dev.ptr), host.ctypes.data_as(C.c_void_p), 2 * plane, None)
    have = 2
    while have < nz:
        n = min(have, nz - have)
        _lib.call("spc_memcpy_d2d", 0, C.c_void_p(dev.ptr + have * plane), C.c_void_p(dev.ptr), n * plane, None)
        have += n
synchronize(0)
ms = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
k2 = Gaussian2DKernel(8 / 2.3548200450309493).array
def timeit(fn, n=5):
    fn(); synchronize(0); ts = []
    for _ in range(n):
        e0, e1 = Event(0), Event(0); e0.record(None); fn(); e1.record(None); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))
m0 = DeviceArray((ny, nx), np.float64)
cen = DeviceArray.from_numpy((np.arange(nz) - nz // 2) * 500.0)
out = DeviceArray((nz, ny, nx), np.float32) if "mat" in which else None
res = []
if "f0" in which.split(","):
    res.append("fused m0 %.3f" % timeit(lambda: ops.spatial_conv_mfma(cube, k2, mask=ms, want_cube=False, want_m0=True, dv=500.0, m0=m0)))
if "f012" in which.split(","):
    res.append("fused m012 %.3f" % timeit(lambda: ops.spatial_conv_mfma_moments(cube, k2, cen, dv=500.0, m1_add=0.0, mask=ms)))
if "mat" in which.split(","):
    res.append("cube->cube %.3f" % timeit(lambda: ops.spatial_conv_mfma(cube, k2, mask=ms, out=out)))
print("nz=%d mask=%s dead-skip=%s: " % (nz, sys.argv[3] if len(sys.argv) > 3 else "random", os.environ.get("SPC_SPLIT_DEAD", "1")) + ", ".join(res) + "  (ms; x%d for 4096 planes)" % (4096 // nz))
