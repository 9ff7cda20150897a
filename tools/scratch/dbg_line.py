import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray
K8 = Gaussian2DKernel(8 / 2.3548200450309493).array
nz, ny, nx = int(sys.argv[1]), 384, 256
z = np.arange(nz)
line = (np.exp(-0.5 * ((z - 0.95 * nz) / 2.0) ** 2) + 0.0).astype(np.float32)
rng = np.random.default_rng(77)
mrow = rng.random((nz, 1, nx)) < 0.7
d = np.broadcast_to(line[:, None, None], (nz, 8, nx)).copy()
mt = np.ascontiguousarray(np.broadcast_to(mrow, (nz, 8, nx)), dtype=np.uint8)      # (astype of a broadcast view keeps its zero stride: not C order)
cube, mk = DeviceArray((nz, ny, nx), np.float32), DeviceArray((nz, ny, nx), np.uint8)
for dev, host, isz in ((cube, d, 4), (mk, mt, 1)):
    row = nx * isz
    _lib.call("spc_memcpy3d_h2d", 0, C.c_void_p(dev.ptr), row, ny * row, host.ctypes.data_as(C.c_void_p), row, 8 * row, row, 8, nz, None)
    have = 8
    while have < ny:
        n = min(have, ny - have)
        _lib.call("spc_memcpy3d_d2d", 0, C.c_void_p(dev.ptr + have * row), row, ny * row, C.c_void_p(dev.ptr), row, ny * row, row, n, nz, None)
        have += n
_lib.call("spc_device_sync", 0)
back = cube.get()
print("replication ok:", np.array_equal(back[:, 200], d[:, 0]), np.array_equal(mk.get()[:, 333], mt[:, 0]))
cen = (z - nz // 2) * 1.0
_, maps = ops.spatial_conv_mfma_moments(cube, K8, DeviceArray.from_numpy(cen), dv=1.0, m1_add=0.0, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
inc = mrow[:, 0, :].astype(np.float64)
f = inc * line[:, None].astype(np.float64)
s0 = f.sum(0); mu = (f * cen[:, None]).sum(0) / s0
m2 = (f * (cen[:, None] - mu) ** 2).sum(0) / s0
g0, g1, g2 = maps["m0"].get(), maps["m1"].get(), maps["m2"].get()
x = 77
print("expected s0 mu m2 at x=77:", s0[x], mu[x], m2[x])
for y in (20, 50, 95, 96, 100, 150, 191, 192, 200, 287, 288, 300, 370):
    print(y, g0[y, x], g1[y, x], g2[y, x])
_, m0only = ops.spatial_conv_mfma(cube, K8, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk), want_cube=False, want_m0=True, dv=1.0)
print("m0-only form at rows 100, 200, 300:", m0only.get()[[100, 200, 300], x])
