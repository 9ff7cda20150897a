import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np
import oracle_np as O
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray
np.set_printoptions(linewidth=250, precision=4, suppress=True)
k8 = Gaussian2DKernel(8 / 2.3548200450309493).array
def run(d, m):
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    out, _ = ops.spatial_conv_mfma(cube, k8, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
    return out.get()
# 1. all ones, all valid, interior-sized plane
for shape in ((1, 40, 64), (1, 96, 192), (1, 16, 480)):
    d = np.ones(shape, np.float32); m = np.ones(shape, bool)
    g = run(d, m); e = O.spatial_smooth(d, m, k8)
    nanrows = np.where(np.isnan(g[0]).any(axis=1))[0]; nancols = np.where(np.isnan(g[0]).any(axis=0))[0]
    print(shape, "ones: NaN rows", nanrows[:40], "NaN cols", nancols[:80], "count", np.isnan(g).sum())
    ok = np.isfinite(g)
    print("   max err where finite", np.abs(g[ok] - e[ok]).max() if ok.any() else None)
    print("   got row 20 (or 8):", g[0, min(20, shape[1] - 1), :40])
    print("   exp row 20 (or 8):", e[0, min(20, shape[1] - 1), :40])
# 2. delta in the interior
shape = (1, 96, 192)
d = np.zeros(shape, np.float32); d[0, 50, 100] = 1.0; m = np.ones(shape, bool)
g = run(d, m); e = O.spatial_smooth(d, m, k8)
print("delta: NaN", np.isnan(g).sum(), "max err", np.nanmax(np.abs(g - e)), "peak got", np.nanmax(g), "at", np.unravel_index(np.nanargmax(g), g.shape), "exp", e.max(), np.unravel_index(e.argmax(), e.shape))
err = np.abs(g - e)[0]
print("   err rows with > 1e-6:", np.where((err > 1e-6).any(axis=1))[0], "cols:", np.where((err > 1e-6).any(axis=0))[0])
# 3. random, all valid, interior window error map
rng = np.random.default_rng(1)
d = (rng.standard_normal(shape) + 2).astype(np.float32); m = np.ones(shape, bool)
g = run(d, m); e = O.spatial_smooth(d, m, k8)
err = np.abs(g - e)[0] / np.abs(e).max()
print("random all valid: NaN", np.isnan(g).sum(), "max scaled err", np.nanmax(err))
print("   per row tile / col tile max err:")
for r0 in range(0, 96, 16):
    print("   ", ["%.1e" % np.nanmax(err[r0:r0 + 16, c0:c0 + 16]) for c0 in range(0, 192, 16)])
