import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np
import oracle_np as O
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray
K8 = Gaussian2DKernel(8 / 2.3548200450309493).array
shape = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (90, 45, 200)
rng = np.random.default_rng(9)
d = ((rng.standard_normal(shape) + 2.0)).astype(np.float32)
m = rng.random(shape) < 0.7
cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
cen = (np.arange(shape[0]) - shape[0] // 2) * 500.0
_, maps = ops.spatial_conv_mfma_moments(cube, K8, DeviceArray.from_numpy(cen), dv=500.0, m1_add=0.0, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
sm = O.spatial_smooth(d, m, K8)
f0 = np.where(m, sm, 0.0)
s0, s1, s2 = f0.sum(0), (f0 * cen[:, None, None]).sum(0), (f0 * (cen ** 2)[:, None, None]).sum(0)
e0, e1, e2 = 500 * s0, s1 / s0, s2 / s0 - (s1 / s0) ** 2
g0, g1, g2 = maps["m0"].get(), maps["m1"].get(), maps["m2"].get()
for nm, g, e in (("m0", g0, e0), ("m1", g1, e1), ("m2", g2, e2)):
    err = np.abs(g - e)
    print(nm, "max err", np.nanmax(err), "scale", np.nanmax(np.abs(e)), "rows with err > 1e-4 scale:", np.unique(np.argwhere(err > 1e-4 * np.nanmax(np.abs(e)))[:, 0])[:40],
          "cols:", np.unique(np.argwhere(err > 1e-4 * np.nanmax(np.abs(e)))[:, 1])[:60])
