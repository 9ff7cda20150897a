import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray
K8 = Gaussian2DKernel(8 / 2.3548200450309493).array
tot = 0
for shape in ((8, 45, 200), (16, 45, 200), (90, 45, 200), (90, 100, 64), (64, 300, 256)):
    d = np.ones(shape, np.float32)
    m = np.ones(shape, bool)
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    cen = (np.arange(shape[0]) - shape[0] // 2) * 1.0
    s0, s1 = float(shape[0]), cen.sum()
    for rep in range(3):
        _, maps = ops.spatial_conv_mfma_moments(cube, K8, DeviceArray.from_numpy(cen), dv=1.0, m1_add=0.0, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
        g0, g1, g2 = maps["m0"].get(), maps["m1"].get(), maps["m2"].get()
        e1 = s1 / s0
        ds1 = (g1 - e1) * g0
        bad = np.argwhere(np.abs(ds1) > 1e-3)
        tot += len(bad)
        print(shape, rep, "bad pixels", len(bad), "cols", np.unique(bad[:, 1])[:20], "rows", np.unique(bad[:, 0])[:8], "dS1", np.unique(np.round(ds1[np.abs(ds1) > 1e-3]))[:8])
print("TOTAL BAD", tot)
