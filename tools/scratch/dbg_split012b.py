import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray
K8 = Gaussian2DKernel(8 / 2.3548200450309493).array
for shape in ((8, 45, 200), (16, 45, 200), (90, 45, 200), (90, 100, 64)):
    for mode in ("ones", "rand"):
        rng = np.random.default_rng(9)
        d = np.ones(shape, np.float32)
        m = np.ones(shape, bool) if mode == "ones" else rng.random(shape) < 0.7
        cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
        cen = (np.arange(shape[0]) - shape[0] // 2) * 1.0
        _, maps = ops.spatial_conv_mfma_moments(cube, K8, DeviceArray.from_numpy(cen), dv=1.0, m1_add=0.0, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
        f0 = np.where(m, 1.0, 0.0)
        s0, s1 = f0.sum(0), (f0 * cen[:, None, None]).sum(0)
        g0, g1 = maps["m0"].get(), maps["m1"].get()
        e1 = s1 / s0
        ds1 = (g1 - e1) * s0          # error of S1 in units of channel x value
        bad = np.argwhere(np.abs(ds1) > 1e-3)
        print(shape, mode, "m0 err", np.nanmax(np.abs(g0 - s0)), "bad pixels", len(bad), "cols", np.unique(bad[:, 1])[:30], "rows", np.unique(bad[:, 0])[:50])
        if len(bad):
            for y, x in bad[:6]:
                print("   ", y, x, "dS1", ds1[y, x], "mask column", m[:, y, x].astype(int).tolist() if shape[0] <= 16 else "")
