import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray
K8 = Gaussian2DKernel(8 / 2.3548200450309493).array
shape = (8, 45, 200)
d = np.ones(shape, np.float32)
m = np.ones(shape, bool)
cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
cen = 10.0 ** np.arange(8)
cen[4] = 0.0            # the chunk's middle channel: delta_w = cen_w
s0, s1 = 8.0, cen.sum()
for rep in range(12):
    _, maps = ops.spatial_conv_mfma_moments(cube, K8, DeviceArray.from_numpy(cen), dv=1.0, m1_add=0.0, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
    g0, g1 = maps["m0"].get(), maps["m1"].get()
    S1 = g1 * g0
    ds1 = S1 - s1 * g0 / s0
    inner = np.zeros_like(ds1, bool); inner[15:30, 16:184] = True      # away from the plane's edges: every value is 1
    bad = np.argwhere((np.abs(ds1) > 0.5) & inner)
    if len(bad):
        print(rep, "bad", len(bad), "cols", np.unique(bad[:, 1])[:10], "rows", np.unique(bad[:, 0]), "dS1 values", np.unique(np.round(ds1[inner & (np.abs(ds1) > 0.5)]))[:10])
    else:
        print(rep, "ok")
