"""Scratch: spatial stencils (29 taps) at 512 x 2048^2, all valid and with a uint8 mask, with / without the XCD-aware block order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from spectral_cube_amd import ops, _lib, Gaussian2DKernel
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from bench_configs_helpers import replicate_planes
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (nz, 2048, 2048)
rng = np.random.default_rng(2003)
tile = rng.standard_normal((2,) + shape[1:], dtype=np.float32) + 2.0
tmask = (rng.random((2,) + shape[1:], dtype=np.float32) > 0.2).view(np.uint8)
cube, mask, out = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8), DeviceArray(shape, np.float32)
replicate_planes(cube, tile); replicate_planes(mask, tmask)
k2 = Gaussian2DKernel(8 / 2.3548200450309493).array
mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
def t(fn):
    for _ in range(2): fn()
    synchronize(); ts = []
    for _ in range(7):
        e0, e1 = Event(), Event(); e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))
for sw in ("0", "1", "0", "1"):
    os.environ["SPC_XCD_SWIZZLE"] = sw
    print("XCD swizzle %s: all valid %.3f ms | uint8 mask %.3f ms" % (sw, t(lambda: ops.spatial_conv(cube, k2, out=out)), t(lambda: ops.spatial_conv(cube, k2, mask=mspec, out=out))), flush=True)
