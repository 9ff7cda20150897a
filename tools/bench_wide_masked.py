"""masked separable spatial_smooth with 29 .. 65 taps at 256 x 2048^2 + 80 %-valid uint8 mask: the split form against the ring kernels"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from bench import replicate_planes
nz, ny, nx = 256, 2048, 2048
rng = np.random.default_rng(2003)
tile = rng.standard_normal((2, ny, nx), dtype=np.float32) + 2.0
tmask = (rng.random((2, ny, nx), dtype=np.float32) > 0.2).view(np.uint8)
cube = DeviceArray((nz, ny, nx), np.float32); replicate_planes(cube, tile)
maskd = DeviceArray((nz, ny, nx), np.uint8); replicate_planes(maskd, tmask)
spec = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
sm = DeviceArray((nz, ny, nx), np.float32)
m0 = DeviceArray((ny, nx), np.float64)

def ev(fn, n=5, warm=2):
    for _ in range(warm): fn()
    synchronize(0)
    e0, e1 = Event(0), Event(0); ts = []
    for _ in range(n):
        e0.record(None); fn(); e1.record(None); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return np.median(ts)

base = None
for sd in (8 / 2.3548200450309493, 4.0, 5.0, 6.0, 8.0):
    k2 = Gaussian2DKernel(sd).array
    taps = k2.shape[0]
    os.environ.pop("SPC_SPATIAL_RING", None)
    t_split = ev(lambda: ops.spatial_conv(cube, k2, mask=spec, out=sm))
    t_fused = ev(lambda: ops.spatial_conv_mfma(cube, k2, mask=spec, want_cube=False, want_m0=True, dv=1.0, m0=m0))
    os.environ["SPC_SPATIAL_RING"] = "1"
    t_ring = ev(lambda: ops.spatial_conv(cube, k2, mask=spec, out=sm))
    os.environ.pop("SPC_SPATIAL_RING", None)
    if base is None: base = (taps, t_split, t_fused)
    print("%2d taps: split cube -> cube %7.3f ms (per tap x%.2f of 29 taps), fused moment 0 %7.3f ms (x%.2f), ring kernel %7.3f ms" % (
        taps, t_split, (t_split / taps) / (base[1] / base[0]), t_fused, (t_fused / taps) / (base[2] / base[0]), t_ring), flush=True)
