"""timing of the MFMA-denominator spatial stencil against the grouped ring kernel (masked C4 shapes)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray, Event, synchronize

nz = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ny = nx = 2048
rng = np.random.default_rng(2003)
tile = rng.standard_normal((2, ny, nx), dtype=np.float32) + 2.0
tmask = (rng.random((2, ny, nx), dtype=np.float32) > 0.2).view(np.uint8)
sys.path.insert(0, ".")
from bench import replicate_planes
cube = DeviceArray((nz, ny, nx), np.float32); replicate_planes(cube, tile)
maskd = DeviceArray((nz, ny, nx), np.uint8); replicate_planes(maskd, tmask)
spec = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
k2 = Gaussian2DKernel(8 / 2.3548200450309493).array
sm = DeviceArray((nz, ny, nx), np.float32)
m0 = DeviceArray((ny, nx), np.float64)

def ev(fn, n=5, warm=2):
    for _ in range(warm): fn()
    synchronize(0)
    e0, e1 = Event(0), Event(0); ts = []
    for _ in range(n):
        e0.record(None); fn(); e1.record(None); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return np.median(ts), min(ts)

vox = nz * ny * nx
for name, fn, byt in (
    ("grouped ring kernel, store", lambda: ops.spatial_conv(cube, k2, mask=spec, out=sm), 9),
    ("mfma kernel, store", lambda: ops.spatial_conv_mfma(cube, k2, mask=spec, out=sm), 9),
    ("mfma kernel, moment0 only (fused)", lambda: ops.spatial_conv_mfma(cube, k2, mask=spec, want_cube=False, want_m0=True, dv=500.0, m0=m0), 5),
    ("mfma kernel, store + moment0", lambda: ops.spatial_conv_mfma(cube, k2, mask=spec, out=sm, want_m0=True, dv=500.0, m0=m0), 9),
):
    med, mn = ev(fn)
    print("%-40s median %8.3f ms  min %8.3f ms  -> at 4096 planes %7.2f ms  %.3f of 8 TB/s" % (name, med, mn, med * 4096 / nz, vox * byt / (med * 1e-3) / 8e12), flush=True)
