"""Scratch: timing-only ablations of the masked 29-tap spatial stencil (library built with -DSPC_ABLATE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import ops, _lib, Gaussian2DKernel
from spectral_cube_amd.device import DeviceArray, Event, synchronize
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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
names = {0: "baseline", 1: "no stores", 2: "no mask loads", 3: "no stores, no mask loads", 6: "no loads at all", 7: "no memory traffic at all",
         8: "no y-pass FMAs", 16: "no x-pass FMAs", 24: "no FMAs", 31: "skeleton: LDS + barriers + classification + division only"}
for abl in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,3,6,7,8,16,24,31".split(","))]:
    os.environ["SPC_SPATIAL_ABLATE"] = str(abl)
    for _ in range(2): ops.spatial_conv(cube, k2, mask=mspec, out=out)
    synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = Event(), Event(); e0.record(); ops.spatial_conv(cube, k2, mask=mspec, out=out); e1.record(); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    print("ABL %2d %-62s %7.3f ms" % (abl, names.get(abl, "?"), float(np.median(ts))), flush=True)
