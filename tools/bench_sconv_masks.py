"""masked spectral_smooth (33 taps) at 1024 x 2048 x 1024 under three masks, with and without the dead-step skipping"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np
from spectral_cube_amd import Gaussian1DKernel, _lib, ops, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows
shape = (2048, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 2, shape[2]), 2002, chunk_rows=2)
cube = DeviceArray(shape, np.float32); _replicate_rows(cube, tile, 4)
out = DeviceArray(shape, np.float32)
k = Gaussian1DKernel(4.0).array
cen = DeviceArray.from_numpy((np.arange(shape[0]) - shape[0] // 2) * 500.0)
masks = {"bench mask (data > 2 sigma | 1 % random)": synth.boolean_mask(tile, 2002),
         "signal mask (data > 5 sigma)": (tile > 2.5).view(np.uint8),
         "random 80 % valid": (np.random.default_rng(1).random(tile.shape) < 0.8).view(np.uint8)}
for name, tm in masks.items():
    md = DeviceArray(shape, np.uint8); _replicate_rows(md, np.ascontiguousarray(tm), 1)
    spec = ops.MaskSpec(_lib.MASK_ARRAY, array=md)
    for skip in ("0", "1"):
        os.environ["SPC_SPECTRAL_SKIP_DEAD"] = skip
        res = []
        for label, fn in (("materialised", lambda: ops.spectral_conv(cube, k, mask=spec, out=out)),
                          ("fused moment1", lambda: ops.spectral_conv_moments(cube, k, cen, dv=500.0, mask=spec, want=("m1",)))):
            ts = []
            for i in range(5):
                e0, e1 = Event(), Event()
                e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
            res.append("%s %.2f ms" % (label, np.median(ts[1:])))
        print("%-42s valid %.3f  skip=%s  %s  (x4 = C3 size)" % (name, tm.mean(), skip, ", ".join(res)), flush=True)
    del md
