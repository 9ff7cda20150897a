"""Scratch: sigma clip at 1024^3 by iteration count and centre function (where the time of the clip kernel goes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, _lib, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows
shape = (1024, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
cube = DeviceArray(shape, np.float32)
_replicate_rows(cube, tile, 4)
out = DeviceArray(shape, np.float32)
def timeit(fn, n=3):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
for bt in ("256", "512"):
  os.environ["SPC_SIGMA_BT"] = bt
  print("SPC_SIGMA_BT", bt)
  for cen in ("median", "mean"):
    for it in (1, 2, 3, 5, None):
        kw = dict(cenfunc=cen)
        kw["maxiters"] = it
        print("cenfunc=%s maxiters=%s: %.2f ms" % (cen, it, timeit(lambda: ops.sigma_clip_axis0(cube, sigma=3.0, **kw))), flush=True)
print("mad_std maxiters=1: %.2f ms" % timeit(lambda: ops.sigma_clip_axis0(cube, sigma=3.0, stdfunc="mad_std", maxiters=1)))
