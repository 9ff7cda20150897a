"""Scratch: block -> tile order that keeps neighbouring 64-byte tiles on one XCD (SPC_XCD_GROUP), sigma clip floor and selection."""
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
def timeit(fn, n=3):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
os.environ["SPC_SIGMA_BT"] = "256"
for g in ("8", "16", "32", "64"):
    os.environ["SPC_XCD_GROUP"] = g
    os.environ["SPC_SELECT_BT"] = "256"
    a = timeit(lambda: ops.percentile_axis0(cube, 50.0))
    os.environ.pop("SPC_SELECT_BT")
    b = timeit(lambda: ops.percentile_axis0(cube, 50.0))
    print("group %s: sigma clip mean/1 iter %.2f ms | median defaults %.2f ms | select 256-thread table %.2f ms | select default %.2f ms" % (
        g, timeit(lambda: ops.sigma_clip_axis0(cube, sigma=3.0, cenfunc="mean", maxiters=1)), timeit(lambda: ops.sigma_clip_axis0(cube, sigma=3.0)), a, b), flush=True)
