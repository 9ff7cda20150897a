"""Scratch: sigma clipping (astropy defaults) by ray length, short-ray table on / off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, _lib, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows
def timeit(fn, n=3):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
for shape in ((100, 2048, 4096), (256, 2048, 2048), (512, 1024, 2048), (1024, 1024, 1024)):
    tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
    cube = DeviceArray(shape, np.float32)
    _replicate_rows(cube, tile, 4)
    row = []
    for sh in ("0", "1"):
        os.environ["SPC_SIGMA_SHORT"] = sh
        row.append("short=%s: %.2f ms (maxiters=1: %.2f)" % (sh, timeit(lambda: ops.sigma_clip_axis0(cube, sigma=3.0)), timeit(lambda: ops.sigma_clip_axis0(cube, sigma=3.0, maxiters=1))))
    print(shape, " | ".join(row), flush=True)
    del cube
