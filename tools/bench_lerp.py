"""Scratch: C5 spectral_interpolate (2048 -> 4096 channels x 1024^2) only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, synth
from spectral_cube_amd.device import DeviceArray, Event
from test_gpu_fullsize import _replicate_rows
shape = (2048, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 2, shape[2]), 2004, chunk_rows=2)
cube = DeviceArray(shape, np.float32); _replicate_rows(cube, tile, 4)
v = synth.spectral_axis(shape[0]); grid = np.linspace(v[0], v[-1], 4096)
lo, t, inv, _, _, fill = ops.lerp_plan(v, grid)
out = DeviceArray((4096,) + shape[1:], np.float32)
for tiles, jc, nt in ((1, 0, 0), (0, 0, 1), (0, 32, 0), (1, 0, 0)):
    os.environ["SPC_LERP_TILES"] = str(tiles); os.environ["SPC_LERP_JCHUNK"] = str(jc); os.environ["SPC_LERP_NT"] = str(nt)
    ts = []
    for i in range(8):
        e0, e1 = Event(), Event()
        e0.record(); ops.spectral_lerp(cube, lo, t, inv, fill, out=out); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_ms(e1))
    m = np.median(ts[2:])
    print("tiles=%d jchunk=%2d nt=%d  median %.3f ms  min %.3f  %.0f GB/s algorithmic" % (tiles, jc, nt, m, min(ts), 3 * np.prod(shape, dtype=np.int64) * 4 / m / 1e6), flush=True)
