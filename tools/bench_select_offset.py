"""Scratch: order statistics on data in a narrow relative range (every sample shares its leading key bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows
shape = (1024, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
def timeit(fn, n=3):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
for name, t in (("median near 0", tile), ("data + 10", tile + np.float32(10.0)), ("data + 1000", tile + np.float32(1000.0)), ("quantised", np.round(tile * 4).astype(np.float32))):
    cube = DeviceArray(shape, np.float32); _replicate_rows(cube, t, 4)
    with np.errstate(all="ignore"):
        exp = np.nanmedian(t, axis=0)
    got = ops.percentile_axis0(cube, 50.0).get()[:8]
    assert np.array_equal(got, exp, equal_nan=True), name
    print("%-14s median %.3f ms | sigma clip %.2f ms | sigma clip mad_std %.2f ms" % (name, timeit(lambda: ops.percentile_axis0(cube, 50.0)),
          timeit(lambda: ops.sigma_clip_axis0(cube, sigma=3.0)), timeit(lambda: ops.sigma_clip_axis0(cube, sigma=3.0, stdfunc="mad_std"))), flush=True)
    del cube
