"""Scratch: does holding back the first generation's second block of every CU (SPC_SELECT_STAGGER = mode + 16 * units of ~4 us)
take the median's two blocks per CU out of step?  python tools/bench_select_stagger.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, _lib, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows
shape = (1024, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
tmask = synth.boolean_mask(tile, 2001)
rng = np.random.default_rng(5)
dmask = (rng.random(tile.shape) < 0.8).astype(np.uint8)
cube, mask, dense = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8), DeviceArray(shape, np.uint8)
_replicate_rows(cube, tile, 4); _replicate_rows(mask, tmask, 1); _replicate_rows(dense, dmask, 1)
ms, md = ops.MaskSpec(_lib.MASK_ARRAY, array=mask), ops.MaskSpec(_lib.MASK_ARRAY, array=dense)
def timeit(fn, n=5):
    fn(); synchronize(); ts = []
    for _ in range(n):
        e0, e1 = Event(), Event(); e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))
ref = None
for st in (0, 1 + 16 * 1, 1 + 16 * 2, 1 + 16 * 3, 1 + 16 * 4, 1 + 16 * 6, 1 + 16 * 8, 2 + 16 * 2, 2 + 16 * 3, 2 + 16 * 4, 2 + 16 * 6, 0):
    os.environ["SPC_SELECT_STAGGER"] = str(st)
    a = timeit(lambda: ops.percentile_axis0(cube, 50.0, mask=ms)); b = timeit(lambda: ops.percentile_axis0(cube, 50.0, mask=md))
    c = timeit(lambda: ops.percentile_axis0(cube, 50.0))
    out = ops.percentile_axis0(cube, 50.0, mask=md).get()
    if ref is None: ref = out
    print("stagger mode %d units %d: median signal mask %.3f ms | dense mask %.3f ms | no mask %.3f ms | identical %s" % (
        st & 15, st >> 4, a, b, c, np.array_equal(ref, out, equal_nan=True)), flush=True)
