import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, _lib, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows
shape = (1024, 1024, 1024)
tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
tmask = synth.boolean_mask(tile, 2001)
cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
_replicate_rows(cube, tile, 4); _replicate_rows(mask, tmask, 1)
mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
def timeit(fn, n=5):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
ref = {}
for bt in ("", "1024", "256"):
    if bt: os.environ["SPC_SELECT_BT"] = bt
    a = ops.percentile_axis0(cube, 50.0, mask=mspec).get(); b = ops.percentile_axis0(cube, 50.0).get()
    if not ref: ref = dict(a=a, b=b)
    print("BT", bt or "default", "masked %.3f ms  no mask %.3f ms" % (timeit(lambda: ops.percentile_axis0(cube, 50.0, mask=mspec)), timeit(lambda: ops.percentile_axis0(cube, 50.0))),
          "equal:", np.array_equal(a, ref["a"], equal_nan=True), np.array_equal(b, ref["b"], equal_nan=True), flush=True)
