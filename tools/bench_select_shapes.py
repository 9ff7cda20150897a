"""Scratch: median along the spectral axis for every ray-length class, 512-thread table against the former 256-thread one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, _lib, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows
def timeit(fn, n=5):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
for shape in ((100, 2048, 4096), (256, 2048, 2048), (512, 1024, 2048), (1024, 1024, 1024), (2048, 512, 1024), (4096, 512, 512)):
    tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
    tmask = synth.boolean_mask(tile, 2001)
    cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
    _replicate_rows(cube, tile, 4); _replicate_rows(mask, tmask, 1)
    mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
    row = []
    for bt in ("256", "512", "0"):
        os.environ["SPC_SELECT_BT"] = bt
        row.append("bt=%s: no mask %.3f ms, u8 mask %.3f ms" % (bt, timeit(lambda: ops.percentile_axis0(cube, 50.0)), timeit(lambda: ops.percentile_axis0(cube, 50.0, mask=mspec))))
    os.environ.pop("SPC_SELECT_BT")
    gb = shape[0] * shape[1] * shape[2] * 4 / 1e9
    print(shape, "%.1f GB |" % gb, " | ".join(row), flush=True)
    del cube, mask
