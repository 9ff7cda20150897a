"""Scratch: where the time of the register-resident selection goes (library built with -DSPC_ABLATE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, _lib, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1024, 1024, 1024)]
for shape in shapes:
    print(shape, flush=True)
    tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
    tmask = synth.boolean_mask(tile, 2001)
    cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
    _replicate_rows(cube, tile, 4); _replicate_rows(mask, tmask, 1)
    mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
    def timeit(fn, n=5):
        fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
        for _ in range(n): fn()
        e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
    for desc, bt in (("1", "256"), ("1", "512")):
        os.environ["SPC_SELECT_DESC"] = desc
        os.environ["SPC_SELECT_BT"] = bt
        for ab, what in (("0", "full"), ("1", "load only"), ("2", "1 pass"), ("3", "2 passes"), ("4", "3 passes"), ("5", "4 passes (no candidates)")):
            os.environ["SPC_SELECT_ABLATE"] = ab
            print("bt=%s desc=%s %-26s no mask %.3f ms | u8 mask %.3f ms" % (bt, desc, what, timeit(lambda: ops.percentile_axis0(cube, 50.0)),
                  timeit(lambda: ops.percentile_axis0(cube, 50.0, mask=mspec))), flush=True)
