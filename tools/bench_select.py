"""Scratch: order statistics timing at 1024^3 (python tools/bench_select.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, _lib, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows
shape = tuple(int(s) for s in (sys.argv[1:4] or (1024, 1024, 1024)))
tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
tmask = synth.boolean_mask(tile, 2001)
cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
_replicate_rows(cube, tile, 4); _replicate_rows(mask, tmask, 1)
mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
def timeit(fn, n=3):
    fn(); synchronize(); e0, e1 = Event(), Event(); e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_ms(e1) / n
qcube = DeviceArray(shape, np.float32)                  # quantised data (heavy ties: the candidates never get few)
_replicate_rows(qcube, np.round(tile * 4).astype(np.float32), 4)
for env in ({}, {"SPC_SELECT_REG": "0"}):
    for k in ("SPC_SELECT_REG",): os.environ.pop(k, None)
    os.environ.update(env)
    print(env or "default (rays in registers)", "median u8 mask %.3f ms | no mask %.3f ms | p90 %.3f ms | median along y (swap01; default row: along x, no transpose) %.3f ms | quantised, no mask %.3f ms" % (
        timeit(lambda: ops.percentile_axis0(cube, 50.0, mask=mspec)), timeit(lambda: ops.percentile_axis0(cube, 50.0)),
        timeit(lambda: ops.percentile_axis0(cube, 90.0, mask=mspec)),
        timeit(lambda: ops.percentile_axis0(cube.swap01(), 50.0, mask=mspec.swap01())) if env else timeit(lambda: ops.percentile_axis2(cube, 50.0, mask=mspec)),
        timeit(lambda: ops.percentile_axis0(qcube, 50.0))), flush=True)
os.environ.pop("SPC_SELECT_REG", None)
import time
t0 = time.perf_counter(); out = ops.sigma_clip_axis0(cube, sigma=3.0, mask=mspec); synchronize(); print("sigma_clip wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); out = ops.sigma_clip_axis0(cube, sigma=3.0, mask=mspec); synchronize(); print("sigma_clip wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); out = ops.sigma_clip_axis0(cube, sigma=3.0); synchronize(); print("sigma_clip (no mask) wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter(); out = ops.sigma_clip_axis0(cube, sigma=3.0, mask=mspec, stdfunc="mad_std"); synchronize(); t0 = time.perf_counter()
out = ops.sigma_clip_axis0(cube, sigma=3.0, mask=mspec, stdfunc="mad_std"); synchronize(); print("sigma_clip mad_std wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
os.environ["SPC_SIGMA_CLIP_FUSED"] = "0"
out = ops.sigma_clip_axis0(cube, sigma=3.0, mask=mspec, stdfunc="mad_std"); synchronize(); t0 = time.perf_counter()
out = ops.sigma_clip_axis0(cube, sigma=3.0, mask=mspec, stdfunc="mad_std"); synchronize(); print("sigma_clip mad_std, loop of separate kernels, wall %.1f ms" % ((time.perf_counter() - t0) * 1e3))
