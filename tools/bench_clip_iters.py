"""sigma clip at 1024^3 + the signal mask by number of iterations (maxiters 0 = the read + write floor), packing on / off"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from spectral_cube_amd import _lib, ops, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
nz = ny = nx = 1024
rows = 8
tile = synth.gaussian_line_cube((nz, rows, nx), 3, chunk_rows=rows)
sig = synth.boolean_mask(tile, 3).astype(bool)
cube = DeviceArray.from_numpy(np.ascontiguousarray(np.tile(tile, (1, ny // rows, 1))))
rng = np.random.default_rng(1)
masks = {"signal": np.tile(sig, (1, ny // rows, 1)), "random 80 %": np.tile(rng.random((nz, rows, nx)) < 0.8, (1, ny // rows, 1))}
keep = {}


def ev(fn, n=5, warm=1):
    for _ in range(warm): fn()
    synchronize(0)
    e0, e1 = Event(0), Event(0); ts = []
    for _ in range(n):
        e0.record(None); fn(); e1.record(None); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return np.median(ts)


for name, m in masks.items():
    spec = ops.MaskSpec(_lib.MASK_ARRAY, array=DeviceArray.from_numpy(np.ascontiguousarray(m).astype(np.uint8)))
    for c in ("2", "1", "0"):
        os.environ["SPC_SELECT_COMPACT"] = c
        line = []
        for it in (0, 1, 2, 3, 5):
            def clip():
                keep["r"] = None
                keep["r"] = ops.sigma_clip_axis0(cube, sigma=3.0, mask=spec, maxiters=it)
            line.append("%d: %.2f" % (it, ev(clip)))
        for cen in ("mean",):
            def clip2():
                keep["r"] = None
                keep["r"] = ops.sigma_clip_axis0(cube, sigma=3.0, mask=spec, cenfunc=cen)
            line.append("cenfunc mean: %.2f" % ev(clip2))
        print("%-12s packing %s  maxiters -> ms  %s" % (name, c, "   ".join(line)), flush=True)
