"""median(axis=0) and sigma_clip_spectrally at 1024^3 + uint8 mask, with and without the packing of sparse rays (SPC_SELECT_COMPACT):
a signal mask (data > 2 sigma on narrow lines: the bench's configs[1] mask, 4 - 5 % valid), random masks of 5 / 10 / 20 / 80 % and no mask"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from spectral_cube_amd import _lib, ops, synth
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from bench import replicate_planes

nz = ny = nx = 1024
rng = np.random.default_rng(11)


def ev(fn, n=7, warm=2):
    for _ in range(warm): fn()
    synchronize(0)
    e0, e1 = Event(0), Event(0); ts = []
    for _ in range(n):
        e0.record(None); fn(); e1.record(None); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
    return np.median(ts)


def main():
    _lib.require_gpu()
    rows = 8
    tile = synth.gaussian_line_cube((nz, rows, nx), 3, chunk_rows=rows)
    sig = synth.boolean_mask(tile, 3).astype(bool)
    host = np.ascontiguousarray(np.tile(tile, (1, ny // rows, 1)))
    cube = DeviceArray.from_numpy(host)
    med = DeviceArray((ny, nx), np.float32)
    keep = {}
    masks = [("signal mask (synth.boolean_mask)", np.tile(sig, (1, ny // rows, 1)))]
    for f in (0.05, 0.10, 0.20, 0.80):
        masks.append(("random %2.0f %% valid" % (100 * f), np.tile(rng.random((nz, rows, nx)) < f, (1, ny // rows, 1))))
    masks.append(("no mask", None))
    for name, m in masks:
        spec = None
        if m is not None:
            spec = ops.MaskSpec(_lib.MASK_ARRAY, array=DeviceArray.from_numpy(np.ascontiguousarray(m).astype(np.uint8)))
        res = {}
        for c in ("1", "0"):
            os.environ["SPC_SELECT_COMPACT"] = c
            t_med = ev(lambda: ops.percentile_axis0(cube, 50.0, mask=spec, out=med))
            a = med.get().copy()

            def clip():
                keep["r"] = None
                keep["r"] = ops.sigma_clip_axis0(cube, sigma=3.0, mask=spec)
            t_clip = ev(clip, n=5, warm=1)
            b = keep["r"].get()[:, :rows].copy()
            keep.clear()
            res[c] = (t_med, t_clip, a, b)
        same_med = np.array_equal(res["1"][2], res["0"][2], equal_nan=True)
        differ = np.mean(np.isnan(res["1"][3]) != np.isnan(res["0"][3]))
        print("%-30s valid %.3f: median %6.3f ms (packing off %6.3f), sigma clip %6.3f ms (off %6.3f); medians identical %s, clipped sets differ on %.1e of the samples" % (
            name, 1.0 if m is None else m.mean(), res["1"][0], res["0"][0], res["1"][1], res["0"][1], same_med, differ), flush=True)
    os.environ.pop("SPC_SELECT_COMPACT", None)


if __name__ == "__main__":
    main()
