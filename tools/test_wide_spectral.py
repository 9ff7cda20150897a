"""masked spectral_smooth with 35 - 65 symmetric taps: the general form of the 49- / 65-tap rings (spectral_conv_ring_wide_kernel)
against the oracle (bit level), then timings against the runs-of-16 kernel (SPC_SPECTRAL_RING_WIDE=0) at 1024^3 + uint8 mask"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np
import oracle_np as O
from spectral_cube_amd import Gaussian1DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray, Event, synchronize


def check(shape, k, seed, thr=None):
    rng = np.random.default_rng(seed)
    d = (rng.standard_normal(shape) * 3 + 1).astype(np.float32)
    d[3:5, 1, 1] = np.nan
    d[:, 2, 3] = np.nan
    inc = rng.random(shape) > 0.3
    inc[10:10 + len(k) + 3, 3, 4] = False
    inc[:, 0, 0] = False
    ok = True
    for tag, m, spec in (("no mask", None, None), ("uint8 mask", inc, ops.MaskSpec(_lib.MASK_ARRAY, array=DeviceArray.from_numpy(inc.astype(np.uint8)))),
                         ("mask + isfinite", inc & np.isfinite(d), ops.MaskSpec(_lib.MASK_ARRAY | _lib.MASK_FINITE, array=DeviceArray.from_numpy(inc.astype(np.uint8))))):
        out = ops.spectral_conv(DeviceArray.from_numpy(d), k, mask=spec).get()
        exp = O.spectral_smooth(d, m, k)
        nanok = np.array_equal(np.isnan(out), np.isnan(exp))
        fin = np.isfinite(exp) & np.isfinite(out)
        differ = (out[fin] != exp[fin])
        ulp = np.all(np.abs(out[fin][differ] - exp[fin][differ]) <= np.spacing(np.abs(exp[fin][differ]))) if differ.any() else True
        good = nanok and differ.mean() <= 1e-4 and ulp and np.array_equal(np.isinf(out), np.isinf(exp))
        ok &= good
        print("  %s %2d taps %-16s NaN pattern %s, %d of %d differ (<= 1 ulp: %s)%s" % (shape, len(k), tag, "same" if nanok else "DIFFERS", differ.sum(), differ.size, ulp,
                                                                                     "" if good else "   <<<<< FAIL"), flush=True)
    return ok


def main():
    _lib.require_gpu()
    allok = True
    for sd in (4.5, 5.0, 6.0, 7.0, 8.0):
        k = Gaussian1DKernel(sd).array
        for shape in ((300, 6, 70), (97, 4, 64), (700, 4, 64), (40, 5, 9)):
            allok &= check(shape, k, 5)
    print("ALL OK" if allok else "FAILURES", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        nz = ny = nx = 1024
        rng = np.random.default_rng(7)
        sys.path.insert(0, REPO)
        from bench import replicate_planes
        tile = rng.standard_normal((2, ny, nx), dtype=np.float32) + 2.0
        cube = DeviceArray((nz, ny, nx), np.float32); replicate_planes(cube, tile)
        tmask = (rng.random((67, ny, nx), dtype=np.float32) > 0.2).view(np.uint8)
        maskd = DeviceArray((nz, ny, nx), np.uint8); replicate_planes(maskd, tmask)
        spec = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
        out = DeviceArray((nz, ny, nx), np.float32)

        def ev(fn, n=5, warm=1):
            for _ in range(warm): fn()
            synchronize(0)
            e0, e1 = Event(0), Event(0); ts = []
            for _ in range(n):
                e0.record(None); fn(); e1.record(None); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
            return np.median(ts)

        base = None
        for sd in (4.0, 5.0, 6.0, 8.0):
            k = Gaussian1DKernel(sd).array
            os.environ.pop("SPC_SPECTRAL_RING_WIDE", None)
            t_new = ev(lambda: ops.spectral_conv(cube, k, mask=spec, out=out))
            os.environ["SPC_SPECTRAL_RING_WIDE"] = "0"
            t_old = ev(lambda: ops.spectral_conv(cube, k, mask=spec, out=out), n=3)
            os.environ.pop("SPC_SPECTRAL_RING_WIDE", None)
            if base is None: base = (len(k), t_new)
            print("%2d taps, 1024^3 + uint8 mask: %7.3f ms (per tap x%.2f of %d taps); runs-of-16 kernel %7.3f ms" % (len(k), t_new, (t_new / len(k)) / (base[1] / base[0]), base[0], t_old), flush=True)


if __name__ == "__main__":
    main()
