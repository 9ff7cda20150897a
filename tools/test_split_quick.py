"""quick look at the split form (SPC_SPATIAL_MFMA_FORM=3) of the masked spatial stencil: errors against the oracle on small
shapes (smoothed cube, fused moment 0, fused moments 0 / 1 / 2), then timings against form 2 at 512 x 2048^2"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np
import oracle_np as O
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray, Event, synchronize


def case(shape, seed, valid=0.8, nan_frac=0.0, scale=1.0, offset=2.0):
    rng = np.random.default_rng(seed)
    d = ((rng.standard_normal(shape) + offset) * scale).astype(np.float32)
    m = rng.random(shape) < valid
    if nan_frac:
        d[rng.random(shape) < nan_frac] = np.nan
    return d, m


def err(got, exp, what):
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    nanok = np.array_equal(np.isnan(got), np.isnan(exp))
    ok = np.isfinite(exp) & np.isfinite(got)
    scale = np.nanmax(np.abs(exp)) if ok.any() else 1.0
    e = np.abs(got[ok] - exp[ok]).max() / scale if ok.any() else 0.0
    print("  %-58s scaled err %.2e  NaN pattern %s%s" % (what, e, "same" if nanok else "DIFFERS (%d vs %d)" % (np.isnan(got).sum(), np.isnan(exp).sum()),
                                                        "" if e <= 1e-5 and nanok else "   <<<<<< FAIL"), flush=True)
    return e <= 1e-5 and nanok


def main():
    _lib.require_gpu()
    allok = True
    k8 = Gaussian2DKernel(8 / 2.3548200450309493).array
    for shape, fwhm, valid, scale in (((3, 40, 64), 8.0, 0.8, 1.0), ((5, 33, 132), 8.0, 0.5, 1e-4), ((2, 70, 964), 8.0, 0.05, 3e4),
                                      ((4, 16, 480), 4.0, 0.9, 1.0), ((3, 50, 1000), 8.0, 1.0, 1.0), ((3, 97, 1040), 8.0, 1.0, 1e6),
                                      ((2, 130, 260), 8.0, 0.7, 1.0)):
        d, m = case(shape, 3, valid=valid, scale=scale)
        k2 = Gaussian2DKernel(fwhm / 2.3548200450309493).array
        cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
        out, _ = ops.spatial_conv_mfma(cube, k2, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
        exp = O.spatial_smooth(d, m, k2)
        allok &= err(out.get(), exp, "smoothed cube %s valid %.2f scale %g" % (shape, valid, scale))
    # dynamic range inside one wave region: a bright compact source on a faint background
    d, m = case((2, 96, 192), 11, valid=0.9, scale=1e-3, offset=0.0)
    d[:, 40:43, 90:93] += 5e3
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    out, _ = ops.spatial_conv_mfma(cube, k8, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
    allok &= err(out.get(), O.spatial_smooth(d, m, k8), "bright source on a faint background (dynamic range 5e6)")
    # fused moment 0, both mask forms, NaN samples
    shape = (37, 45, 528)
    d, m = case(shape, 9, valid=0.7, nan_frac=0.01)
    m[:, 3:6, 10:14] = False
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    sm = O.spatial_smooth(d, m, k8)
    for flags, inc in ((_lib.MASK_ARRAY, m), (_lib.MASK_ARRAY | _lib.MASK_FINITE, m & np.isfinite(d))):
        _, m0 = ops.spatial_conv_mfma(cube, k8, mask=ops.MaskSpec(flags, array=mk), want_cube=False, want_m0=True, dv=500.0)
        filled = np.where(inc, sm, np.nan)
        exp = 500.0 * np.nansum(filled, axis=0)
        exp[np.all(np.isnan(filled), axis=0)] = np.nan
        allok &= err(m0.get(), exp, "fused moment 0, flags %d" % flags)
        # moments 0 / 1 / 2
        cen = (np.arange(shape[0]) - shape[0] // 2) * 500.0
        d_cen = DeviceArray.from_numpy(cen)
        _, maps = ops.spatial_conv_mfma_moments(cube, k8, d_cen, dv=500.0, m1_add=123.0, mask=ops.MaskSpec(flags, array=mk))
        f0 = np.where(inc, sm, 0.0); f0[np.isnan(f0)] = 0.0
        s0 = f0.sum(0); s1 = (f0 * cen[:, None, None]).sum(0); s2 = (f0 * (cen ** 2)[:, None, None]).sum(0)
        with np.errstate(all="ignore"):
            e1 = s1 / s0 + 123.0
            e2 = s2 / s0 - (s1 / s0) ** 2
        allok &= err(maps["m0"].get(), exp, "fused moments: m0, flags %d" % flags)
        g1, g2 = maps["m1"].get(), maps["m2"].get()
        ok = np.isfinite(e1)
        span = 500.0 * shape[0]
        print("  fused moments: m1 err / span %.2e, m2 err / max %.2e, NaN %s" % (np.abs(g1[ok] - e1[ok]).max() / span, np.abs(g2[ok] - e2[ok]).max() / np.abs(e2[ok]).max(),
              np.array_equal(np.isnan(g1), np.isnan(e1))), flush=True)
    # many channels, several chunks, store + m0 together
    shape = (150, 16, 96)
    d, m = case(shape, 5, valid=0.6)
    k17 = Gaussian2DKernel(2.0).array
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    out, m0 = ops.spatial_conv_mfma(cube, k17, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk), want_cube=True, want_m0=True, dv=2.0)
    sm = O.spatial_smooth(d, m, k17)
    allok &= err(out.get(), sm, "17 taps, 150 channels: smoothed cube")
    allok &= err(m0.get(), 2.0 * np.nansum(np.where(m, sm, np.nan), axis=0), "17 taps, 150 channels: moment 0")
    print("ALL OK" if allok else "FAILURES", flush=True)

    if len(sys.argv) > 1 and sys.argv[1] == "time":
        nz, ny, nx = int(sys.argv[2]) if len(sys.argv) > 2 else 512, 2048, 2048
        rng = np.random.default_rng(2003)
        tile = rng.standard_normal((2, ny, nx), dtype=np.float32) + 2.0
        tmask = (rng.random((2, ny, nx), dtype=np.float32) > 0.2).view(np.uint8)
        sys.path.insert(0, REPO)
        from bench import replicate_planes
        cube = DeviceArray((nz, ny, nx), np.float32); replicate_planes(cube, tile)
        maskd = DeviceArray((nz, ny, nx), np.uint8); replicate_planes(maskd, tmask)
        spec = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
        sm = DeviceArray((nz, ny, nx), np.float32)
        m0 = DeviceArray((ny, nx), np.float64)
        d_cen = DeviceArray.from_numpy((np.arange(nz) - nz // 2) * 500.0)

        def ev(fn, n=5, warm=2):
            for _ in range(warm): fn()
            synchronize(0)
            e0, e1 = Event(0), Event(0); ts = []
            for _ in range(n):
                e0.record(None); fn(); e1.record(None); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
            return np.median(ts), min(ts)

        vox = nz * ny * nx
        for form in ("3", "2"):
            os.environ["SPC_SPATIAL_MFMA_FORM"] = form
            rows = [("store", lambda: ops.spatial_conv_mfma(cube, k8, mask=spec, out=sm), 9),
                    ("moment0 only (fused)", lambda: ops.spatial_conv_mfma(cube, k8, mask=spec, want_cube=False, want_m0=True, dv=500.0, m0=m0), 5)]
            if form == "3":
                rows.append(("moments 0/1/2 (fused)", lambda: ops.spatial_conv_mfma_moments(cube, k8, d_cen, dv=500.0, mask=spec), 5))
            for name, fn, byt in rows:
                med, mn = ev(fn)
                print("form %s %-28s median %8.3f ms  min %8.3f ms  -> at 4096 planes %7.2f ms  %.3f of 8 TB/s" % (form, name, med, mn, med * 4096 / nz, vox * byt / (med * 1e-3) / 8e12), flush=True)


if __name__ == "__main__":
    main()
