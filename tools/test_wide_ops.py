"""Scratch: the float64 operators (spc_wide_ops.hip) against the oracle on float64 samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import oracle_np as O
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
dev = DeviceArray.from_numpy
fails = 0
def cmp(a, b, tol, what):
    global fails
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    nanbad = int((np.isnan(a) != np.isnan(b)).sum())
    fin = np.isfinite(a) & np.isfinite(b)
    sc = np.max(np.abs(b[fin])) if fin.any() else 1.0
    err = np.max(np.abs(a[fin] - b[fin])) / max(sc, 1e-300) if fin.any() else 0.0
    ok = nanbad == 0 and err <= tol
    if not ok: fails += 1
    print("%s %s: nan mismatches %d, max err / scale %.2e" % ("ok  " if ok else "FAIL", what, nanbad, err), flush=True)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 16):
    nz, ny, nx = int(rng.integers(1, 70)), int(rng.integers(1, 60)), int(rng.integers(1, 200))
    d = 1000.0 + rng.standard_normal((nz, ny, nx)) * 10.0 ** rng.uniform(-6, 1)          # float32 cannot hold these
    d[rng.random(d.shape) < rng.choice([0.0, 0.02, 0.3])] = np.nan
    kind = int(rng.integers(0, 3))
    thr = 1000.0 + 1e-9
    if kind == 0: inc, spec = None, None
    elif kind == 1:
        inc = rng.random(d.shape) > 0.3; spec = ops.MaskSpec(_lib.MASK_ARRAY, array=dev(inc.astype(np.uint8)))
    else:
        inc = (d > thr) & np.isfinite(d); spec = ops.MaskSpec(_lib.MASK_GT | _lib.MASK_FINITE, thr)
    tag = "it%d %s mask%d" % (it, (nz, ny, nx), kind)
    dd = dev(d)
    assert dd.dtype == np.float64
    st = ops.stats_global_f64(dd, mask=spec); es = O.statistics(d, inc)
    ok = st["npts"] == es["npts"] and (es["npts"] == 0 or (st["min"] == es["min"] and st["max"] == es["max"] and
         abs(st["sum"] - es["sum"]) <= 1e-12 * abs(es["sum"]) and abs(st["sumsq"] - es["sumsq"]) <= 1e-12 * abs(es["sumsq"])))
    if not ok: fails += 1; print("FAIL", tag, "stats_global", st, es)
    for ax in (0, 1, 2):
        ra = ops.stats_axis_f64(dd, ax, mask=spec)
        cmp(ra["sum"].get(), O.reduce(d, inc, "sum", axis=ax), 1e-13, tag + " sum ax%d" % ax)
        cmp(ra["max"].get(), O.reduce(d, inc, "max", axis=ax), 0.0, tag + " max ax%d" % ax)
        cmp(ra["min"].get(), O.reduce(d, inc, "min", axis=ax), 0.0, tag + " min ax%d" % ax)
        filled = O.filled(d, inc, np.nan)
        cnt = np.sum(~np.isnan(filled), axis=ax)
        if not np.array_equal(ra["count"].get(), cnt): fails += 1; print("FAIL", tag, "count ax%d" % ax)
    nt = int(rng.choice([1, 3, 9, 17, 33, 41, 81]))
    k = np.abs(rng.standard_normal(nt)) + 0.05
    if it % 3 == 0 and nt > 1: k[rng.integers(0, nt)] = 0.0
    cmp(ops.spectral_conv_f64(dd, k, mask=spec).get(), O.spectral_smooth(d, inc, k), 1e-13, tag + " sconv%d" % nt)
    ky = int(rng.choice([3, 9, 17, 29, 51])); g = np.exp(-0.5 * (np.arange(-(ky // 2), ky // 2 + 1) / (ky / 6.0)) ** 2)
    k2 = np.outer(g, g)
    small, sinc = d[:min(nz, 3)], (None if inc is None else inc[:min(nz, 3)])
    sspec = None if spec is None else (ops.MaskSpec(_lib.MASK_ARRAY, array=dev(sinc.astype(np.uint8))) if kind == 1 else spec)
    got = ops.spatial_conv_f64(dev(small), k2, mask=sspec)
    assert got.dtype == np.float64
    cmp(got.get(), O.spatial_smooth(small, sinc, k2), 1e-13, tag + " spconv%d" % ky)
    kk = int(rng.choice([5, 9]))
    yy, xx = np.mgrid[-(kk // 2):kk // 2 + 1, -(kk // 2):kk // 2 + 1]
    kn = np.exp(-0.5 * (((xx + 0.5 * yy) / 2.0) ** 2 + (yy / 1.2) ** 2))
    cmp(ops.spatial_conv_f64(dev(small), kn, mask=sspec).get(), O.spatial_smooth(small, sinc, kn), 1e-13, tag + " nonsep%d" % kk)
    if nz >= 2:
        xin = np.arange(nz) * 2.0; xout = np.linspace(rng.uniform(-3, nz), rng.uniform(nz, 2 * nz + 3), int(rng.integers(2, 120)))
        lo, t, inv, _, _, fill = ops.lerp_plan(xin, xout)
        eo, _ = O.spectral_interpolate(d, inc, xin, xout)
        cmp(ops.spectral_lerp_f64(dd, lo, t, inv, fill, mask=spec).get(), eo, 1e-14, tag + " lerp")
    # order statistics: sorted float64 rays
    import warnings
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fz = O.filled(d, inc, np.nan)
        cmp(ops.percentile_axis0_f64(dd, 50.0, mask=spec).get(), np.nanmedian(fz, axis=0), 0.0, tag + " median")
        qq = float(rng.uniform(0, 100))
        cmp(ops.percentile_axis0_f64(dd, qq, mask=spec).get(), np.nanpercentile(fz, qq, axis=0), 1e-15, tag + " percentile %.1f" % qq)
        if ny > 1:
            ms1 = None if spec is None else spec.swap01()
            cmp(ops.percentile_axis0_f64(dd.swap01(), 50.0, mask=ms1).get(), np.nanmedian(fz, axis=1), 0.0, tag + " median along y")
        med = np.nanmedian(fz, axis=0)
        mad = 1.482602218505602 * np.nanmedian(np.abs(fz - med), axis=0)
        gm = ops.percentile_axis0_f64(dd, 50.0, mask=spec)
        cmp(ops.percentile_axis0_f64(dd, 50.0, mask=spec, center=gm, scale=1.482602218505602).get(), mad, 1e-15, tag + " mad_std")
        for cenf, stdf in (("median", "std"), ("mean", "std"), ("median", "mad_std"), ("mean", "mad_std")):
            mi = [1, 3, 5, None][int(rng.integers(0, 4))]
            sg = float(rng.uniform(1.5, 3.5))
            got = ops.sigma_clip_axis0_f64(dd, sigma=sg, maxiters=mi, cenfunc=cenf, stdfunc=stdf, mask=spec).get()
            exp = O.sigma_clip(d, (np.ones(d.shape, bool) if inc is None else inc) & ~np.isnan(d), sigma=sg, maxiters=mi, cenfunc=cenf, stdfunc=stdf, out_dtype=np.float64)
            bad = np.mean(np.isnan(got) != np.isnan(exp))
            same = ~np.isnan(got) & ~np.isnan(exp)
            okc = bad <= 2e-4 and np.array_equal(got[same], exp[same])
            if not okc: fails += 1
            print("%s %s sigma_clip %s / %s maxiters %s: NaN-pattern mismatch fraction %.1e" % ("ok  " if okc else "FAIL", tag, cenf, stdf, mi, bad), flush=True)
    nar = ops.narrow_f64(dd).get()
    if not np.array_equal(nar, d.astype(np.float32), equal_nan=True): fails += 1; print("FAIL narrow", tag)
    gi = ops.mask_include_f64(dd, spec).get().astype(bool)
    ei = (np.ones(d.shape, bool) if inc is None else (inc if kind == 1 else inc))
    if not np.array_equal(gi, ei): fails += 1; print("FAIL include", tag, int((gi != ei).sum()))
print("failures", fails)
