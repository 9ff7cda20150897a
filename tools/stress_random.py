"""Scratch: randomised shapes / masks through every kernel family against the oracle (run on a GPU box:
python tools/stress_random.py [nrounds] [seed]).  Not part of the pytest suite."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import oracle_np as O
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray
warnings.simplefilter("ignore")
nround = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = DeviceArray.from_numpy
fails = 0

def close(a, b, tol, what):
    global fails
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    bad = np.isnan(a) != np.isnan(b)
    fin = np.isfinite(a) & np.isfinite(b)
    scale = np.max(np.abs(b[fin])) if fin.any() else 1.0
    err = np.max(np.abs(a[fin] - b[fin])) if fin.any() else 0.0
    infbad = (np.isinf(a) != np.isinf(b)) | (np.isinf(a) & np.isinf(b) & (np.sign(a) != np.sign(b)))
    if bad.any() or infbad.any() or err > tol * max(scale, 1e-30):
        fails += 1
        print("FAIL", what, "nan-mismatch", int(bad.sum()), "inf-mismatch", int(infbad.sum()), "err", err, "scale", scale, flush=True)

for it in range(nround):
    nz, ny, nx = int(rng.integers(1, 90)), int(rng.integers(1, 70)), int(rng.integers(1, 300))
    if rng.random() < 0.3: nx = int(rng.integers(1, 16)) * 4
    d = (rng.standard_normal((nz, ny, nx)) * 3 + 1).astype(np.float32)
    d[rng.random(d.shape) < rng.choice([0.0, 0.02, 0.3])] = np.nan
    kind = rng.integers(0, 3)
    if kind == 0: inc, spec = None, None
    elif kind == 1:
        inc = rng.random(d.shape) > 0.3
        spec = ops.MaskSpec(_lib.MASK_ARRAY, array=dev(inc.astype(np.uint8)))
    else:
        inc = (d > 0.5) & np.isfinite(d); spec = ops.MaskSpec(_lib.MASK_GT | _lib.MASK_FINITE, 0.5)
    tag = "it%d %s mask%d" % (it, (nz, ny, nx), kind)
    dd = dev(d)
    # moments + argmax
    cen = np.cumsum(rng.uniform(0.5, 1.5, nz)); cref = cen[nz // 2]
    r = ops.moments(dd, dev(cen - cref), dv=1.3, m1_add=cref + 10.0, mask=spec, want=("m0", "m1", "argmax", "nvalid"))
    e0, e1, e2 = O.moments012(d, inc, cen, 1.3, 10.0)
    close(r["m0"].get(), e0, 1e-5, tag + " m0")
    g1, x1 = r["m1"].get(), e1
    ok = np.abs(e0) > 1e-3 * np.nanmax(np.abs(e0)) if np.isfinite(e0).any() else np.zeros_like(e0, bool)
    close(np.where(ok, g1, 0), np.where(ok, x1, 0), 1e-5, tag + " m1")
    am = r["argmax"].get(); ea = O.argmax(d, inc)
    if not np.array_equal(am, ea): fails += 1; print("FAIL", tag, "argmax", int((am != ea).sum()), flush=True)
    # statistics
    st = ops.stats_global(dd, mask=spec); es = O.statistics(d, inc)
    if st["npts"] != es["npts"] or (es["npts"] and (st["min"] != es["min"] or st["max"] != es["max"] or abs(st["sum"] - es["sum"]) > 1e-9 * (abs(es["sum"]) + 1))):
        fails += 1; print("FAIL", tag, "stats_global", st, es, flush=True)
    for ax in (0, 1, 2):
        ra = ops.stats_axis(dd, ax, mask=spec, want=("sum", "max"))
        close(ra["sum"].get(), O.reduce(d, inc, "sum", axis=ax), 1e-10, tag + " sum ax%d" % ax)
        close(ra["max"].get(), O.reduce(d, inc, "max", axis=ax), 0.0, tag + " max ax%d" % ax)
    # order statistics
    close(ops.percentile_axis0(dd, 50.0, mask=spec).get(), O.median(d, inc), 0.0, tag + " median")
    q = float(rng.uniform(0, 100))
    close(ops.percentile_axis0(dd, q, mask=spec).get(), O.percentile(d.astype(np.float64), inc, q), 3e-6, tag + " pct")
    # order statistics along y / x / the whole cube; argmax / argmin along the spatial axes
    fz = O.filled(d, inc, np.nan).astype(np.float32)
    mspec = spec if spec is not None else ops.MaskSpec()
    with np.errstate(all="ignore"):
        close(ops.percentile_axis0(dd.swap01(), 50.0, mask=mspec.swap01()).get(), np.nanmedian(fz, axis=1), 0.0, tag + " median ax1")
        close(ops.percentile_axis0(ops.fill_masked_transposed(dd, spec).swap01(), 50.0).get(), np.nanmedian(fz, axis=2), 0.0, tag + " median ax2")
        gm = np.float32(ops.percentile_global(dd, 50.0, mask=spec)); em = np.nanmedian(fz)
        if not ((np.isnan(gm) and np.isnan(em)) or gm == em): fails += 1; print("FAIL", tag, "global median", gm, em, flush=True)
        gq = ops.percentile_global(dd, q, mask=spec); eq_ = np.nanpercentile(fz.astype(np.float64), q) if np.isfinite(fz).any() else np.nan
        if not ((np.isnan(gq) and np.isnan(eq_)) or abs(gq - eq_) <= 3e-6 * max(1.0, abs(eq_))): fails += 1; print("FAIL", tag, "global pct", q, gq, eq_, flush=True)
    for ax in (1, 2):
        ra = ops.argextrema_axis(dd, ax, mask=spec)
        if not np.array_equal(ra["argmax"].get(), O.argmax(d, inc, axis=ax)): fails += 1; print("FAIL", tag, "argmax ax%d" % ax, flush=True)
        if not np.array_equal(ra["argmin"].get(), O.argmin(d, inc, axis=ax)): fails += 1; print("FAIL", tag, "argmin ax%d" % ax, flush=True)
    # spectral smoothing (ring sizes + generic), fused moments
    nt = int(rng.choice([1, 3, 7, 9, 15, 33, 41]))
    k = np.abs(rng.standard_normal(nt)) + 0.05
    close(ops.spectral_conv(dd, k, mask=spec).get(), O.spectral_smooth(d, inc, k), 1e-5, tag + " sconv%d" % nt)
    if nt <= 33:
        sm = O.spectral_smooth(d, inc, k)
        f0 = O.moment(sm, inc, 0, cen, 1.3)
        rf = ops.spectral_conv_moments(dd, k, dev(cen - cref), dv=1.3, m1_add=cref + 10.0, mask=spec, want=("m0",), cen_host=cen - cref)
        close(rf["m0"].get(), f0, 1e-5, tag + " fused m0 taps%d" % nt)
        if os.environ.get("SPC_STRESS_DUMP") == str(it):
            np.savez("gpurun_out/stress_dump.npz", d=d, k=k, cen=cen, f0=f0, got=rf["m0"].get(), sm=sm)
    # spatial smoothing: separable and not
    ky = int(rng.choice([3, 9, 17, 29])); g = np.exp(-0.5 * (np.arange(-(ky // 2), ky // 2 + 1) / (ky / 6.0)) ** 2)
    k2 = np.outer(g, g)
    small = d[:min(nz, 3)]
    sinc = None if inc is None else inc[:min(nz, 3)]
    sspec = None if spec is None else (ops.MaskSpec(_lib.MASK_ARRAY, array=dev(sinc.astype(np.uint8))) if kind == 1 else spec)
    close(ops.spatial_conv(dev(small), k2, mask=sspec).get(), O.spatial_smooth(small, sinc, k2), 1e-5, tag + " spconv%d" % ky)
    kk = int(rng.choice([5, 9, 13]))
    yy, xx = np.mgrid[-(kk // 2):kk // 2 + 1, -(kk // 2):kk // 2 + 1]
    kn = np.exp(-0.5 * (((xx + 0.5 * yy) / 2.0) ** 2 + (yy / 1.2) ** 2))
    close(ops.spatial_conv(dev(small), kn, mask=sspec).get(), O.spatial_smooth(small, sinc, kn), 1e-5, tag + " nonsep%d" % kk)
    # lerp + bilinear
    if nz >= 2:
        xin = np.arange(nz) * 2.0; xout = np.linspace(rng.uniform(-3, nz), rng.uniform(nz, 2 * nz + 3), int(rng.integers(2, 120)))
        lo, t, inv, _, _, fill = ops.lerp_plan(xin, xout)
        eo, _ = O.spectral_interpolate(d, inc, xin, xout)
        close(ops.spectral_lerp(dd, lo, t, inv, fill, mask=spec).get(), eo, 1e-5, tag + " lerp")
    nyo, nxo = int(rng.integers(1, 80)), int(rng.integers(1, 150))
    yy, xx = np.mgrid[0:nyo, 0:nxo].astype(np.float64)
    a = rng.uniform(0, 2 * np.pi); sc = rng.uniform(0.4, 2.5)
    xs = sc * (np.cos(a) * xx - np.sin(a) * yy) + rng.uniform(-5, nx)
    ys = sc * (np.sin(a) * xx + np.cos(a) * yy) + rng.uniform(-5, ny)
    filled = O.filled(d, inc, np.nan)
    eb, ef = O.resample_bilinear(filled, xs, ys)
    ob, of = ops.resample_bilinear(dd, xs, ys, mask=spec, fill=np.nan)
    close(ob.get(), eb, 1e-5, tag + " bilinear")
    if not np.array_equal(of.get().astype(bool), ef[0]): fails += 1; print("FAIL", tag, "footprint", flush=True)
# tall, thin cubes: the register-resident selection / sigma-clip kernels at every (spaxels per block, keys per lane)
# combination, the z-split of the masked spectral stencil
for it in range(max(nround // 5, 2)):
    nz, ny, nx = int(rng.integers(90, 2100)), int(rng.integers(1, 4)), int(rng.integers(1, 45))
    d = (rng.standard_normal((nz, ny, nx)) * 2).astype(np.float32)
    d[rng.random(d.shape) < 0.03] *= 9.0
    d[rng.random(d.shape) < rng.choice([0.0, 0.02])] = np.nan
    if rng.random() < 0.5: d = np.round(d * 2) / 2                    # ties
    inc = rng.random(d.shape) > rng.choice([0.0, 0.3])
    spec = ops.MaskSpec(_lib.MASK_ARRAY, array=dev(inc.astype(np.uint8)))
    tag = "tall%d %s" % (it, (nz, ny, nx))
    dd = dev(d)
    fz = np.where(inc, d, np.nan).astype(np.float32)
    q = float(rng.choice([50.0, rng.uniform(0, 100)]))
    exp = np.nanpercentile(fz.astype(np.float64), q, axis=0) if q != 50.0 else np.nanmedian(fz, axis=0)
    close(ops.percentile_axis0(dd, q, mask=spec).get(), exp, 0.0 if q == 50.0 else 3e-6, tag + " pct%g" % q)
    kw = dict(sigma=float(rng.uniform(1.5, 3.5)), maxiters=[1, 3, 5, None][int(rng.integers(0, 4))], cenfunc=str(rng.choice(["median", "mean"])),
              stdfunc=str(rng.choice(["std", "mad_std"])))
    os.environ.pop("SPC_SIGMA_CLIP_FUSED", None)
    got = ops.sigma_clip_axis0(dd, mask=spec, **kw).get()
    os.environ["SPC_SIGMA_CLIP_FUSED"] = "0"
    ref = ops.sigma_clip_axis0(dd, mask=spec, **kw).get()
    os.environ.pop("SPC_SIGMA_CLIP_FUSED")
    if not np.array_equal(got, ref, equal_nan=True): fails += 1; print("FAIL", tag, "sigma_clip fused != loop", kw, int((np.isnan(got) != np.isnan(ref)).sum()), flush=True)
    eo = O.sigma_clip(d, inc & ~np.isnan(d), **kw)
    if np.mean(np.isnan(got) != np.isnan(eo)) > 5e-4: fails += 1; print("FAIL", tag, "sigma_clip vs oracle", kw, float(np.mean(np.isnan(got) != np.isnan(eo))), flush=True)
    nt = int(rng.choice([9, 17, 33]))
    k = np.abs(rng.standard_normal(nt)) + 0.05
    if rng.random() < 0.5: k = k + k[::-1]
    close(ops.spectral_conv(dd, k, mask=spec).get(), O.spectral_smooth(d, inc, k), 1e-5, tag + " sconv%d" % nt)

# round-5 kernels: the split form of the masked spatial stencil (3 / 5 Toeplitz blocks, cube -> cube, fused moment 0 and
# moments 0 / 1 / 2), the wide masked spectral rings (35 - 65 taps) and the packed rays of the sigma clip
def gauss(nt, sig):
    x = np.arange(nt) - nt // 2
    g = np.exp(-0.5 * (x / sig) ** 2)
    return g / g.sum()

for it in range(max(nround // 2, 3)):
    nz, ny, nx = int(rng.integers(1, 7)), int(rng.integers(1, 330)), int(rng.integers(1, 420))
    if rng.random() < 0.75: nx = 4 * int(rng.integers(1, 110))           # the split form wants 16-byte rows
    d = (rng.standard_normal((nz, ny, nx)) * float(rng.choice([1e-3, 1.0, 3e4])) + float(rng.choice([0.0, 2.0]))).astype(np.float32)
    if rng.random() < 0.5: d[rng.random(d.shape) < 0.01] = np.nan
    if rng.random() < 0.3: d[rng.random(d.shape) < 1e-3] *= 1e4
    dens = float(rng.choice([0.03, 0.5, 0.8, 1.0]))
    inc = rng.random(d.shape) < dens
    if rng.random() < 0.3: inc[:, : ny // 2] = False                     # dead regions: empty windows
    spec = ops.MaskSpec(_lib.MASK_ARRAY, array=dev(inc.astype(np.uint8)))
    inc_f = inc & np.isfinite(d)
    nt = int(rng.choice([3, 9, 17, 29, 31, 33, 35, 41, 49, 57, 65]))
    g = gauss(nt, nt / 6.0 + rng.uniform(0, 2)); k2 = np.outer(g, g)
    tag = "split%d %s taps%d dens%.2f" % (it, (nz, ny, nx), nt, dens)
    dd = dev(d)
    exp = O.spatial_smooth(d, inc, k2)
    close(ops.spatial_conv(dd, k2, mask=spec).get(), exp, 1e-5, tag + " spatial_conv")
    cen = np.cumsum(rng.uniform(0.5, 1.5, nz)); cref = cen[nz // 2]
    try:
        oc, m0 = ops.spatial_conv_mfma(dd, k2, mask=spec, want_cube=True, want_m0=True, dv=1.3)
        close(oc.get(), exp, 1e-5, tag + " mfma cube")
        scale = np.nanmax(np.abs(exp)) if np.isfinite(exp).any() else 1.0
        e0 = O.moment(exp, inc, 0, cen, 1.3)
        g0 = m0.get()
        if (np.isnan(g0) != np.isnan(e0)).any() or (np.isfinite(e0).any() and np.nanmax(np.abs(g0 - e0)) > 1e-5 * scale * 1.3 * nz):
            fails += 1; print("FAIL", tag, "fused m0", int((np.isnan(g0) != np.isnan(e0)).sum()), flush=True)
        _, mm = ops.spatial_conv_mfma_moments(dd, k2, dev(cen - cref), dv=1.3, m1_add=cref, mask=spec, want=("m0", "m1", "m2"))
        g0b = mm["m0"].get()
        if (np.isnan(g0b) != np.isnan(e0)).any() or (np.isfinite(e0).any() and np.nanmax(np.abs(g0b - e0)) > 1e-5 * scale * 1.3 * nz):
            fails += 1; print("FAIL", tag, "fused m012: m0", flush=True)
        e1 = O.moment(exp, inc, 1, cen, 1.3)
        okm = np.isfinite(e1) & (np.abs(e0) > 1e-2 * np.nanmax(np.abs(e0)))
        g1 = mm["m1"].get()
        if okm.any() and np.max(np.abs(g1[okm] - e1[okm])) > 1e-3 * (cen[-1] - cen[0] + 1.0):
            fails += 1; print("FAIL", tag, "fused m012: m1", float(np.max(np.abs(g1[okm] - e1[okm]))), flush=True)
    except _lib.HipUnsupported as e:
        print("(unsupported)", tag, str(e)[:80], flush=True)

for it in range(max(nround // 2, 3)):
    nz, ny, nx = int(rng.integers(40, 400)), int(rng.integers(1, 9)), int(rng.integers(1, 700))
    d = (rng.standard_normal((nz, ny, nx)) * 2 + float(rng.choice([0.0, 5.0]))).astype(np.float32)
    if rng.random() < 0.5: d[rng.random(d.shape) < 0.01] = np.nan
    dens = float(rng.choice([0.03, 0.5, 0.8, 1.0]))
    inc = rng.random(d.shape) < dens
    if rng.random() < 0.3: inc[nz // 3: nz // 3 + 80] = False
    kind = int(rng.integers(0, 3))
    if kind == 0: spec = ops.MaskSpec(_lib.MASK_ARRAY, array=dev(inc.astype(np.uint8)))
    elif kind == 1: inc = None; spec = None
    else: inc = (d > 0.5) & np.isfinite(d); spec = ops.MaskSpec(_lib.MASK_GT | _lib.MASK_FINITE, 0.5)
    nt = int(rng.choice([35, 37, 41, 47, 49, 51, 57, 63, 65, 67, 81]))
    k = gauss(nt, nt / 6.0 + rng.uniform(0, 3))
    if rng.random() < 0.15: k = np.abs(rng.standard_normal(nt)) + 0.05      # asymmetric: the runs-of-16 kernel
    tag = "wide%d %s taps%d kind%d" % (it, (nz, ny, nx), nt, kind)
    got = ops.spectral_conv(dev(d), k, mask=spec).get()
    exp64 = O.spectral_smooth(d, inc, k)
    close(got, exp64, 1e-6, tag + " sconv")

for it in range(max(nround // 2, 3)):
    nz, ny, nx = int(rng.integers(200, 2100)), int(rng.integers(1, 5)), int(rng.integers(1, 200))
    d = (rng.standard_normal((nz, ny, nx)) * 2).astype(np.float32)
    d[rng.random(d.shape) < 0.03] *= 9.0
    if rng.random() < 0.5: d = np.round(d * 2) / 2
    dens = float(rng.choice([0.0, 0.005, 0.03, 0.06, 0.12]))
    inc = rng.random(d.shape) < dens
    if rng.random() < 0.3: inc[:, 0, : max(nx // 8, 1)] = rng.random((nz, max(nx // 8, 1))) < 0.5   # a few dense rays among the sparse
    spec = ops.MaskSpec(_lib.MASK_ARRAY, array=dev(inc.astype(np.uint8)))
    tag = "packed%d %s dens%.3f" % (it, (nz, ny, nx), dens)
    dd = dev(d)
    kw = dict(sigma=float(rng.uniform(1.2, 3.5)), maxiters=[1, 3, 5, None][int(rng.integers(0, 4))], cenfunc=str(rng.choice(["median", "mean"])),
              stdfunc=str(rng.choice(["std", "mad_std"])))
    got = ops.sigma_clip_axis0(dd, mask=spec, **kw).get()
    os.environ["SPC_SIGMA_CLIP_FUSED"] = "0"
    ref = ops.sigma_clip_axis0(dd, mask=spec, **kw).get()
    os.environ.pop("SPC_SIGMA_CLIP_FUSED")
    nd = int((np.isnan(got) != np.isnan(ref)).sum())
    if nd > 2e-4 * got.size or not np.array_equal(got[~np.isnan(got) & ~np.isnan(ref)], ref[~np.isnan(got) & ~np.isnan(ref)]):
        fails += 1; print("FAIL", tag, "sigma_clip fused != loop", kw, nd, flush=True)
    eo = O.sigma_clip(d, inc & ~np.isnan(d), **kw)
    if np.mean(np.isnan(got) != np.isnan(eo)) > 5e-4: fails += 1; print("FAIL", tag, "sigma_clip vs oracle", kw, float(np.mean(np.isnan(got) != np.isnan(eo))), flush=True)
    fz = np.where(inc, d, np.nan).astype(np.float32)
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        close(ops.percentile_axis0(dd, 50.0, mask=spec).get(), np.nanmedian(fz, axis=0), 0.0, tag + " median")
print("rounds", nround, "failures", fails)
