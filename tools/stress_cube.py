"""Scratch: randomised cubes through the SpectralCube-level interface (lazy smoothing, the algebraic smooth -> moment
shortcuts, order statistics along every axis, sigma clipping, statistics) against the oracle - the host-side logic that
tools/stress_random.py (which drives ops.* directly) does not touch.  python tools/stress_cube.py [nrounds] [seed]"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import oracle_np as O
from spectral_cube_amd import SpectralCube, Gaussian1DKernel, Gaussian2DKernel
warnings.simplefilter("ignore")
nround = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0


def close(a, b, tol, what, scale=None):
    global fails
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape:
        fails += 1; print("FAIL", what, "shape", a.shape, b.shape, flush=True); return
    bad = np.isnan(a) != np.isnan(b)
    fin = np.isfinite(a) & np.isfinite(b)
    sc = scale if scale is not None else (np.max(np.abs(b[fin])) if fin.any() else 1.0)
    err = np.max(np.abs(a[fin] - b[fin])) if fin.any() else 0.0
    if bad.any() or err > tol * max(sc, 1e-30):
        fails += 1; print("FAIL", what, "nan-mismatch", int(bad.sum()), "err", err, "scale", sc, flush=True)


for it in range(nround):
    nz, ny, nx = int(rng.integers(2, 70)), int(rng.integers(2, 50)), int(rng.integers(2, 200))
    d = (rng.standard_normal((nz, ny, nx)) * 2 + 3).astype(np.float32)
    nan_frac = float(rng.choice([0.0, 0.0, 0.02]))
    d[rng.random(d.shape) < nan_frac] = np.nan
    hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3, "CDELT3": float(rng.uniform(0.2, 2.0)),
           "CUNIT3": "km/s", "CRPIX1": nx / 2, "CRPIX2": ny / 2, "CRPIX3": 1, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": float(rng.uniform(-50, 50)), "BUNIT": "K"}
    cube = SpectralCube.read(d, hdr)
    kind = int(rng.integers(0, 3))
    inc = np.isfinite(d)
    if kind == 1:
        m = rng.random(d.shape) > 0.25
        cube = cube.with_mask(m); inc = inc & m
    elif kind == 2:
        cube = cube.with_mask(cube > 2.0); inc = inc & (np.nan_to_num(d, nan=-1e30) > 2.0)
    tag = "it%d %s kind%d nan%g" % (it, (nz, ny, nx), kind, nan_frac)
    fz = np.where(inc, d, np.nan).astype(np.float32)
    # moments along every axis, orders 0..3
    for axis in (0, 1, 2):
        cen, size = cube._pix_cen_axis(axis), cube._pix_size_slice(axis)
        for order in (0, 1, 2, 3):
            got = np.asarray(cube.moment(order=order, axis=axis))
            exp = O.moment(d, inc, order, cen, size, axis=axis, world0=cube.spectral_axis[0] if axis == 0 else None)
            e0 = O.moment(d, inc, 0, cen, size, axis=axis)
            wc = np.abs(e0) > 1e-2 * np.nanmax(np.abs(e0)) if np.isfinite(e0).any() else np.zeros(e0.shape, bool)
            span = float(np.nanmax(np.abs(cen)) + abs(cube.spectral_axis[0])) if order else None
            close(np.where(wc, got, 0), np.where(wc, exp, 0), 1e-5, tag + " moment%d ax%d" % (order, axis),
                  scale=None if order == 0 else span ** order)
    # lazy spectral / spatial smoothing followed by moments
    k1 = Gaussian1DKernel(float(rng.choice([0.8, 2.0, 4.0])))
    sm = O.spectral_smooth(d, inc, k1.array)
    cen0, size0 = cube._pix_cen_axis(0), cube._pix_size_slice(0)
    sc1 = cube.spectral_smooth(k1)
    close(np.asarray(sc1.moment0()), O.moment(sm, inc, 0, cen0, size0), 1e-5, tag + " spectral_smooth.moment0")
    close(sc1.filled_data[:] if hasattr(sc1.filled_data, "__getitem__") else sc1._device_data().get(), np.where(inc, sm, np.nan), 1e-5, tag + " spectral_smooth data")
    k2 = Gaussian2DKernel(float(rng.choice([0.7, 1.5])))
    if nz <= 24:
        sp = O.spatial_smooth(d, inc, k2.array)
        sc2 = cube.spatial_smooth(k2)
        e0 = O.moment(sp, inc, 0, cen0, size0)
        close(np.asarray(sc2.moment0()), e0, 1e-5, tag + " spatial_smooth.moment0")
        e1 = O.moment(sp, inc, 1, cen0, size0, world0=cube.spectral_axis[0])
        wc = np.abs(e0) > 1e-2 * np.nanmax(np.abs(e0)) if np.isfinite(e0).any() else np.zeros(e0.shape, bool)
        close(np.where(wc, np.asarray(cube.spatial_smooth(k2).moment1()), 0), np.where(wc, e1, 0), 1e-5, tag + " spatial_smooth.moment1",
              scale=float(np.nanmax(np.abs(cen0)) + abs(cube.spectral_axis[0])))
    # order statistics
    for axis in (0, 1, 2):
        close(np.asarray(cube.median(axis=axis)), np.nanmedian(fz, axis=axis), 0.0, tag + " median ax%d" % axis)
        q = float(rng.uniform(0, 100))
        close(np.asarray(cube.percentile(q, axis=axis)), np.nanpercentile(fz.astype(np.float64), q, axis=axis), 3e-6, tag + " pct ax%d" % axis)
    if np.isfinite(fz).any():
        if float(cube.median()) != float(np.nanmedian(fz)): fails += 1; print("FAIL", tag, "median()", flush=True)
    # statistics and reductions
    for op in ("sum", "mean", "max", "min", "std"):
        for axis in (None, 0, 1, 2):
            got = getattr(cube, op)(axis=axis)
            exp = O.reduce(d, inc, op, axis=axis)
            close(np.asarray(got), np.asarray(exp), 1e-6 if op in ("max", "min") else 2e-6, tag + " %s ax%s" % (op, axis))
    # spectral interpolation onto a random linear grid (finer / coarser, partly outside, sometimes reversed)
    ax = cube.spectral_axis
    lo_, hi_ = ax[0] - rng.uniform(0, 2) * abs(ax[1] - ax[0]), ax[-1] + rng.uniform(0, 2) * abs(ax[1] - ax[0])
    grid = np.linspace(lo_, hi_, int(rng.integers(2, 3 * nz)))
    if rng.random() < 0.3: grid = grid[::-1]
    gi = cube.spectral_interpolate(grid, suppress_smooth_warning=True)._device_data().get()
    ei = O.spectral_interpolate(d, inc, ax, grid)[0]
    close(gi, ei, 1e-5, tag + " spectral_interpolate")
    # sigma clipping
    thr = float(rng.uniform(1.5, 3.0))
    got = cube.sigma_clip_spectrally(thr)._device_data().get()
    exp = O.sigma_clip(d, inc, thr)
    if np.mean(np.isnan(got) != np.isnan(exp)) > 1e-3: fails += 1; print("FAIL", tag, "sigma_clip", float(np.mean(np.isnan(got) != np.isnan(exp))), flush=True)
print("rounds", nround, "failures", fails)
