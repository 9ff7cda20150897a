"""Scratch: resample_bilinear_lerp against the oracle's two compositions and against the two-pass device form; C5 timing."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import oracle_np as O
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray, Event, synchronize
rng = np.random.default_rng(3)
dev = DeviceArray.from_numpy
fails = 0
def cmp(a, b, tol, what):
    global fails
    nanbad = int((np.isnan(a) != np.isnan(b)).sum())
    fin = np.isfinite(a) & np.isfinite(b)
    sc = np.max(np.abs(b[fin])) if fin.any() else 1.0
    err = np.max(np.abs(a[fin] - b[fin])) / sc if fin.any() else 0.0
    ok = nanbad == 0 and err <= tol
    if not ok: fails += 1
    print("%s %s: nan mismatches %d, max err / scale %.2e" % ("ok  " if ok else "FAIL", what, nanbad, err), flush=True)
for it in range(24 if 'time' not in sys.argv else 0):
    nz, ny, nx = int(rng.integers(2, 60)), int(rng.integers(2, 150)), int(rng.integers(2, 200))
    d = (rng.standard_normal((nz, ny, nx)) * 2 + 1).astype(np.float32)
    d[rng.random(d.shape) < rng.choice([0.0, 0.01, 0.2])] = np.nan
    kind = int(rng.integers(0, 3))
    if kind == 0: inc, spec = None, None
    elif kind == 1:
        inc = rng.random(d.shape) > 0.2; spec = ops.MaskSpec(_lib.MASK_ARRAY, array=dev(inc.astype(np.uint8)))
    else:
        inc = (d > -1.0) & np.isfinite(d); spec = ops.MaskSpec(_lib.MASK_GT | _lib.MASK_FINITE, -1.0)
    nyo, nxo = int(rng.integers(1, 200)), int(rng.integers(1, 260))
    if it % 3 == 0: nyo, nxo = int(rng.integers(128, 200)), int(rng.integers(128, 260))       # 64 x 64 tiles
    yy, xx = np.mgrid[0:nyo, 0:nxo].astype(np.float64)
    a = rng.uniform(0, 2 * np.pi); sc = rng.uniform(0.5, 1.6)
    xs = sc * (np.cos(a) * xx - np.sin(a) * yy) + rng.uniform(-5, nx)
    ys = sc * (np.sin(a) * xx + np.cos(a) * yy) + rng.uniform(-5, ny)
    xin = np.cumsum(rng.uniform(0.5, 1.5, nz))
    nzo = int(rng.integers(2, 130))
    xout = np.linspace(rng.uniform(xin[0] - 3, xin[-1]), rng.uniform(xin[0], xin[-1] + 3), nzo)
    if xout[0] > xout[-1]: xout = xout[::-1].copy()
    if it % 5 == 0: xout = np.linspace(xin[0], xin[-1], nzo)             # exact hits at both ends
    lo, t, inv, _, _, fill = ops.lerp_plan(xin, xout)
    tag = "it%d %s -> (%d, %d, %d) mask%d order%d" % (it, (nz, ny, nx), nzo, nyo, nxo, kind, 1)
    if not ops.lerp_plan_is_foldable(lo):
        print("(plan not foldable)", tag); continue
    dd = dev(d)
    for order in (1, 0) if it % 4 == 0 else (1,):
        got, foot = ops.resample_bilinear_lerp(dd, xs, ys, lo, t, inv, mask=spec, order=order)
        got = got.get()
        # the device's two passes, both orders of the operators
        l1 = ops.spectral_lerp(dd, lo, t, inv, np.nan, mask=spec)
        two_a, foot_a = ops.resample_bilinear(l1, xs, ys, fill=np.nan, order=order)
        r1, foot_b = ops.resample_bilinear(dd, xs, ys, fill=np.nan, mask=spec, order=order)
        two_b = ops.spectral_lerp(r1, lo, t, inv, np.nan)
        cmp(got, two_a.get(), 2e-6, tag + " order%d vs lerp -> resample (device)" % order)
        cmp(got, two_b.get(), 1e-6, tag + " order%d vs resample -> lerp (device)" % order)
        if not np.array_equal(foot.get(), foot_a.get()): fails += 1; print("FAIL footprint", tag)
        if order == 1:
            ei, _ = O.spectral_interpolate(d, inc, xin, xout)
            eo, ef = O.resample_bilinear(ei, xs, ys)
            cmp(got, eo, 1e-5, tag + " vs oracle (interpolate, then resample)")
print("failures", fails)
if len(sys.argv) > 1 and sys.argv[1] == "time":
    from spectral_cube_amd import synth
    shape = (2048, 1024, 1024)
    cube = DeviceArray(shape, np.float32)
    tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2004, chunk_rows=8)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_gpu_fullsize import _replicate_rows
    _replicate_rows(cube, tile, 4)
    v = np.arange(2048) * 500.0 - 512000.0
    grid = np.linspace(v[0], v[-1], 4096)
    lo, t, inv, _, _, fill = ops.lerp_plan(v, grid)
    yy, xx = np.mgrid[0:1024, 0:1024].astype(np.float64)
    th = np.deg2rad(30.0)
    xs = np.cos(th) * (xx - 511.5) - np.sin(th) * (yy - 511.5) + 511.5
    ys = np.sin(th) * (xx - 511.5) + np.cos(th) * (yy - 511.5) + 511.5
    dxs, dys = dev(xs), dev(ys)
    out = DeviceArray((4096, 1024, 1024), np.float32)
    mid = DeviceArray((4096, 1024, 1024), np.float32)
    def timeit(fn, n=5):
        fn(); synchronize(); ts = []
        for _ in range(n):
            e0, e1 = Event(), Event(); e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_ms(e1))
        return float(np.median(ts))
    os.environ["SPC_BILINEAR_TILE"] = "32"
    for zc in (None, "128", "512"):
        if zc: os.environ["SPC_BILINEAR_ZCHUNK"] = zc
        else: os.environ.pop("SPC_BILINEAR_ZCHUNK", None)
        tf = timeit(lambda: ops.resample_bilinear_lerp(cube, dxs, dys, lo, t, inv, out=out, want_footprint=False))
        print("32 x 32 tiles, zchunk %s: fused interpolate + reproject %.3f ms" % (zc, tf), flush=True)
    os.environ.pop("SPC_BILINEAR_TILE")
    for zc in (None, "256", "1024", "2048"):
        if zc: os.environ["SPC_BILINEAR_ZCHUNK"] = zc
        else: os.environ.pop("SPC_BILINEAR_ZCHUNK", None)
        tf = timeit(lambda: ops.resample_bilinear_lerp(cube, dxs, dys, lo, t, inv, out=out, want_footprint=False))
        print("zchunk %s: fused interpolate + reproject %.3f ms = %.2f TB/s algorithmic (25.77 GB)" % (zc, tf, 25.77e9 / tf / 1e9), flush=True)
    os.environ.pop("SPC_BILINEAR_ZCHUNK", None)
    t1 = timeit(lambda: ops.spectral_lerp(cube, lo, t, inv, np.nan, out=mid))
    t2 = timeit(lambda: ops.resample_bilinear(mid, dxs, dys, out=out, want_footprint=False))
    print("two passes: spectral_lerp %.3f ms + resample_bilinear %.3f ms = %.3f ms" % (t1, t2, t1 + t2))
    a = ops.resample_bilinear_lerp(cube, dxs, dys, lo, t, inv, want_footprint=False)[0]
    ga = a.get()[::97, ::5, ::3]
    gb = out.get()[::97, ::5, ::3]
    cmp(ga, gb, 2e-6, "C5 fused vs two passes (sampled)")
