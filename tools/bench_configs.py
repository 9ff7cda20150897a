"""Scratch/report: time every BASELINE.json config at FULL size on one MI355X
(device-resident, HIP events, median of N).  Cubes are seeded tiles replicated on
the device (tests/test_gpu_fullsize.py helpers), so the run takes seconds."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
from spectral_cube_amd import ops, _lib, synth, Gaussian1DKernel, Gaussian2DKernel
from spectral_cube_amd.device import DeviceArray, Event, synchronize
from test_gpu_fullsize import _replicate_rows, _replicate_planes


def med_ms(fn, n=7, warm=2):
    for _ in range(warm): fn()
    synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = Event(), Event()
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_ms(e1))
    return float(np.median(ts))


def row(name, vox, ms, bytes_per_vox):
    gbs = vox * bytes_per_vox / ms / 1e6
    print("%-58s %9.3f ms %10.0f Mvox/s %7.0f GB/s (%4.1f%% of 8 TB/s)" % (name, ms, vox / ms / 1e3, gbs, gbs / 80), flush=True)
    return dict(name=name, ms=ms, mvox_s=vox / ms / 1e3, algorithmic_GBps=gbs, frac_of_8TBps=gbs / 8000)


res = []
# ---------------- C2: 1024^3 + uint8 mask, moments
shape = (1024, 1024, 1024); vox = np.prod(shape, dtype=np.int64)
tile = synth.gaussian_line_cube((shape[0], 8, shape[2]), 2001, chunk_rows=8)
cube = DeviceArray(shape, np.float32); mask = DeviceArray(shape, np.uint8)
_replicate_rows(cube, tile, 4); _replicate_rows(mask, synth.boolean_mask(tile, 2001), 1)
cen = DeviceArray.from_numpy((np.arange(shape[0]) - shape[0] // 2) * 500.0)
out = {k: DeviceArray(shape[1:], np.float64) for k in ("m0", "m1", "m2")}
ws = DeviceArray((max(1, _lib.load().spc_moments_workspace_bytes(*shape)),), np.uint8)
ms = med_ms(lambda: ops.moments(cube, cen, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mask), out=out, workspace=ws))
res.append(row("C2 1024^3 u8 mask: fused moment0+1+2", vox, ms, 5))
ms = med_ms(lambda: ops.moments(cube, cen, mask=ops.MaskSpec(_lib.MASK_FINITE), out=out, workspace=ws))
res.append(row("C2 1024^3 isfinite predicate: fused moment0+1+2", vox, ms, 4))
outx = dict(out, argmax=DeviceArray(shape[1:], np.int64))
ms = med_ms(lambda: ops.moments(cube, cen, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mask), out=outx, workspace=ws, want=("m0", "m1", "m2", "argmax")))
res.append(row("C2 1024^3 u8 mask: moment0+1+2 + argmax", vox, ms, 5))
# statistics() and the nan-reductions, one pass each
mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
ms = med_ms(lambda: ops.stats_global(cube, mask=mspec))
res.append(row("C2 1024^3 u8 mask: statistics() (5 stats, one pass)", vox, ms, 5))
ms = med_ms(lambda: ops.stats_global(cube))
res.append(row("C2 1024^3 no mask: statistics()", vox, ms, 4))
so = None
for ax in (0, 1, 2):
    so = ops.stats_axis(cube, ax, mask=mspec)
    ms = med_ms(lambda: ops.stats_axis(cube, ax, mask=mspec, out=so))
    res.append(row("C2 1024^3 u8 mask: count/min/max/sum/sumsq along axis %d" % ax, vox, ms, 5))
r_ = ops.moments(cube, cen, mask=mspec, want=("mu", "s0"), workspace=ws)
ms = med_ms(lambda: ops.moment_order(cube, cen, 3, r_["mu"], r_["s0"], mask=mspec), n=3, warm=1)
res.append(row("C2 1024^3 u8 mask: moment order 3 (second pass)", vox, ms, 5))
for ax in (1, 2):
    cen2 = DeviceArray.from_numpy(np.tile((np.arange(shape[ax]) * 1.0)[:, None] if ax == 1 else (np.arange(shape[ax]) * 1.0)[None, :], (1, shape[2]) if ax == 1 else (shape[1], 1)))
    ms = med_ms(lambda: ops.moments_spatial(cube, cen2, ax, 1.0, mask=mspec), n=3, warm=1)
    res.append(row("C2 1024^3 u8 mask: moment0+1+2 along spatial axis %d" % ax, vox, ms, 5))
# order statistics along the spectral axis (33 streaming reads each)
om = DeviceArray(shape[1:], np.float32)
ms = med_ms(lambda: ops.percentile_axis0(cube, 50.0, mask=mspec, out=om), n=3, warm=1)
res.append(row("C2 1024^3 u8 mask: median along the spectral axis", vox, ms, 5))
ms = med_ms(lambda: ops.percentile_axis0(cube, 50.0, out=om), n=3, warm=1)
res.append(row("C2 1024^3 no mask: median along the spectral axis", vox, ms, 4))
cubeobj_mask = mspec
for ax, fn in ((1, lambda: ops.percentile_axis0(cube.swap01(), 50.0, mask=mspec.swap01())),
               (2, lambda: ops.percentile_axis2(cube, 50.0, mask=mspec))):
    ms = med_ms(fn, n=3, warm=1)
    res.append(row("C2 1024^3 u8 mask: median along spatial axis %d" % ax, vox, ms, 5))
ms = med_ms(lambda: ops.percentile_global(cube, 50.0, mask=mspec), n=3, warm=1)
res.append(row("C2 1024^3 u8 mask: median of the whole cube (axis=None)", vox, ms, 5))
ms = med_ms(lambda: ops.percentile_global(cube, 50.0), n=3, warm=1)
res.append(row("C2 1024^3 no mask: median of the whole cube (axis=None)", vox, ms, 4))
ms = med_ms(lambda: ops.argextrema_axis(cube, 1, mask=mspec), n=3, warm=1)
res.append(row("C2 1024^3 u8 mask: argmax+argmin along spatial axis 1", vox, ms, 5))
ms = med_ms(lambda: ops.argextrema_axis(cube, 2, mask=mspec), n=3, warm=1)
res.append(row("C2 1024^3 u8 mask: argmax+argmin along spatial axis 2", vox, ms, 5))
import time as _t
synchronize(); t0 = _t.perf_counter()
clipped = ops.sigma_clip_axis0(cube, sigma=3.0)
synchronize(); ms = (_t.perf_counter() - t0) * 1e3
res.append(row("C2 1024^3: sigma_clip_spectrally(3), astropy defaults (wall clock)", vox, ms, 8))
del clipped
del cube, mask
# ---------------- C3: 2048^3, spectral_smooth sigma=4 then moment1
shape = (2048, 2048, 2048); vox = np.prod(shape, dtype=np.int64)
tile = synth.gaussian_line_cube((shape[0], 2, shape[2]), 2002, chunk_rows=2)
cube = DeviceArray(shape, np.float32); _replicate_rows(cube, tile, 4)
cen_h = (np.arange(shape[0]) - shape[0] // 2) * 500.0
cen = DeviceArray.from_numpy(cen_h)
k = Gaussian1DKernel(4).array
o1 = {"m1": DeviceArray(shape[1:], np.float64)}
ms = med_ms(lambda: ops.spectral_conv_moments(cube, k, cen, out=o1, want=("m1",), cen_host=cen_h), n=5, warm=1)
res.append(row("C3 2048^3: spectral_smooth(33) -> moment1 FUSED", vox, ms, 4))
sm = DeviceArray(shape, np.float32)
ms_a = med_ms(lambda: ops.spectral_conv(cube, k, out=sm), n=5, warm=1)
res.append(row("C3 2048^3: spectral_smooth(33) materialised", vox, ms_a, 8))
ws = DeviceArray((max(1, _lib.load().spc_moments_workspace_bytes(*shape)),), np.uint8)
ms_b = med_ms(lambda: ops.moments(sm, cen, out=o1, want=("m1",), workspace=ws), n=5, warm=1)
res.append(row("C3 2048^3: moment1 of the smoothed cube", vox, ms_b, 4))
del cube, sm
# ---------------- C4: 4096x2048x2048 + u8 mask, spatial_smooth FWHM=8 + moment0
shape = (4096, 2048, 2048); vox = np.prod(shape, dtype=np.int64)
rng = np.random.default_rng(2003)
tile = rng.standard_normal((2,) + shape[1:], dtype=np.float32) + 2.0
cube = DeviceArray(shape, np.float32); _replicate_planes(cube, tile, 4)
k2 = Gaussian2DKernel(8 / 2.3548200450309493).array
sm = DeviceArray(shape, np.float32)
ms = med_ms(lambda: ops.spatial_conv(cube, k2, out=sm), n=3, warm=1)
res.append(row("C4 4096x2048x2048 no NaN: spatial_smooth 29x29 (fast path)", vox, ms, 8))
# config 3 as a pipeline through the cube API: spatial_smooth(FWHM=8) -> moment0, no invalid voxel:
# algebraic path (one pass over the unsmoothed cube + one 2048^2 map convolution); wall clock incl. host maps
import time as _time
from spectral_cube_amd import SpectralCube, synth as _synth
hdr = {"NAXIS": 3, "NAXIS1": shape[2], "NAXIS2": shape[1], "NAXIS3": shape[0], "CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN",
       "CTYPE3": "VRAD", "CRVAL3": 0.0, "CDELT3": 500.0, "CRPIX3": 1.0, "CUNIT3": "m/s", "CDELT1": -1e-4, "CDELT2": 1e-4,
       "CRPIX1": 1.0, "CRPIX2": 1.0, "CRVAL1": 10.0, "CRVAL2": 20.0, "BUNIT": "K"}
sc = SpectralCube.from_device(cube, header=hdr)
kobj = Gaussian2DKernel(8 / 2.3548200450309493)
for _ in range(2): sc.spatial_smooth(kobj).moment0()
synchronize(); t0 = _time.perf_counter()
for _ in range(3): m0map = sc.spatial_smooth(kobj).moment0()
synchronize(); ms = (_time.perf_counter() - t0) / 3 * 1e3
res.append(row("C4 pipeline spatial_smooth->moment0, all valid (algebraic, wall clock)", vox, ms, 4))
mask = DeviceArray(shape, np.uint8)
_replicate_planes(mask, (rng.random((2,) + shape[1:], dtype=np.float32) > 0.2).view(np.uint8), 1)
mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
ms = med_ms(lambda: ops.spatial_conv(cube, k2, mask=mspec, out=sm), n=3, warm=1)
res.append(row("C4 4096x2048x2048 u8 mask 80%: spatial_smooth 29x29", vox, ms, 9))
cen = DeviceArray.from_numpy(np.zeros(shape[0]))
o0 = {"m0": DeviceArray(shape[1:], np.float64)}
ws = DeviceArray((max(1, _lib.load().spc_moments_workspace_bytes(*shape)),), np.uint8)
ms = med_ms(lambda: ops.moments(sm, cen, mask=mspec, out=o0, want=("m0",), workspace=ws), n=3, warm=1)
res.append(row("C4 4096x2048x2048 u8 mask: moment0 of the smoothed cube", vox, ms, 5))
ms = med_ms(lambda: ops.moments(cube, cen, mask=mspec, out={k_: DeviceArray(shape[1:], np.float64) for k_ in ("m0", "m1", "m2")}, workspace=ws), n=3, warm=1)
res.append(row("C4 4096x2048x2048 u8 mask: fused moment0+1+2 (north-star)", vox, ms, 5))
del cube, sm, mask
# ---------------- C5: 2048x1024x1024 -> 4096 channels, reproject rotated 30 deg
shape = (2048, 1024, 1024); vox_in = np.prod(shape, dtype=np.int64)
tile = synth.gaussian_line_cube((shape[0], 2, shape[2]), 2004, chunk_rows=2)
cube = DeviceArray(shape, np.float32); _replicate_rows(cube, tile, 4)
v = synth.spectral_axis(shape[0]); grid = np.linspace(v[0], v[-1], 4096)
lo, t, inv, _, _, fill = ops.lerp_plan(v, grid)
out5 = DeviceArray((4096,) + shape[1:], np.float32)
ms = med_ms(lambda: ops.spectral_lerp(cube, lo, t, inv, fill, out=out5), n=5, warm=1)
res.append(row("C5 2048->4096 ch x 1024^2: spectral_interpolate (per OUT vox)", 2 * vox_in, ms, 6))
from spectral_cube_amd.wcs import SimpleWCS, reproject_pixel_map
hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CRVAL1": 150.0, "CRVAL2": 2.0, "CRPIX1": 512.5, "CRPIX2": 512.5,
       "CDELT1": -1 / 3600, "CDELT2": 1 / 3600, "NAXIS": 2}
c, s_ = np.cos(np.radians(30)), np.sin(np.radians(30))
w_in, w_out = SimpleWCS(hdr, naxis=2), SimpleWCS(dict(hdr, PC1_1=c, PC1_2=-s_, PC2_1=s_, PC2_2=c), naxis=2)
xs, ys = reproject_pixel_map(w_in, w_out, (1024, 1024))
del cube
ms = med_ms(lambda: ops.resample_bilinear(out5, xs, ys), n=3, warm=1)
res.append(row("C5 4096x1024^2: reproject bilinear to 30-deg rotated grid", 2 * vox_in, ms, 8))
print(json.dumps(res))
