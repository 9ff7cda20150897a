"""Scratch: launch-to-launch reproducibility of the hot-path kernels.  Every operator runs N times over the same
device-resident input; the outputs are read back and compared bit for bit (NaN == NaN) against the first launch.
A difference is a scheduling dependence - a race (round 4 found one in the one-kernel sigma clip this way)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spectral_cube_amd import ops, _lib, synth, Gaussian1DKernel, Gaussian2DKernel
from spectral_cube_amd.device import DeviceArray

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
shape = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (512, 1024, 1024)
nz, ny, nx = shape
rng = np.random.default_rng(5)
tile = rng.standard_normal((nz, 16, nx)).astype(np.float32)
tile[rng.random(tile.shape) < 0.01] = np.nan
tile[rng.random(tile.shape) < 0.02] *= 12.0
tm = (rng.random(tile.shape) < 0.8).astype(np.uint8)
tm[:, 3, 40:72] = 0
host = np.tile(tile, (1, ny // 16, 1))
hmask = np.tile(tm, (1, ny // 16, 1))
cube, mask = DeviceArray.from_numpy(host), DeviceArray.from_numpy(hmask)
del host, hmask
marr = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
mfin = ops.MaskSpec(_lib.MASK_FINITE)
mthr = ops.MaskSpec(_lib.MASK_ARRAY | _lib.MASK_GT | _lib.MASK_FINITE, array=mask, thr_lo=-1.5)
cen = DeviceArray.from_numpy((np.arange(nz) - nz // 2) * 500.0)
k1 = Gaussian1DKernel(4.0).array
k2 = Gaussian2DKernel(8 / 2.35482).array
grid_in = np.arange(nz, dtype=np.float64)
lo, t, inv_dx, _, _, fill = ops.lerp_plan(grid_in, np.linspace(0.2, nz - 1.2, 2 * nz))
th = np.deg2rad(30.0)
yy, xx = np.mgrid[0:ny, 0:nx].astype(np.float64)
xs = np.cos(th) * (xx - nx / 2) - np.sin(th) * (yy - ny / 2) + nx / 2
ys = np.sin(th) * (xx - nx / 2) + np.cos(th) * (yy - ny / 2) + ny / 2


def maps(d):
    return [d[k].get() for k in sorted(d) if not k.startswith("_")]


OPS = {
    "moments (uint8 mask, + argmax)": lambda: maps(ops.moments(cube, cen, mask=marr, want=("m0", "m1", "m2", "argmax", "nvalid"))),
    "moments (isfinite)": lambda: maps(ops.moments(cube, cen, mask=mfin)),
    "statistics (mask + threshold)": lambda: [np.array([ops.stats_global(cube, mask=mthr)[k] for k in ("npts", "min", "max", "sum", "sumsq")])],
    "stats_axis 0 (uint8 mask)": lambda: maps(ops.stats_axis(cube, 0, mask=marr)),
    "spectral_smooth 33 taps (uint8 mask)": lambda: [ops.spectral_conv(cube, k1, mask=marr).get()],
    "spectral_smooth 33 taps (isfinite: fast pass + dirty tiles)": lambda: [ops.spectral_conv(cube, k1, mask=mfin).get()],
    "spectral_smooth -> moments, fused (uint8 mask)": lambda: maps(ops.spectral_conv_moments(cube, k1, cen, mask=marr)),
    "spatial_smooth 29 x 29 (uint8 mask)": lambda: [ops.spatial_conv(cube, k2, mask=marr).get()],
    "spatial_smooth 29 x 29 (isfinite: fast pass + dirty tiles)": lambda: [ops.spatial_conv(cube, k2, mask=mfin).get()],
    "spatial_smooth -> moment0 on the matrix cores (uint8 mask)": lambda: [ops.spatial_conv_mfma(cube, k2, mask=marr, want_cube=False, want_m0=True)[1].get()],
    "spatial_smooth on the matrix cores, cube (uint8 mask)": lambda: [ops.spatial_conv_mfma(cube, k2, mask=marr)[0].get()],
    "median (uint8 mask)": lambda: [ops.percentile_axis0(cube, 50.0, mask=marr).get()],
    "percentile 30 (isfinite)": lambda: [ops.percentile_axis0(cube, 30.0, mask=mfin).get()],
    "sigma clip (uint8 mask)": lambda: [ops.sigma_clip_axis0(cube, 3.0, mask=marr).get()],
    "sigma clip mad_std (isfinite)": lambda: [ops.sigma_clip_axis0(cube, 3.0, mask=mfin, stdfunc="mad_std").get()],
    "sigma clip mean / std (uint8 mask)": lambda: [ops.sigma_clip_axis0(cube, 2.5, mask=marr, cenfunc="mean").get()],
    "spectral_interpolate x 2 (uint8 mask)": lambda: [ops.spectral_lerp(cube, lo, t, inv_dx, fill=fill, mask=marr).get()],
    "reproject bilinear 30 deg (uint8 mask)": lambda: [ops.resample_bilinear(cube, xs, ys, mask=marr, want_footprint=False)[0].get()],
    "spatial_smooth -> moments 0+1+2, three sums (uint8 mask)": lambda: maps(ops.spatial_conv_mfma_moments(cube, k2, cen, dv=500.0, m1_add=0.0, mask=marr)[1]),
    "float64 spatial_smooth, ring form (uint8 mask)": lambda: [ops.spatial_conv_f64(cube64, k2, mask=marr64).get()],
    "float64 spectral_smooth, ring form (uint8 mask)": lambda: [ops.spectral_conv_f64(cube64, k1, mask=marr64).get()],
    "float64 median (uint8 mask)": lambda: [ops.percentile_axis0_f64(cube64, 50.0, mask=marr64).get()],
    "float64 sigma clip (uint8 mask)": lambda: [ops.sigma_clip_axis0_f64(cube64, sigma=3.0, mask=marr64).get()],
    "float64 statistics (uint8 mask)": lambda: [np.array([ops.stats_global_f64(cube64, mask=marr64)[k] for k in ("npts", "min", "max", "sum", "sumsq")])],
}
# (round 6) a float64 cube of a quarter of the rows for the float64 operators
n64 = max(16, ny // 4 // 16 * 16)
cube64 = DeviceArray.from_numpy(np.tile(1000.0 + tile.astype(np.float64), (1, n64 // 16, 1)))
mask64 = DeviceArray.from_numpy(np.tile(tm, (1, n64 // 16, 1)))
marr64 = ops.MaskSpec(_lib.MASK_ARRAY, array=mask64)
only = os.environ.get("STRESS_ONLY")
total_bad = 0
for name, fn in OPS.items():
    if only and not any(o in name for o in only.split(",")):
        continue
    try:
        ref = fn()
    except Exception as exc:                                  # an operator that refuses the case is not a reproducibility finding
        print("%-64s skipped: %s" % (name, str(exc)[:80]), flush=True)
        continue
    bad = 0
    for it in range(1, N):
        got = fn()
        same = all(np.array_equal(a, b, equal_nan=True) if np.asarray(a).dtype.kind == "f" else np.array_equal(a, b) for a, b in zip(got, ref))
        bad += 0 if same else 1
        del got
    total_bad += bad
    print("%-64s %d launches, %d differ from the first" % (name, N, bad), flush=True)
    del ref
print("differing launches in total: %d" % total_bad)
