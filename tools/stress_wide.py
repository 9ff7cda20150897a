"""randomised differential runs of the float64 stencils (python tools/stress_wide.py [rounds] [seed], on a GPU box): random shapes
(rows / columns / channels around the ring kernels' strip, band and chunk edges), tap counts 1 .. 41, symmetric and asymmetric
kernels, zero / negative taps now and then, mask kinds (array, isfinite, thresholds), NaN and infinite samples - the ring forms
against the forms they replace (bit for bit, NaN patterns included) and against the float64 oracle (1e-12 of the largest value)"""
import os, sys, warnings
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "oracle"))
import numpy as np
import oracle_np as O
from spectral_cube_amd import ops, _lib
from spectral_cube_amd.device import DeviceArray

warnings.simplefilter("ignore")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0


def kernel1d(n, kind):
    x = np.arange(n) - n // 2
    k = np.exp(-0.5 * (x / max(n / 7.0, 0.4)) ** 2)
    if kind == "asym":
        k = k * (1.0 + 0.4 * (x > 0))
    elif kind == "zeros" and n >= 5:
        k[1] = 0.0; k[-2] = 0.0
    elif kind == "neg" and n >= 3:
        k[0] = -0.1 * k[n // 2]; k[-1] = k[0]
    return k / k.sum()


def compare(tag, ring, other, exp):
    global fails
    bad = []
    if not np.array_equal(ring, other, equal_nan=True):
        bad.append("ring != replaced form (%d voxels)" % int((~((ring == other) | (np.isnan(ring) & np.isnan(other)))).sum()))
    fin = np.isfinite(exp)
    if not np.array_equal(np.isnan(ring), np.isnan(exp)) or not np.array_equal(np.isinf(ring), np.isinf(exp)):
        bad.append("NaN / inf pattern differs from the oracle")
    elif fin.any() and np.abs(ring[fin] - exp[fin]).max() > 1e-12 * np.abs(exp[fin]).max():
        bad.append("values differ from the oracle by %.3g" % (np.abs(ring[fin] - exp[fin]).max() / np.abs(exp[fin]).max()))
    if bad:
        fails += 1
        print("FAIL", tag, "; ".join(bad), flush=True)


for it in range(rounds):
    spatial = it % 2 == 0
    if spatial:
        nz = int(rng.integers(1, 5))
        ny = int(rng.choice([1, 3, 17, 40, 63, 64, 65, 130, 300]))
        nx = int(rng.choice([1, 5, 31, 33, 255, 256, 257, 300, 520]))
    else:
        nz = int(rng.choice([1, 2, 16, 33, 67, 131, 140, 300, 700]))
        ny = int(rng.integers(1, 7))
        nx = int(rng.choice([1, 7, 40, 129, 256, 300]))
    shape = (nz, ny, nx)
    d = 100.0 + 30.0 * rng.standard_normal(shape)
    d[rng.random(shape) < 0.03] = np.nan
    with_inf = rng.random() < 0.25
    if with_inf:
        d[rng.random(shape) < 0.004] = np.inf
        d[rng.random(shape) < 0.004] = -np.inf
    kind = str(rng.choice(["array", "array+finite", "finite", "none", "gt", "array+lt"]))
    m = rng.random(shape) < rng.choice([0.3, 0.8, 0.97])
    flags, lo, hi = 0, 0.0, 0.0
    inc = ~np.isnan(d)
    if "array" in kind:
        flags |= _lib.MASK_ARRAY; inc &= m
    if "finite" in kind:
        flags |= _lib.MASK_FINITE; inc &= np.isfinite(d)
    if "gt" in kind:
        flags |= _lib.MASK_GT; lo = 70.0; inc &= d > lo
    if "lt" in kind:
        flags |= _lib.MASK_LT; hi = 140.0; inc &= d < hi
    mk = DeviceArray.from_numpy(m.astype(np.uint8)) if "array" in kind else None
    spec = ops.MaskSpec(flags, lo, hi, mk) if flags else None
    cube = DeviceArray.from_numpy(d)
    tk = str(rng.choice(["sym", "sym", "asym", "zeros", "neg"]))
    if spatial:
        ty, tx = int(rng.choice([1, 3, 9, 17, 19, 29, 33, 41])), int(rng.choice([1, 5, 15, 17, 29, 33]))
        k2 = np.outer(kernel1d(ty, tk), kernel1d(tx, "sym" if tk == "asym" else tk))
        tag = "it%d spatial %s taps %dx%d %s mask %s inf %d" % (it, shape, ty, tx, tk, kind, with_inf)
        os.environ["SPC_SPATIAL64_RING"] = "1"
        ring = ops.spatial_conv_f64(cube, k2, mask=spec).get()
        os.environ["SPC_SPATIAL64_RING"] = "0"
        other = ops.spatial_conv_f64(cube, k2, mask=spec).get()
        exp = O.spatial_smooth(d, inc, k2)
    else:
        t1 = int(rng.choice([1, 3, 9, 17, 19, 25, 33, 41]))
        k1 = kernel1d(t1, tk)
        tag = "it%d spectral %s taps %d %s mask %s inf %d" % (it, shape, t1, tk, kind, with_inf)
        os.environ["SPC_SPECTRAL64_RING"] = "1"
        ring = ops.spectral_conv_f64(cube, k1, mask=spec).get()
        os.environ["SPC_SPECTRAL64_RING"] = "0"
        other = ops.spectral_conv_f64(cube, k1, mask=spec).get()
        exp = O.spectral_smooth(d, inc, k1)
    compare(tag, ring, other, exp)
print("rounds %d failures %d" % (rounds, fails))
