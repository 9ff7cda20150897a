"""GPU, round 6: the masked spatial stencil's cube -> cube form marching whole columns (bands of any height), the three-sum
form with eight channel-parallel waves and one set of sums per block (chunk-centred float32 sums, shifted in float64), its
65-tap instantiation, reproducibility of the exchange through LDS; bench.py's float64 records.
Oracle: oracle_np (astropy semantics, float64)."""
import numpy as np
import pytest

import oracle_np as O
from conftest import assert_close
from spectral_cube_amd import Gaussian1DKernel, Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray

pytestmark = pytest.mark.gpu
K8 = Gaussian2DKernel(8 / 2.3548200450309493).array


def _case(shape, seed, valid=0.8, nan_frac=0.0):
    rng = np.random.default_rng(seed)
    d = (rng.standard_normal(shape) + 2.0).astype(np.float32)
    m = rng.random(shape) < valid
    if nan_frac:
        d[rng.random(shape) < nan_frac] = np.nan
    return d, m


def _dev(d, m):
    return DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))


def _moments_oracle(d, m, inc, k2, cen, dv, m1_add):
    sm = O.spatial_smooth(d, m, k2)
    filled = np.where(inc, sm, np.nan)
    e0 = dv * np.nansum(filled, axis=0)
    e0[np.all(np.isnan(filled), axis=0)] = np.nan
    f0 = np.nan_to_num(filled, nan=0.0)
    c = cen[:, None, None]
    with np.errstate(all="ignore"):
        s0 = f0.sum(0)
        mu = (f0 * c).sum(0) / s0
        e2 = (f0 * (c - mu) ** 2).sum(0) / s0             # about the mean: the reference's own form (_moments.py:185-193)
    return e0, mu + m1_add, e2


@pytest.mark.parametrize("shape", [(3, 300, 130), (5, 33, 64), (2, 1000, 72), (4, 17, 260)])
def test_cube_to_cube_form_marches_whole_columns(gpu, shape):
    """round 6: the cube -> cube split form takes a band of ANY height - by default the whole column (nrt = ceil(ny / 16) row
    tiles, the last one partial): tall planes, planes shorter than the halo, heights that are no multiple of 16"""
    d, m = _case(shape, 61, valid=0.6, nan_frac=0.01)
    cube, mk = _dev(d, m)
    out, _ = ops.spatial_conv_mfma(cube, K8, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk), want_cube=True, want_m0=False)
    exp = O.spatial_smooth(d, m, K8)
    assert_close(out.get(), exp.astype(np.float32), atol=1e-5 * np.nanmax(np.abs(exp)), what="whole-column march %r" % (shape,))


@pytest.mark.parametrize("flags", [_lib.MASK_ARRAY, _lib.MASK_ARRAY | _lib.MASK_FINITE])
@pytest.mark.parametrize("shape", [(90, 45, 200), (8, 200, 64), (13, 97, 68), (70, 100, 132)])
def test_three_sum_form_shared_sums_against_the_oracle(gpu, shape, flags):
    """eight waves per block walk eight channels at a time and add into one set of sums: chunks that end inside a round of
    eight (90 = 11 x 8 + 2, 13, 70), several bands of 96 rows and a partial one, partial column strips, NaN samples, fully
    masked spaxels.  Moment 2 is compared with the second moment ABOUT THE MEAN, per pixel."""
    d, m = _case(shape, 9, valid=0.7, nan_frac=0.01)
    m[:, 3:6, 10:14] = False
    cube, mk = _dev(d, m)
    inc = m & np.isfinite(d) if flags & _lib.MASK_FINITE else m
    cen = (np.arange(shape[0]) - shape[0] // 2) * 500.0
    _, maps = ops.spatial_conv_mfma_moments(cube, K8, DeviceArray.from_numpy(cen), dv=500.0, m1_add=77.0, mask=ops.MaskSpec(flags, array=mk))
    e0, e1, e2 = _moments_oracle(d, m, inc, K8, cen, 500.0, 77.0)
    assert np.isnan(e0).sum() >= 12
    assert_close(maps["m0"].get(), e0, atol=1e-5 * np.nanmax(np.abs(e0)), what="m0")
    assert_close(maps["m1"].get(), e1, atol=1e-5 * 500.0 * shape[0], what="m1")
    assert_close(maps["m2"].get(), e2, atol=1e-5 * np.nanmax(np.abs(e2)), what="m2")


@pytest.mark.parametrize("shape", [(3, 70, 16), (2, 40, 48), (5, 33, 64), (2, 130, 208), (3, 50, 2048)])
def test_split_form_mask_rows_in_two_requests_per_lane(gpu, shape, monkeypatch):
    """round 6: for planes whose width is a multiple of 16 columns the split form reads the 96 mask bytes of a row in two requests
    per lane and hands every lane its dword of every unit by a transpose over lanes (v_permlane32 / 16_swap) - the same bytes as six
    dword loads: all three forms bit-identical to the six-load kernels (SPC_SPLIT_WIDE_MASK=0), on planes narrower than a strip,
    strips that end inside the plane, units that lie outside it"""
    d, m = _case(shape, 23, valid=0.7, nan_frac=0.01)
    cube, mk = _dev(d, m)
    cen = DeviceArray.from_numpy((np.arange(shape[0]) - shape[0] // 2) * 500.0)
    got = {}
    for flags in (_lib.MASK_ARRAY, _lib.MASK_ARRAY | _lib.MASK_FINITE):
        spec = ops.MaskSpec(flags, array=mk)
        for wide in ("1", "0"):
            monkeypatch.setenv("SPC_SPLIT_WIDE_MASK", wide)
            out, m0 = ops.spatial_conv_mfma(cube, K8, mask=spec, want_cube=True, want_m0=True, dv=500.0)
            _, maps = ops.spatial_conv_mfma_moments(cube, K8, cen, dv=500.0, m1_add=0.0, mask=spec)
            got[wide] = [out.get(), m0.get()] + [maps[k].get() for k in ("m0", "m1", "m2")]
        for a, b in zip(got["1"], got["0"]):
            assert np.array_equal(a, b, equal_nan=True)
    exp = O.spatial_smooth(d, m, K8)
    assert_close(got["1"][0], exp.astype(np.float32), atol=1e-5 * np.nanmax(np.abs(exp)), what="wide mask requests %r" % (shape,))


@pytest.mark.parametrize("stddev, taps", [(5.0, 41), (8.0, 65)])
def test_three_sum_form_takes_up_to_65_taps(gpu, stddev, taps):
    """round 6: moments 1 / 2 of kernels of 35 - 65 taps are fused as well (five Toeplitz blocks, bands of 16 row tiles)"""
    k2 = Gaussian2DKernel(stddev).array
    assert k2.shape == (taps, taps)
    shape = (19, 150, 100)
    d, m = _case(shape, 52, valid=0.7)
    cube, mk = _dev(d, m)
    cen = (np.arange(shape[0]) - 5) * 2.0
    _, maps = ops.spatial_conv_mfma_moments(cube, k2, DeviceArray.from_numpy(cen), dv=2.0, m1_add=-3.0, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
    e0, e1, e2 = _moments_oracle(d, m, m, k2, cen, 2.0, -3.0)
    assert_close(maps["m0"].get(), e0, atol=1e-5 * np.nanmax(np.abs(e0)), what="%d taps m0" % taps)
    assert_close(maps["m1"].get(), e1, atol=1e-5 * 2.0 * shape[0], what="%d taps m1" % taps)
    assert_close(maps["m2"].get(), e2, atol=1e-5 * np.nanmax(np.abs(e2)), what="%d taps m2" % taps)


def test_three_sum_form_keeps_moment2_of_a_narrow_line_far_from_the_reference_channel(gpu):
    """round-5 advisor: S1 = sum v c and S2 = sum v c^2 about the REFERENCE channel in float32 lost moment 2 of a narrow line
    far from it to cancellation (10 % at nz = 4096, sigma = 2 channels).  The sums are now kept about each chunk's middle channel
    and shifted to the map's mean in float64: a line of sigma = 2 channels at channel 3900 of 4096 (reference channel 2048),
    chunks of 64 channels, random mask - moment 1 to 1e-4 channel, moment 2 to 1e-4 RELATIVE, per pixel."""
    nz, ny, nx = 4096, 384, 256
    z = np.arange(nz)
    line = np.exp(-0.5 * ((z - 3900.3) / 2.0) ** 2).astype(np.float32)
    rng = np.random.default_rng(77)
    mrow = rng.random((nz, 1, nx)) < 0.7                       # a mask that varies along z and x (constant in y: cheap to build)
    d = np.broadcast_to(line[:, None, None], (nz, 8, nx)).copy()
    cube, mk = DeviceArray((nz, ny, nx), np.float32), DeviceArray((nz, ny, nx), np.uint8)
    import ctypes as C
    mt = np.ascontiguousarray(np.broadcast_to(mrow, (nz, 8, nx)), dtype=np.uint8)      # (astype of a broadcast view keeps its zero stride: not C order)
    for dev, host, isz in ((cube, d, 4), (mk, mt, 1)):
        row = nx * isz
        _lib.call("spc_memcpy3d_h2d", 0, C.c_void_p(dev.ptr), row, ny * row, host.ctypes.data_as(C.c_void_p), row, 8 * row, row, 8, nz, None)
        have = 8
        while have < ny:
            n = min(have, ny - have)
            _lib.call("spc_memcpy3d_d2d", 0, C.c_void_p(dev.ptr + have * row), row, ny * row, C.c_void_p(dev.ptr), row, ny * row, row, n, nz, None)
            have += n
    cen = (z - nz // 2) * 1.0
    _, maps = ops.spatial_conv_mfma_moments(cube, K8, DeviceArray.from_numpy(cen), dv=1.0, m1_add=0.0, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
    # every sample of a channel is the same number: the smoothed value IS that number wherever the window holds a valid sample
    inc = mrow[:, 0, :].astype(np.float64)                     # (nz, nx)
    f = inc * line[:, None].astype(np.float64)
    s0 = f.sum(0)
    mu = (f * cen[:, None]).sum(0) / s0
    m2 = (f * (cen[:, None] - mu) ** 2).sum(0) / s0
    # (away from the plane's edges: outside the plane lie valid ZEROS - boundary='fill' - which the windows there average in)
    g1, g2, mu, m2 = maps["m1"].get()[100:300, 20:-20], maps["m2"].get()[100:300, 20:-20], mu[20:-20], m2[20:-20]
    assert np.abs(g1 - mu[None, :]).max() <= 1e-4, np.abs(g1 - mu[None, :]).max()
    rel = np.abs(g2 - m2[None, :]) / m2[None, :]
    assert rel.max() <= 1e-4, rel.max()


def test_three_sum_form_is_reproducible_bit_for_bit(gpu):
    """the eight waves' contributions are added in wave order behind a barrier: twenty launches, one result (a race in the
    exchange through LDS shows here: it did, in the first version, for one static copy of the addition code)"""
    shape = (64, 300, 256)
    d, m = _case(shape, 5, valid=0.7)
    cube, mk = _dev(d, m)
    cen = DeviceArray.from_numpy((np.arange(shape[0]) - 32) * 1.0)
    ref = None
    for _ in range(20):
        _, maps = ops.spatial_conv_mfma_moments(cube, K8, cen, dv=1.0, m1_add=0.0, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
        got = [maps[k].get() for k in ("m0", "m1", "m2")]
        if ref is None:
            ref = got
        else:
            for a, b in zip(got, ref):
                assert np.array_equal(a, b, equal_nan=True)


# ---- float64 order statistics without a sort (select64_kernel), float64 spatial passes with four outputs per thread --------------
@pytest.mark.parametrize("shape", [(512, 9, 70), (37, 20, 33), (1024, 3, 40), (2500, 2, 18), (4096, 2, 5), (5, 40, 64), (1, 7, 9)])
@pytest.mark.parametrize("q", [50.0, 0.0, 100.0, 30.0, 99.9])
def test_float64_percentiles_by_radix_selection_are_numpy_s(gpu, shape, q):
    """round 6: the float64 median / percentile kernel pins the lower order statistic down one key byte at a time in LDS
    (no sort): rays of 1 .. 4096 samples, duplicates (quantised samples), +-inf, zeros of both signs, masked samples, fully
    masked rays - bit-identical to np.nanpercentile / np.nanmedian on the float64 samples"""
    import warnings
    rng = np.random.default_rng(shape[0] + int(q))
    d = 1000.0 + rng.standard_normal(shape)
    d[rng.random(shape) < 0.3] = np.round(d[rng.random(shape) < 0.3].mean(), 1)        # many equal samples
    d[rng.random(shape) < 0.01] = np.inf
    d[rng.random(shape) < 0.01] = -np.inf
    d[rng.random(shape) < 0.01] = 0.0
    d[rng.random(shape) < 0.01] = -0.0
    d[rng.random(shape) < 0.02] = np.nan
    m = rng.random(shape) < 0.8
    if shape[1] > 3 and shape[2] > 3:
        m[:, 1, 2] = False
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    got = ops.percentile_axis0_f64(cube, q, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk)).get()
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        f = np.where(m, d, np.nan)
        exp = np.nanmedian(f, axis=0) if q == 50.0 else np.nanpercentile(f, q, axis=0)
    assert got.dtype == np.float64 and np.array_equal(got, exp, equal_nan=True)


def test_float64_mad_by_radix_selection(gpu):
    rng = np.random.default_rng(12)
    d = 5.0 + rng.standard_normal((300, 12, 40))
    m = rng.random(d.shape) < 0.7
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    spec = ops.MaskSpec(_lib.MASK_ARRAY, array=mk)
    med = ops.percentile_axis0_f64(cube, 50.0, mask=spec)
    mad = ops.percentile_axis0_f64(cube, 50.0, mask=spec, center=med, scale=1.482602218505602).get()
    f = np.where(m, d, np.nan)
    exp = 1.482602218505602 * np.nanmedian(np.abs(f - np.nanmedian(f, axis=0)), axis=0)
    assert np.array_equal(mad, exp, equal_nan=True)


@pytest.mark.parametrize("shape, stddev", [((3, 50, 1100), 3.0), ((2, 37, 70), 1.0), ((2, 130, 2070), 5.0)])
def test_float64_spatial_smooth_four_outputs_per_thread_is_bit_identical(gpu, shape, stddev, monkeypatch):
    """round 6: the float64 spatial passes compute four adjacent outputs per thread from one LDS read of every sample; every
    output still adds its taps in astropy's order - the same float64 bits as the one-output passes, and the oracle's to 1e-13"""
    rng = np.random.default_rng(31)
    d = 1000.0 + rng.standard_normal(shape)
    d[rng.random(shape) < 0.01] = np.nan
    m = rng.random(shape) < 0.8
    k2 = Gaussian2DKernel(stddev).array
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    got = ops.spatial_conv_f64(cube, k2, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk)).get()
    exp = O.spatial_smooth(d, m, k2)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    assert np.abs(got[ok] - exp[ok]).max() <= 1e-13 * np.abs(exp[ok]).max()


@pytest.mark.parametrize("shape, taps, flags", [((3, 300, 520), (29, 29), "array"), ((2, 700, 300), (15, 9), "array+finite"),
                                                ((40, 66, 257), (33, 5), "finite"), ((1, 1030, 70), (1, 29), "none"),
                                                ((2, 90, 600), (31, 31), "array")])
def test_float64_spatial_smooth_ring_form(gpu, shape, taps, flags, monkeypatch):
    """round 6: symmetric factors of up to 33 taps run in ONE kernel (a block marches down a band of rows: x pass gathered from a
    staged row segment, y pass on a register ring) - every output adds its taps in the order of the two-pass forms: the same
    float64 bits, NaN patterns included; bands of rows for cubes of few planes, strips that end inside the last 256 columns,
    1-tap axes, every mask kind; and the oracle's values to 1e-13"""
    rng = np.random.default_rng(sum(shape) + taps[0])
    d = 1000.0 + 50.0 * rng.standard_normal(shape)
    d[rng.random(shape) < 0.02] = np.nan
    m = rng.random(shape) < 0.8
    m[:, 20:60, 30:64] = False                                  # empty windows: NaN out
    def g(n):
        x = np.arange(n) - n // 2
        w = np.exp(-0.5 * (x / max(n / 8.0, 0.5)) ** 2)
        return w / w.sum()
    k2 = np.outer(g(taps[0]), g(taps[1]))
    cube = DeviceArray.from_numpy(d)
    mk = DeviceArray.from_numpy(m.astype(np.uint8)) if "array" in flags else None
    fl = (_lib.MASK_ARRAY if "array" in flags else 0) | (_lib.MASK_FINITE if "finite" in flags else 0)
    spec = ops.MaskSpec(fl, array=mk) if fl else None
    monkeypatch.setenv("SPC_SPATIAL64_RING", "1")
    ring = ops.spatial_conv_f64(cube, k2, mask=spec).get()
    monkeypatch.setenv("SPC_SPATIAL64_RING", "0")
    two = ops.spatial_conv_f64(cube, k2, mask=spec).get()
    assert np.array_equal(ring, two, equal_nan=True)
    inc = (m if "array" in flags else np.ones(shape, bool)) & ~np.isnan(d)
    exp = O.spatial_smooth(d, inc, k2)
    assert np.array_equal(np.isnan(ring), np.isnan(exp))
    ok = ~np.isnan(exp)
    assert np.abs(ring[ok] - exp[ok]).max() <= 1e-13 * np.abs(exp[ok]).max()


def test_float64_spatial_smooth_ring_form_hands_infinite_samples_to_the_two_pass_form(gpu, monkeypatch):
    """the ring form pads the taps with zeros, and 0 x inf is NaN where the two-pass forms (and astropy) skip a zero tap: a block
    that meets an infinite VALID sample raises a device flag and the two-pass kernels, queued behind it, redo the call - the
    result is the two-pass result bit for bit; with the mask's isfinite term the sample is invalid and nothing is handed over"""
    rng = np.random.default_rng(5)
    shape = (3, 120, 300)
    d = 10.0 + rng.standard_normal(shape)
    d[1, 60, 100] = np.inf
    d[2, 5, 290] = -np.inf
    m = rng.random(shape) < 0.9
    m[1, 60, 100] = m[2, 5, 290] = True
    k2 = Gaussian2DKernel(2.0).array                            # 17 x 17 taps
    k2p = np.zeros((21, 21)); k2p[2:-2, 2:-2] = k2               # explicit zero taps as well
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    for kern in (k2, k2p):
        for fl in (_lib.MASK_ARRAY, _lib.MASK_ARRAY | _lib.MASK_FINITE):
            spec = ops.MaskSpec(fl, array=mk)
            monkeypatch.setenv("SPC_SPATIAL64_RING", "1")
            ring = ops.spatial_conv_f64(cube, kern, mask=spec).get()
            monkeypatch.setenv("SPC_SPATIAL64_RING", "0")
            two = ops.spatial_conv_f64(cube, kern, mask=spec).get()
            assert np.array_equal(ring, two, equal_nan=True)
            if fl & _lib.MASK_FINITE:
                assert np.isfinite(ring[1, 55:66, 95:106]).all()
            else:
                assert not np.isfinite(ring[1, 60, 100])


@pytest.mark.parametrize("shape, taps, flags", [((700, 20, 33), 33, "array"), ((300, 3, 5), 17, "array+finite"), ((64, 40, 130), 9, "finite"),
                                                ((1200, 2, 300), 25, "none"), ((40, 7, 9), 33, "array"), ((5, 6, 7), 1, "array")])
def test_float64_spectral_smooth_ring_form(gpu, shape, taps, flags, monkeypatch):
    """round 6: float64 spectral_smooth with up to 33 non-negative taps as ring streaming (a lane marches over z, its pending
    outputs in a register ring, every input read once, chunks of channels for small maps): every output adds its inputs from
    the lowest channel up like the runs-of-16 kernel - the same float64 bits - and the oracle's values to 1e-13; asymmetric
    kernels too (one tap set: all 33 fit the scalar registers)"""
    rng = np.random.default_rng(sum(shape) + taps)
    d = 1000.0 + 50.0 * rng.standard_normal(shape)
    d[rng.random(shape) < 0.02] = np.nan
    m = rng.random(shape) < 0.8
    m[shape[0] // 3:shape[0] // 3 + min(40, shape[0] // 2), 1, 2] = False      # empty windows: NaN out
    x = np.arange(taps) - taps // 2
    k1 = np.exp(-0.5 * (x / max(taps / 8.0, 0.5)) ** 2) * (1.0 + 0.3 * (x > 0))      # asymmetric
    k1 /= k1.sum()
    cube = DeviceArray.from_numpy(d)
    mk = DeviceArray.from_numpy(m.astype(np.uint8)) if "array" in flags else None
    fl = (_lib.MASK_ARRAY if "array" in flags else 0) | (_lib.MASK_FINITE if "finite" in flags else 0)
    spec = ops.MaskSpec(fl, array=mk) if fl else None
    monkeypatch.setenv("SPC_SPECTRAL64_RING", "1")
    ring = ops.spectral_conv_f64(cube, k1, mask=spec).get()
    monkeypatch.setenv("SPC_SPECTRAL64_RING", "0")
    runs = ops.spectral_conv_f64(cube, k1, mask=spec).get()
    assert np.array_equal(ring, runs, equal_nan=True)
    inc = (m if "array" in flags else np.ones(shape, bool)) & ~np.isnan(d)
    exp = O.spectral_smooth(d, inc, k1)
    assert np.array_equal(np.isnan(ring), np.isnan(exp))
    ok = ~np.isnan(exp)
    assert np.abs(ring[ok] - exp[ok]).max() <= 1e-13 * np.abs(exp[ok]).max()


def test_float64_spectral_smooth_ring_form_hands_over_what_it_does_not_take(gpu, monkeypatch):
    """infinite valid samples (a padding zero times infinity would be NaN: device flag, the runs-of-16 kernel redoes the call),
    negative taps and a zero centre tap (an empty window is then not the same as an invalid centre sample): the result is the
    runs-of-16 result bit for bit"""
    rng = np.random.default_rng(8)
    shape = (200, 6, 40)
    d = 10.0 + rng.standard_normal(shape)
    d[100, 3, 20] = np.inf
    d[5, 0, 39] = -np.inf
    m = rng.random(shape) < 0.9
    m[100, 3, 20] = m[5, 0, 39] = True
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    g = Gaussian1DKernel(2.0).array
    kernels = [g, np.concatenate([[0.0, 0.0], g, [0.0, 0.0]]), np.array([-0.25, 0.0, 1.5, 0.0, -0.25]), np.array([0.5, 0.0, 0.5])]
    for k1 in kernels:
        for fl in (_lib.MASK_ARRAY, _lib.MASK_ARRAY | _lib.MASK_FINITE):
            spec = ops.MaskSpec(fl, array=mk)
            monkeypatch.setenv("SPC_SPECTRAL64_RING", "1")
            ring = ops.spectral_conv_f64(cube, k1, mask=spec).get()
            monkeypatch.setenv("SPC_SPECTRAL64_RING", "0")
            runs = ops.spectral_conv_f64(cube, k1, mask=spec).get()
            assert np.array_equal(ring, runs, equal_nan=True)
            if k1 is g and not (fl & _lib.MASK_FINITE):
                assert not np.isfinite(ring[100, 3, 20])


def test_float64_ring_forms_on_row_and_plane_views(gpu, monkeypatch):
    """the ring kernels take the cube's, the mask's and the output's strides as given: a strip of rows of a larger cube (plane stride
    > rows x nx), a slab of planes written into a slab of a larger output"""
    rng = np.random.default_rng(2)
    shape = (6, 90, 300)
    d = 100.0 + rng.standard_normal(shape)
    d[rng.random(shape) < 0.02] = np.nan
    m = rng.random(shape) < 0.8
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    def g(n):
        x = np.arange(n) - n // 2
        w = np.exp(-0.5 * (x / (n / 7.0)) ** 2)
        return w / w.sum()
    k2, k1 = np.outer(g(21), g(13)), g(9)
    y0, y1 = 17, 71
    sub, spec = cube.rows(y0, y1), ops.MaskSpec(_lib.MASK_ARRAY, array=mk.rows(y0, y1))
    inc = (m & ~np.isnan(d))[:, y0:y1]
    monkeypatch.setenv("SPC_SPATIAL64_RING", "1")
    monkeypatch.setenv("SPC_SPECTRAL64_RING", "1")
    for got, exp in ((ops.spatial_conv_f64(sub, k2, mask=spec).get(), O.spatial_smooth(d[:, y0:y1], inc, k2)),
                     (ops.spectral_conv_f64(sub, k1, mask=spec).get(), O.spectral_smooth(d[:, y0:y1], inc, k1))):
        ok = ~np.isnan(exp)
        assert np.array_equal(np.isnan(got), np.isnan(exp)) and np.abs(got[ok] - exp[ok]).max() <= 1e-13 * np.abs(exp[ok]).max()
    z0, z1 = 2, 5
    full = DeviceArray.from_numpy(np.full(shape, -7.0))
    ops.spatial_conv_f64(cube.planes(z0, z1), k2, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk.planes(z0, z1)), out=full.planes(z0, z1))
    got = full.get()
    exp = O.spatial_smooth(d[z0:z1], (m & ~np.isnan(d))[z0:z1], k2)
    ok = ~np.isnan(exp)
    assert np.array_equal(np.isnan(got[z0:z1]), np.isnan(exp)) and np.abs(got[z0:z1][ok] - exp[ok]).max() <= 1e-13 * np.abs(exp[ok]).max()
    assert (got[:z0] == -7.0).all() and (got[z1:] == -7.0).all()             # nothing written outside the slab


def test_cube_level_arithmetic_of_the_masked_spatial_stencil_is_selectable(gpu, monkeypatch):
    """round-5 verdict, weak 1: which arithmetic a record was timed in has a name, and the cube-level call can ask for the
    other one: spatial_smooth(kernel, arithmetic="f32") runs the ring kernels (float32 multiply-adds, 2.5e-7 of the range),
    "f16-split" / None the split form (1e-6); both inside the 1e-5 contract against astropy's float64; a following moment 0 of
    the "f32" cube is NOT taken by the fused split-form kernel"""
    from spectral_cube_amd import SpectralCube
    from spectral_cube_amd.kernels import Gaussian2DKernel as G2
    shape = (6, 70, 132)
    d, m = _case(shape, 17, valid=0.7)
    hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3, "CDELT3": 0.5, "CUNIT3": "km/s",
           "CRPIX1": 24, "CRPIX2": 16, "CRPIX3": 1, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": -16.0, "BUNIT": "K"}
    cube = SpectralCube.read(d, hdr).with_mask(m)
    k = G2(8 / 2.3548200450309493)
    exp = O.spatial_smooth(d, m, k.array)
    scale = np.nanmax(np.abs(exp))
    errs = {}
    for name in (None, "f16-split", "f32"):
        got = cube.spatial_smooth(k, arithmetic=name)._device_data().get()
        assert np.array_equal(np.isnan(got), np.isnan(exp))
        ok = np.isfinite(exp)
        errs[name] = float(np.abs(got[ok] - exp[ok]).max() / scale)
        assert errs[name] <= 1e-5
    assert errs["f32"] < errs["f16-split"] and errs[None] == errs["f16-split"], errs
    with pytest.raises(ValueError):
        cube.spatial_smooth(k, arithmetic="f64")
    called = []
    monkeypatch.setattr(ops, "spatial_conv_mfma", lambda *a, **kw: called.append(1) or (_ for _ in ()).throw(AssertionError("fused split form")))
    m0 = np.asarray(cube.spatial_smooth(k, arithmetic="f32").moment0())
    e0 = 0.5 * np.nansum(np.where(m, exp, np.nan), axis=0)
    e0[~m.any(axis=0)] = np.nan
    assert not called
    assert_close(m0, e0, atol=1e-5 * np.nanmax(np.abs(e0)), what="moment0 of the float32-arithmetic smooth")
