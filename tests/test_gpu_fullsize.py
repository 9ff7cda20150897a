"""GPU parity at BASELINE.json's FULL sizes through size-independent
properties: the full-size cube is a small seeded tile replicated on the device
(D2D doubling, seconds), so the expected full-size result follows from the
oracle's result on the tile by periodicity / linearity.  What this exercises
that the small parity tests cannot: 64-bit offsets, >4 GiB planes-of-cube
addressing, grid limits and multi-round scheduling at 1e9-1.7e10 voxels.
"""
import ctypes as C
import warnings

import numpy as np
import pytest

import oracle_np as O
from conftest import assert_close
from spectral_cube_amd import _lib, ops, synth, Gaussian1DKernel, Gaussian2DKernel
from spectral_cube_amd.device import DeviceArray, device_info

pytestmark = pytest.mark.gpu


def _need(nbytes):
    import gc
    gc.collect()            # cubes of earlier tests live in reference cycles (cube <-> LazyMask) until collected
    free = device_info(0)["free_mem"]
    if free < nbytes * 1.05:
        pytest.skip("needs %.0f GiB of HBM, %.0f GiB free" % (nbytes / 2**30, free / 2**30))


def _replicate_rows(dev, tile, itemsize):
    """dev: (nz, ny, nx) device array; tile: host (nz, ty, nx).  Fill rows by
    uploading the tile once and doubling it along y with strided D2D copies."""
    nz, ny, nx = dev.shape
    ty = tile.shape[1]
    assert ny % ty == 0
    row = nx * itemsize
    _lib.call("spc_memcpy3d_h2d", 0, C.c_void_p(dev.ptr), row, ny * row, tile.ctypes.data_as(C.c_void_p),
              row, ty * row, row, ty, nz, None)
    lib = _lib.load()
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    have = ty
    while have < ny:
        n = min(have, ny - have)
        # hipMemcpy2D: nz "rows" of n*row bytes with pitch ny*row
        rc = hip.hipMemcpy2D(C.c_void_p(dev.ptr + have * row), C.c_size_t(ny * row), C.c_void_p(dev.ptr),
                             C.c_size_t(ny * row), C.c_size_t(n * row), C.c_size_t(nz), 3)
        assert rc == 0, rc
        have += n


def _replicate_planes(dev, tile, itemsize):
    """fill (nz, ny, nx) with tile (tz, ny, nx) repeated along z (D2D doubling)."""
    nz, ny, nx = dev.shape
    tz = tile.shape[0]
    assert nz % tz == 0
    plane = ny * nx * itemsize
    _lib.call("spc_memcpy_h2d", 0, C.c_void_p(dev.ptr), tile.ctypes.data_as(C.c_void_p), tz * plane, None)
    have = tz
    while have < nz:
        n = min(have, nz - have)
        _lib.call("spc_memcpy_d2d", 0, C.c_void_p(dev.ptr + have * plane), C.c_void_p(dev.ptr), n * plane, None)
        have += n


def test_c2_moments_1024cubed_periodic_rows(gpu):
    """configs[1]: 1024^3 fp32 + uint8 mask, fused moment 0/1/2 + argmax."""
    shape, ty = (1024, 1024, 1024), 8
    _need(shape[0] * shape[1] * shape[2] * 5)
    tile = synth.gaussian_line_cube((shape[0], ty, shape[2]), synth.SEEDS["C2"], chunk_rows=ty)
    tile[:, 2, 16:24] = np.nan
    tmask = synth.boolean_mask(tile, synth.SEEDS["C2"])
    cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
    _replicate_rows(cube, tile, 4)
    _replicate_rows(mask, tmask, 1)
    v = synth.spectral_axis(shape[0])
    cen = v - v[0]
    cref = cen[shape[0] // 2]
    r = ops.moments(cube, DeviceArray.from_numpy(cen - cref), dv=500.0, m1_add=cref + v[0],
                    mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mask), want=("m0", "m1", "m2", "argmax"))
    e = O.moments012(tile, tmask.astype(bool), cen, 500.0, v[0])
    ea = O.argmax(tile, tmask.astype(bool))
    scales = (np.nanmax(np.abs(e[0])), 500.0 * shape[0], np.nanmax(np.abs(e[2])))
    for k, exp, sc in zip(("m0", "m1", "m2"), e, scales):
        got = r[k].get().reshape(shape[1] // ty, ty, shape[2])
        assert np.array_equal(np.isnan(got[0]), np.isnan(exp))
        assert np.array_equal(np.isnan(got), np.isnan(got[:1]).repeat(got.shape[0], 0))
        ok = np.isfinite(exp)
        assert np.abs(got[0][ok] - exp[ok]).max() <= 1e-5 * sc, k
        assert np.array_equal(got[:, ok], got[:1, ok].repeat(got.shape[0], 0)), k + " not periodic"
    am = r["argmax"].get().reshape(shape[1] // ty, ty, shape[2])
    assert np.array_equal(am[0], ea) and np.array_equal(am, am[:1].repeat(am.shape[0], 0))


def test_c3_spectral_smooth_moment1_2048cubed(gpu):
    """configs[2]: 2048^3 fp32, spectral_smooth(Gaussian sigma=4) then moment1 (fused)."""
    shape, ty = (2048, 2048, 2048), 2
    _need(shape[0] * shape[1] * shape[2] * 4)
    tile = synth.gaussian_line_cube((shape[0], ty, shape[2]), synth.SEEDS["C3"], chunk_rows=ty)
    tile[100:140, 0, 5] = np.nan                     # a NaN run longer than the kernel
    cube = DeviceArray(shape, np.float32)
    _replicate_rows(cube, tile, 4)
    k = Gaussian1DKernel(4).array
    assert k.size == 33
    v = synth.spectral_axis(shape[0])
    cen = v - v[0]
    cref = cen[shape[0] // 2]
    r = ops.spectral_conv_moments(cube, k, DeviceArray.from_numpy(cen - cref), dv=500.0, m1_add=cref + v[0],
                                  mask=ops.MaskSpec(_lib.MASK_FINITE | _lib.MASK_GT, 1.0), want=("m1",),
                                  cen_host=cen - cref)
    with np.errstate(invalid="ignore"):
        inc = np.isfinite(tile) & (tile > np.float32(1.0))     # data > 2*noise keeps the line, like C2's mask
    sm = O.spectral_smooth(tile, inc, k)
    exp = O.moment(sm, inc, 1, cen, 500.0, world0=v[0])
    got = r["m1"].get().reshape(shape[1] // ty, ty, shape[2])
    assert np.array_equal(np.isnan(got[0]), np.isnan(exp))
    # moment 1 = S1/S0 is ill-conditioned where the unmasked noise sums to S0 ~ 0
    # (SURVEY.md section 7 "parity metric"): compare the well-conditioned spaxels
    s0 = O.moment(sm, inc, 0, cen, 1.0)
    wc = np.abs(s0) > 5.0
    assert wc.mean() > 0.5
    assert np.abs(got[0][wc] - exp[wc]).max() <= 1e-5 * 500.0 * shape[0]
    assert np.array_equal(got, got[:1].repeat(got.shape[0], 0)), "not periodic"


def test_c4_spatial_smooth_moment0_4096x2048x2048(gpu):
    """configs[3]: 4096x2048x2048 fp32 + uint8 mask, spatial_smooth FWHM=8 px
    (29x29 Gaussian) then moment0.  Planes repeat with period 2, so the smoothed
    cube is periodic in z and moment0 = (nz/2) * sum of the two smoothed planes."""
    shape, tz = (4096, 2048, 2048), 2
    _need(shape[0] * shape[1] * shape[2] * 9)
    rng = np.random.default_rng(synth.SEEDS["C4"])
    tile = rng.standard_normal((tz,) + shape[1:], dtype=np.float32) + 2.0
    tmask = (rng.random((tz,) + shape[1:], dtype=np.float32) > 0.2).view(np.uint8)
    tmask[:, :8, :8] = 0
    cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
    _replicate_planes(cube, tile, 4)
    _replicate_planes(mask, tmask, 1)
    k2 = Gaussian2DKernel(8 / 2.3548200450309493).array
    assert k2.shape == (29, 29)
    mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
    sm = ops.spatial_conv(cube, k2, mask=mspec)
    cen = DeviceArray.from_numpy(np.zeros(shape[0]))
    # the smoothed cube keeps the ORIGINAL mask
    m0 = ops.moments(sm, cen, dv=500.0, mask=mspec, want=("m0",))["m0"].get()
    # oracle on the 2-plane tile (windowed to keep the direct 841-tap numpy sum cheap)
    win = (slice(None), slice(0, 96), slice(0, 160))
    pad = 14
    sub = (slice(None), slice(0, 96 + pad), slice(0, 160 + pad))
    exp_sm = O.spatial_smooth(tile[sub], tmask[sub].astype(bool), k2)[win]
    inc = tmask[win].astype(bool)
    exp_m0 = (shape[0] // tz) * 500.0 * np.where(inc, exp_sm, 0.0).sum(axis=0)
    exp_m0[~inc.any(axis=0)] = np.nan
    got_planes = np.empty((tz, 96, 160), np.float32)
    for z in range(tz):      # two smoothed planes from the far end of the cube (z = nz-2, nz-1)
        row = np.empty((96, 2048), np.float32)
        _lib.call("spc_memcpy_d2h", 0, row.ctypes.data_as(C.c_void_p),
                  C.c_void_p(sm.ptr + ((shape[0] - tz + z) * shape[1] * shape[2]) * 4), row.nbytes, None)
        got_planes[z] = row[:, :160]
    assert_close(got_planes, exp_sm, atol=1e-5 * np.nanmax(np.abs(exp_sm)), what="C4 smoothed planes")
    assert_close(m0[:96, :160], exp_m0, atol=1e-5 * np.nanmax(np.abs(exp_m0)), what="C4 moment0")
    # round 4: the same pipeline FUSED on the matrix cores (the smoothed cube never written): the whole 2048 x 2048 map against the
    # materialised one (the two share nothing but the inputs), the window against the oracle, and the size-independent properties of the
    # construction - planes repeat with period 2, so the map is (nz / 2) x the map of a 2-plane cube, and the map of the first half of
    # the channels is exactly half of it in every spaxel both halves see
    del sm
    _, m0f = ops.spatial_conv_mfma(cube, k2, mask=mspec, want_cube=False, want_m0=True, dv=500.0)
    m0f = m0f.get()
    assert_close(m0f[:96, :160], exp_m0, atol=1e-5 * np.nanmax(np.abs(exp_m0)), what="C4 fused moment0 vs oracle")
    assert_close(m0f, m0, atol=1e-5 * np.nanmax(np.abs(m0)), what="C4 fused vs materialised, whole map")
    _, m2p = ops.spatial_conv_mfma(cube.planes(0, 2), k2, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mask.planes(0, 2)), want_cube=False,
                                   want_m0=True, dv=500.0)
    assert_close(m0f, (shape[0] // 2) * m2p.get(), atol=1e-5 * np.nanmax(np.abs(m0)), what="C4 fused: periodicity in z")
    half = shape[0] // 2
    _, mh = ops.spatial_conv_mfma(cube.planes(0, half), k2, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mask.planes(0, half)), want_cube=False,
                                  want_m0=True, dv=500.0)
    assert_close(2.0 * mh.get(), m0f, atol=1e-5 * np.nanmax(np.abs(m0)), what="C4 fused: additivity over channel halves")


def test_c5_spectral_interpolate_2048_to_4096(gpu):
    """configs[4] (first half): 2048x1024x1024 -> 4096 channels."""
    shape, ty = (2048, 1024, 1024), 2
    _need(shape[0] * shape[1] * shape[2] * 12)
    tile = synth.gaussian_line_cube((shape[0], ty, shape[2]), synth.SEEDS["C5"], chunk_rows=ty)
    tile[700, 1, 9] = np.nan
    cube = DeviceArray(shape, np.float32)
    _replicate_rows(cube, tile, 4)
    v = synth.spectral_axis(shape[0])
    grid = np.linspace(v[0], v[-1], 4096)
    lo, t, inv, rin, rout, fill = ops.lerp_plan(v, grid)
    out = ops.spectral_lerp(cube, lo, t, inv, fill)
    exp, _ = O.spectral_interpolate(tile, None, v, grid)
    # compare rows 0..ty-1 and the last tile (rows ny-ty..ny-1) of a few channels + all channels of one row
    full = np.empty((4096, 1024), np.float32)          # row y = ny - 1 of every output channel
    lib = C.CDLL("libamdhip64.so")
    rowb = 1024 * 4
    rc = lib.hipMemcpy2D(full.ctypes.data_as(C.c_void_p), C.c_size_t(rowb),
                         C.c_void_p(out.ptr + (shape[1] - 1) * rowb), C.c_size_t(shape[1] * rowb),
                         C.c_size_t(rowb), C.c_size_t(4096), 2)
    assert rc == 0
    assert_close(full, exp[:, ty - 1, :], atol=1e-5 * np.nanmax(np.abs(exp)), what="C5 lerp last row")

def _rotated_map(ny, nx, deg):
    yy, xx = np.mgrid[0:ny, 0:nx].astype(np.float64)
    a = np.deg2rad(deg)
    xs = np.cos(a) * (xx - nx / 2) - np.sin(a) * (yy - ny / 2) + nx / 2
    ys = np.sin(a) * (xx - nx / 2) + np.cos(a) * (yy - ny / 2) + ny / 2
    return xs, ys


def _periodic_expected(small, xs, ys, ny, nx, rows, chans):
    """bilinear samples of a source that repeats *small* (tz, P, P) in z, y and x: inside the image and away
    from its border an output pixel only sees values of the tile, at the coordinates taken modulo P (the
    fractions are untouched: P is an integer) - the oracle runs on a 3 x 3 arrangement of the tile."""
    tz, P, _ = small.shape
    big = np.tile(small, (1, 3, 3))
    x, y = xs[rows], ys[rows]
    inner = (x >= 1) & (x <= nx - 2) & (y >= 1) & (y <= ny - 2)
    exp, _ = O.resample_bilinear(big[[c % tz for c in chans]], np.mod(x, P) + P, np.mod(y, P) + P)
    return exp, inner


def test_c5_reproject_4096x1024x1024_rotated_30deg(gpu):
    """configs[4] (second half) at full size: a 4096 x 1024 x 1024 cube resampled onto the same grid rotated
    by 30 degrees.  The source repeats a 64 x 64 x 4 tile in all three directions, so every output pixel
    whose footprint lies inside the image follows from the oracle on the tile (64-bit addressing of the
    resampler: 16 GiB in, 16 GiB out; the round-1 test set stopped at 150 x 170 images)."""
    shape, P, tz = (4096, 1024, 1024), 64, 4
    _need(shape[0] * shape[1] * shape[2] * 4 * 2.1)
    rng = np.random.default_rng(synth.SEEDS["C5"])
    small = rng.standard_normal((tz, P, P)).astype(np.float32)
    small[1, 10, 20] = np.nan
    tile = np.tile(small, (1, 1, shape[2] // P))                       # (tz, P, nx): periodic along x already
    cube = DeviceArray(shape, np.float32)
    plane = DeviceArray((tz, shape[1], shape[2]), np.float32)
    _replicate_rows(plane, tile, 4)
    _replicate_planes(cube, plane.get(), 4)
    del plane
    xs, ys = _rotated_map(shape[1], shape[2], 30.0)
    out, foot = ops.resample_bilinear(cube, xs, ys)
    f = foot.get().astype(bool)
    inside = (xs >= -0.5) & (xs <= shape[2] - 0.5) & (ys >= -0.5) & (ys <= shape[1] - 0.5)
    assert np.array_equal(f, inside)
    rows = [0, 1, 200, 511, 512, 777, 1023]
    chans = [0, 1, 2, 3, 2049, 4094, 4095]
    exp, inner = _periodic_expected(small, xs, ys, shape[1], shape[2], rows, chans)
    rowb = shape[2] * 4
    for ci, c in enumerate(chans):
        for ri, r in enumerate(rows):
            got = np.empty(shape[2], np.float32)
            _lib.call("spc_memcpy_d2h", 0, got.ctypes.data_as(C.c_void_p),
                      C.c_void_p(out.ptr + (c * shape[1] + r) * rowb), rowb, None)
            assert np.all(np.isnan(got[~inside[r]]))
            ok = inner[ri]
            e = exp[ci, ri]
            assert np.array_equal(np.isnan(got[ok]), np.isnan(e[ok])), (c, r)
            fin = ok & ~np.isnan(e)
            assert np.abs(got[fin] - e[fin]).max() <= 1e-5 * np.nanmax(np.abs(small)), (c, r)


def test_c5_interpolate_then_reproject_chain_fullsize(gpu):
    """configs[4] end to end at full size through the operator interface: 2048 x 1024 x 1024 ->
    spectral_interpolate to 4096 channels -> reproject onto the grid rotated by 30 degrees (celestial header:
    the channels are kept).  The cube repeats a 64 x 64 spatial tile and is LINEAR along the spectral axis
    (value = a(y, x) + b(y, x) * channel), so interpolation is exact and every interior output pixel is the
    oracle's bilinear sample of a(.) + b(.) * (its fractional source channel)."""
    from spectral_cube_amd import SpectralCube
    shape, P = (2048, 1024, 1024), 64
    _need(shape[0] * shape[1] * shape[2] * 4 * 5.2)
    rng = np.random.default_rng(synth.SEEDS["C5"] + 1)
    a0 = rng.standard_normal((P, P)).astype(np.float32)
    b0 = (rng.standard_normal((P, P)) * 1e-3).astype(np.float32)
    A, B = np.tile(a0, (shape[1] // P, shape[2] // P)), np.tile(b0, (shape[1] // P, shape[2] // P))
    cube = DeviceArray(shape, np.float32)
    for z0 in range(0, shape[0], 64):                                   # staged 64 planes at a time
        blk = (A[None] + B[None] * np.arange(z0, z0 + 64, dtype=np.float32)[:, None, None]).astype(np.float32)
        _lib.call("spc_memcpy_h2d", 0, C.c_void_p(cube.ptr + z0 * shape[1] * shape[2] * 4),
                  blk.ctypes.data_as(C.c_void_p), blk.nbytes, None)
    hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-4, "CDELT2": 1e-4,
           "CDELT3": 500.0, "CUNIT3": "m/s", "CRPIX1": 512.5, "CRPIX2": 512.5, "CRPIX3": 1, "CRVAL1": 40.0,
           "CRVAL2": 0.0, "CRVAL3": 0.0, "NAXIS": 3, "NAXIS1": 1024, "NAXIS2": 1024, "NAXIS3": 2048}
    sc = SpectralCube.from_device(cube, header=hdr)
    sc.allow_huge_operations = True
    v = sc.spectral_axis
    grid = np.linspace(v[0], v[-1], 4096)
    up = sc.spectral_interpolate(grid, suppress_smooth_warning=True)
    up.allow_huge_operations = True
    a = np.deg2rad(30.0)
    tgt = {k: hdr[k] for k in ("CTYPE1", "CTYPE2", "CDELT1", "CDELT2", "CRPIX1", "CRPIX2", "CRVAL1", "CRVAL2")}
    tgt.update(NAXIS=2, NAXIS1=1024, NAXIS2=1024, PC1_1=np.cos(a), PC1_2=-np.sin(a), PC2_1=np.sin(a), PC2_2=np.cos(a))
    with pytest.raises(ValueError, match="allow_huge_operations"):
        SpectralCube.from_device(cube, header=hdr).reproject(tgt)       # the guard of utils.py:41-75 (no work is queued)
    res = up.reproject(tgt)
    assert res.shape == (4096, 1024, 1024)
    np.testing.assert_allclose(res.spectral_axis, grid, rtol=1e-12, atol=1e-6)
    xs, ys = ops.wcs_pixel_map(sc.wcs, res.wcs, (1024, 1024))
    xs, ys = xs.get(), ys.get()
    assert np.array_equal(res._footprint, (xs >= -0.5) & (xs <= 1023.5) & (ys >= -0.5) & (ys <= 1023.5))
    dev = res._device_data()
    rows, chans = [3, 400, 512, 1020], [0, 1, 2047, 2048, 4095]
    zfrac = (grid - v[0]) / (v[1] - v[0])                              # fractional source channel of every output channel
    big = np.stack([np.tile(a0, (3, 3)), np.tile(b0, (3, 3))])
    rowb = 1024 * 4
    for r in rows:
        x, y = xs[r], ys[r]
        inner = (x >= 1) & (x <= 1022) & (y >= 1) & (y <= 1022)
        ab, _ = O.resample_bilinear(big, np.mod(x, P)[None] + P, np.mod(y, P)[None] + P)
        for c in chans:
            got = np.empty(1024, np.float32)
            _lib.call("spc_memcpy_d2h", 0, got.ctypes.data_as(C.c_void_p), C.c_void_p(dev.ptr + (c * 1024 + r) * rowb), rowb, None)
            e = ab[0, 0] + ab[1, 0] * zfrac[c]
            assert np.abs(got[inner] - e[inner]).max() <= 1e-5 * (np.abs(a0).max() + 2.1), (r, c, np.abs(got[inner] - e[inner]).max())


def test_c2_statistics_1024cubed_periodic_rows(gpu):
    """SURVEY.md section 8f rank 1 at configs[1] size: statistics() and the per-axis reductions
    of a 1024^3 cube + uint8 mask built from a replicated tile: counts / sums scale with the
    replication factor, extrema are the tile's, axis-0 maps are periodic in y."""
    shape, ty = (1024, 1024, 1024), 8
    _need(shape[0] * shape[1] * shape[2] * 5)
    tile = synth.gaussian_line_cube((shape[0], ty, shape[2]), synth.SEEDS["C2"], chunk_rows=ty)
    tile[:, 2, 16:24] = np.nan
    tmask = synth.boolean_mask(tile, synth.SEEDS["C2"])
    cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
    _replicate_rows(cube, tile, 4)
    _replicate_rows(mask, tmask, 1)
    mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
    rep = shape[1] // ty
    e = O.statistics(tile, tmask.astype(bool))
    st = ops.stats_global(cube, mask=mspec)
    assert st["npts"] == e["npts"] * rep
    assert st["min"] == e["min"] and st["max"] == e["max"]
    assert st["sum"] == pytest.approx(e["sum"] * rep, rel=1e-10)
    assert st["sumsq"] == pytest.approx(e["sumsq"] * rep, rel=1e-10)
    r0 = ops.stats_axis(cube, 0, mask=mspec)
    e0 = O.reduce(tile, tmask.astype(bool), "sum", axis=0)
    got = r0["sum"].get()
    for y0 in (0, 512, shape[1] - ty):
        assert_close(got[y0:y0 + ty], e0, rtol=1e-12, what="axis-0 sum rows %d" % y0)
    assert_close(r0["max"].get()[:ty], O.reduce(tile, tmask.astype(bool), "max", axis=0), what="axis-0 max")
    # axis 1 / 2 totals must agree with the global numbers
    for ax in (1, 2):
        r = ops.stats_axis(cube, ax, mask=mspec, want=("count", "sum"))
        assert int(r["count"].get().astype(np.int64).sum()) == int(st["npts"])
        assert float(np.nansum(r["sum"].get())) == pytest.approx(st["sum"], rel=1e-10)


def test_c2_order_statistics_and_argextrema_1024cubed(gpu):
    """SURVEY.md section 8f rank 4 and row a8 at configs[1] size (1024^3 + uint8 mask, a tile replicated
    along y): the median of the whole cube equals the tile's (replication keeps every quantile) and splits
    the counts (#{x < m} <= n/2 >= #{x > m}, counted by the statistics kernel with threshold predicates);
    medians along z and x are periodic in y and equal the tile's, bit for bit; argmax / argmin along x are
    the tile's, along y they point into the first period (first index wins ties)."""
    shape, ty = (1024, 1024, 1024), 8
    _need(shape[0] * shape[1] * shape[2] * 5 * 2)
    tile = synth.gaussian_line_cube((shape[0], ty, shape[2]), synth.SEEDS["C2"], chunk_rows=ty)
    tile[:, 2, 16:24] = np.nan
    tmask = synth.boolean_mask(tile, synth.SEEDS["C2"]).astype(bool)
    cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
    _replicate_rows(cube, tile, 4)
    _replicate_rows(mask, tmask.astype(np.uint8), 1)
    mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
    filled = np.where(tmask, tile, np.nan).astype(np.float32)
    rep = shape[1] // ty
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # whole cube
        med = np.float32(ops.percentile_global(cube, 50.0, mask=mspec))
        assert med == np.nanmedian(filled)
        n = ops.stats_global(cube, mask=mspec)["npts"]
        below = ops.stats_global(cube, mask=ops.MaskSpec(_lib.MASK_ARRAY | _lib.MASK_LT, 0.0, float(med), mask))["npts"]
        above = ops.stats_global(cube, mask=ops.MaskSpec(_lib.MASK_ARRAY | _lib.MASK_GT, float(med), 0.0, mask))["npts"]
        assert below <= n / 2 and above <= n / 2 and n == np.count_nonzero(~np.isnan(filled)) * rep
        q90 = ops.percentile_global(cube, 90.0, mask=mspec)
        e90 = np.nanpercentile(np.tile(filled.astype(np.float64), (1, 4, 1)), 90.0)       # 4 periods: the quantile converges fast
        assert abs(q90 - e90) <= 1e-3 * abs(e90)
        # along z: periodic in y, equal to the tile's
        m0 = ops.percentile_axis0(cube, 50.0, mask=mspec).get().reshape(rep, ty, shape[2])
        e0 = np.nanmedian(filled, axis=0)
        assert np.array_equal(np.isnan(m0[0]), np.isnan(e0)) and np.array_equal(m0[0][~np.isnan(e0)], e0[~np.isnan(e0)])
        assert np.array_equal(m0[rep // 2], m0[0], equal_nan=True) and np.array_equal(m0[-1], m0[0], equal_nan=True)
        # along x (fill + transpose + selection): (nz, ny) periodic in y
        m2 = ops.percentile_axis0(ops.fill_masked_transposed(cube, mspec).swap01(), 50.0).get().reshape(shape[0], rep, ty)
        e2 = np.nanmedian(filled, axis=2)
        assert np.array_equal(m2[:, 0], e2, equal_nan=True) and np.array_equal(m2[:, -1], e2, equal_nan=True)
        # along y: the replicated ray has the tile ray's median
        m1 = ops.percentile_axis0(cube.swap01(), 50.0, mask=mspec.swap01()).get()
        e1 = np.nanmedian(filled, axis=1)
        assert np.array_equal(m1, e1, equal_nan=True)
    a2 = ops.argextrema_axis(cube, 2, mask=mspec)
    for key, fn in (("argmax", O.argmax), ("argmin", O.argmin)):
        got = a2[key].get().reshape(shape[0], rep, ty)
        exp = fn(tile, tmask, axis=2)
        assert np.array_equal(got[:, 0], exp) and np.array_equal(got[:, rep - 1], exp)
    a1 = ops.argextrema_axis(cube, 1, mask=mspec)
    assert np.array_equal(a1["argmax"].get(), O.argmax(tile, tmask, axis=1))          # first period wins the ties
    assert np.array_equal(a1["argmin"].get(), O.argmin(tile, tmask, axis=1))


def test_c2_sigma_clip_1024cubed_periodic_rows(gpu):
    """sigma_clip_spectrally at configs[1] size (1024^3 + uint8 mask, a tile replicated along y) through the one-kernel
    form: clipping is per ray, so the result is periodic in y and every period equals the oracle's clipped tile (the
    same samples survive, bit for bit; borderline samples at the float32-vs-float64 bounds allowed for: < 2e-5 of them);
    64-bit addressing of the clipped copy (4 GiB written)."""
    shape, ty = (1024, 1024, 1024), 8
    _need(shape[0] * shape[1] * shape[2] * (4 + 4 + 1) * 1.2)
    rng = np.random.default_rng(77)
    tile = rng.standard_normal((shape[0], ty, shape[2])).astype(np.float32)
    tile[rng.random(tile.shape) < 0.02] *= 15.0
    tile[:, 2, 16:24] = np.nan
    tmask = rng.random(tile.shape) < 0.9
    cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
    _replicate_rows(cube, tile, 4)
    _replicate_rows(mask, tmask.astype(np.uint8), 1)
    out = ops.sigma_clip_axis0(cube, sigma=3.0, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mask))
    exp = O.sigma_clip(tile, tmask & ~np.isnan(tile), 3.0)
    rep = shape[1] // ty
    whole = out.get()
    for period in (0, rep // 3, rep - 1):
        got = whole[:, period * ty:(period + 1) * ty]
        assert np.mean(np.isnan(got) != np.isnan(exp)) < 2e-5
        ok = ~np.isnan(got) & ~np.isnan(exp)
        assert np.array_equal(got[ok], exp[ok])
        assert np.array_equal(got, whole[:, :ty], equal_nan=True)


@pytest.mark.gpu
def test_sigma_clip_1024cubed_is_reproducible_from_launch_to_launch(gpu):
    """The one-kernel sigma clip keeps a block's rays resident across its iterations and reuses one set of shared words
    for every descent: a wave that entered the next iteration used to reset the word another wave was still reading
    (the upper middle sample of an even count), and one launch in four clipped one 8-row period differently
    (round 4, a stress script now in the history: 22 of 60 launches; 0 of 80 with the barrier).  Twelve launches over
    the same input must give the same samples: count / sum / sum of squares / extrema of the result and the per-ray
    counts and sums, period by period."""
    shape, ty = (1024, 1024, 1024), 8
    _need(shape[0] * shape[1] * shape[2] * (4 + 4 + 1) * 1.2)
    rng = np.random.default_rng(77)
    tile = rng.standard_normal((shape[0], ty, shape[2])).astype(np.float32)
    tile[rng.random(tile.shape) < 0.02] *= 15.0
    tile[:, 2, 16:24] = np.nan
    tmask = rng.random(tile.shape) < 0.9
    cube, mask = DeviceArray(shape, np.float32), DeviceArray(shape, np.uint8)
    _replicate_rows(cube, tile, 4)
    _replicate_rows(mask, tmask.astype(np.uint8), 1)
    ms = ops.MaskSpec(_lib.MASK_ARRAY, array=mask)
    ref = None
    for launch in range(12):
        out = ops.sigma_clip_axis0(cube, sigma=3.0, mask=ms)
        st = ops.stats_global(out)
        per_ray = ops.stats_axis(out, 0, want=("count", "sum"))
        cnt, s = per_ray["count"].get(), per_ray["sum"].get()
        del out
        key = tuple(st[k] for k in ("npts", "sum", "sumsq", "min", "max"))
        ref = key if ref is None else ref
        assert key == ref, "launch %d: %r != %r" % (launch, key, ref)
        cnt = cnt.reshape(shape[1] // ty, ty, shape[2])
        s = s.reshape(shape[1] // ty, ty, shape[2])
        assert np.array_equal(cnt, np.broadcast_to(cnt[0], cnt.shape)), "launch %d: counts differ between periods" % launch
        assert np.array_equal(s, np.broadcast_to(s[0], s.shape), equal_nan=True), "launch %d: sums differ between periods" % launch
