"""GPU: what round 3 added - reprojection ACROSS celestial frames (the reference's own RA/DEC -> GLON/GLAT test),
spatial stencils at the 1e-5 contract, out-of-core streaming, the dask seam with pinned staging."""
import os
import warnings

import numpy as np
import pytest

import oracle_np as O
from conftest import assert_close, golden
from spectral_cube_amd import SpectralCube, SimpleWCS, _lib, ops
from spectral_cube_amd.device import DeviceArray

pytestmark = pytest.mark.gpu


def test_device_pixel_map_across_celestial_frames(gpu):
    """spc_wcs_pixel_map_f64 with the frame rotation against astropy.wcs + astropy.coordinates
    (tests/golden/wcs_frames.npz): ICRS / FK5(equinox) / Galactic pairs, 1e-9 pixel."""
    g = golden("wcs_frames.npz")
    for i in range(int(g["n"])):
        a, b = SimpleWCS(str(g["in%d" % i]), naxis=2), SimpleWCS(str(g["out%d" % i]), naxis=2)
        exs, eys = g["xs%d" % i], g["ys%d" % i]
        xs, ys = ops.wcs_pixel_map(a, b, exs.shape)
        xs, ys = xs.get(), ys.get()
        ok = np.isfinite(exs) & np.isfinite(eys)
        assert ok.any()
        assert np.abs(xs[ok] - exs[ok]).max() <= 1e-9 and np.abs(ys[ok] - eys[ok]).max() <= 1e-9, i
        assert np.all(xs[~ok] == -1e30)
    with pytest.raises(NotImplementedError):          # (FK4 is built since round 4; ecliptic headers relate to nothing else)
        ops.wcs_pixel_map(SimpleWCS(dict(SimpleWCS(str(g["in1"]), naxis=2).header, CTYPE1="ELON-TAN", CTYPE2="ELAT-TAN"), naxis=2),
                          SimpleWCS(str(g["out1"]), naxis=2), (4, 4))


@pytest.mark.parametrize("host_map", [False, True])
def test_reference_reproject_case_radec_to_galactic(gpu, host_map, monkeypatch):
    """tests/test_regrid.py:99-135 of the reference: the (4, 3, 2) cube on the RA/DEC-SIN header of
    tests/data/header_jybeam.hdr (EPOCH = 2000: FK5) reprojected onto GLON-SIN / GLAT-SIN at 134.37608, -31.939241,
    CRPIX 2, 2, NAXIS 4 x 5.  Round 2 produced an all-NaN map here (frames equated) and raised.  The reference asserts
    shape and WCS; the VALUES are pinned through astropy's cross-frame pixel map + scipy's trilinear call (fixture)."""
    g = golden("wcs_frames.npz")
    if host_map:
        monkeypatch.setenv("SPC_WCS_HOST_MAP", "1")
    d = g["adv_data"]
    cube = SpectralCube.read(d.astype(np.float32), str(g["adv_header"]))
    hdr_out = str(g["adv_header_out"])
    res = cube.reproject(hdr_out)
    assert res.shape == (d.shape[0], 5, 4)
    got = np.asarray(res.filled_data)
    exp = g["adv_expected"]
    assert np.isfinite(got).any()
    assert_close(got, exp.astype(np.float32), atol=1e-5 * np.nanmax(np.abs(exp)), what="reference reproject case")
    # the result carries the target's WCS (what the reference's test compares with wcs.compare)
    w, t = res.wcs, SimpleWCS(hdr_out)
    assert [c[:8] for c in w.ctype[:2]] == ["GLON-SIN", "GLAT-SIN"] and w.frame == ("galactic",)
    np.testing.assert_allclose(w.crval[:2], [134.37608, -31.939241], rtol=0, atol=1e-12)
    np.testing.assert_allclose(w.crpix[:2], [2.0, 2.0])
    np.testing.assert_allclose(w.pixel_scale_matrix, t.pixel_scale_matrix, rtol=1e-12)
    # the target header is wcslib's to_header(): SI units (m/s) for the same channels the source holds in km/s
    assert res.wcs.spectral_unit == "m/s" and cube.wcs.spectral_unit == "km/s"
    np.testing.assert_allclose(res.spectral_axis, np.asarray(cube.spectral_axis) * 1e3, rtol=1e-10)


def test_reproject_raises_for_frames_it_cannot_relate(gpu):
    g = golden("wcs_frames.npz")
    cube = SpectralCube.read(g["adv_data"].astype(np.float32), str(g["adv_header"]))
    tgt = dict(SimpleWCS(str(g["adv_header_out"])).header, CTYPE1="ELON-SIN", CTYPE2="ELAT-SIN")
    with pytest.raises(NotImplementedError, match="frame"):
        cube.reproject(tgt)


# ---- out-of-core streaming (missing 4) ---------------------------------------------------
def _write_cube(tmp_path, d, hdr, name="big.fits", **kw):
    from spectral_cube_amd import io_fits
    p = tmp_path / name
    io_fits.write_fits(str(p), d, hdr, **kw)
    return str(p)


def _c1_header():
    return str(golden("c1_moments.npz")["header"])


@pytest.mark.parametrize("source", ["fits", "ndarray", "memmap_f64"])
def test_out_of_core_moments_and_argmax_equal_the_resident_result(gpu, tmp_path, monkeypatch, source):
    """A cube 4x the HBM budget (SPC_HBM_BUDGET) - a FITS file, a host array, a float64 memory map - stays where it
    is and streams through the device in (y, x) row strips, strip k + 1 staged while strip k computes: moment 0 / 1 / 2,
    argmax / argmin (axis 0 and whole cube), max / min, statistics() are bit-identical to the resident result (the
    z split of the moment kernel pinned: it otherwise follows the strip's size) - the role of _moments.py:89-125 /
    cube_utils.py:277-301 in the reference."""
    from spectral_cube_amd import streaming, synth
    monkeypatch.setenv("SPC_MOMENTS_NSPLIT", "1")
    nz, ny, nx = 96, 200, 64
    d = synth.gaussian_line_cube((nz, ny, nx), 31)
    d[:, 5:9, 3:7] = np.nan
    d[40:, 77, 10] = np.nan
    hdr = _c1_header()
    res = SpectralCube.read(d, hdr)
    budget = d.nbytes // 4
    monkeypatch.setenv("SPC_HBM_BUDGET", str(budget))
    if source == "fits":
        big = SpectralCube.read(_write_cube(tmp_path, d, hdr), device=0)
        assert isinstance(big._source, streaming.FitsSource)
    elif source == "ndarray":
        big = SpectralCube.read(d.copy(), hdr)
    else:
        mm = np.lib.format.open_memmap(str(tmp_path / "c.npy"), mode="w+", dtype=np.float64, shape=d.shape)
        mm[:] = d
        mm.flush()
        big = SpectralCube.read(np.load(str(tmp_path / "c.npy"), mmap_mode="r"), hdr)
    assert big._stream_source() is not None and big._dev is None
    thr = 1.0
    for cube_s, cube_r in ((big, res), (big.with_mask(big > thr), res.with_mask(res > thr))):
        monkeypatch.setenv("SPC_HBM_BUDGET", str(budget))
        got = cube_s.moments012()
        assert cube_s._dev is None, "the cube was never made resident"
        am, an = cube_s.argmax(axis=0), cube_s.argmin(axis=0)
        flat, st = cube_s.argmax(), cube_s.statistics()
        mx = cube_s.max()
        monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
        exp = cube_r.moments012()
        for g_, e_ in zip(got, exp):
            assert np.array_equal(np.asarray(g_), np.asarray(e_), equal_nan=True)
        assert np.array_equal(am, cube_r.argmax(axis=0)) and np.array_equal(an, cube_r.argmin(axis=0))
        assert flat == cube_r.argmax() and mx == cube_r.max()
        est = cube_r.statistics()
        assert st["npts"] == est["npts"] and st["min"] == est["min"] and st["max"] == est["max"]
        assert st["sum"] == pytest.approx(est["sum"], rel=1e-12) and st["sigma"] == pytest.approx(est["sigma"], rel=1e-10)
    # against the oracle too (not only against ourselves)
    inc = np.isfinite(d)
    cen = res.spectral_axis - res.spectral_axis[0]
    e0, e1, e2 = O.moments012(d, inc, cen, res._pix_size_slice(0), res.spectral_axis[0])
    monkeypatch.setenv("SPC_HBM_BUDGET", str(budget))
    m0 = big.moment0()
    assert_close(np.asarray(m0), e0, atol=1e-5 * np.nanmax(np.abs(e0)), what="streamed m0 vs oracle")
    # operators that need the whole cube say so, with the budget in the message
    # order statistics along the spectral axis are per spaxel: they stream too (bit-exact selections)
    masked_s, masked_r = big.with_mask(big > thr), res.with_mask(res > thr)
    got_os = [np.asarray(masked_s.median(axis=0)), np.asarray(masked_s.percentile(30.0, axis=0)), np.asarray(masked_s.mad_std(axis=0)),
              np.asarray(big.median(axis=0))]
    assert big._dev is None
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    exp_os = [np.asarray(masked_r.median(axis=0)), np.asarray(masked_r.percentile(30.0, axis=0)), np.asarray(masked_r.mad_std(axis=0)),
              np.asarray(res.median(axis=0))]
    for g_, e_ in zip(got_os, exp_os):
        assert g_.dtype == e_.dtype and np.array_equal(g_, e_, equal_nan=True)
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.array_equal(got_os[3], np.nanmedian(d, axis=0), equal_nan=True)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(budget))
    # operators that still need the whole cube resident say so, with the budget in the message
    from spectral_cube_amd import Gaussian2DKernel
    with pytest.raises(streaming.HugeCubeError, match="SPC_HBM_BUDGET"):
        big.spatial_smooth(Gaussian2DKernel(1.0)).median(axis=0)


def test_out_of_core_boolean_mask_and_fused_smooth(gpu, tmp_path, monkeypatch):
    """streamed cube + a BooleanArrayMask (strips of the host array travel with the data) and
    spectral_smooth(...).moment1 through the fused kernels, strip by strip."""
    from spectral_cube_amd import Gaussian1DKernel, synth
    monkeypatch.setenv("SPC_MOMENTS_NSPLIT", "1")
    nz, ny, nx = 80, 120, 48
    d = synth.gaussian_line_cube((nz, ny, nx), 32)
    inc = synth.boolean_mask(d, 32).astype(bool)
    hdr = _c1_header()
    res = SpectralCube.read(d, hdr).with_mask(inc)
    exp = res.moments012()
    k = Gaussian1DKernel(2.0)
    exp_s = res.spectral_smooth(k).moment1()
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 5))
    big = SpectralCube.read(d.copy(), hdr).with_mask(inc)
    got = big.moments012()
    for g_, e_ in zip(got, exp):
        assert np.array_equal(np.asarray(g_), np.asarray(e_), equal_nan=True)
    got_s = big.spectral_smooth(k).moment1()
    assert big._dev is None
    assert_close(np.asarray(got_s), np.asarray(exp_s), atol=1e-9 * nz * 500.0, what="streamed fused smooth -> moment1")
    # spatial_smooth -> moment0 of an all-valid streamed cube: the algebraic path (convolution commutes with the sums along
    # z) only ever needs the streamed moment maps
    from spectral_cube_amd import Gaussian2DKernel
    k2 = Gaussian2DKernel(1.5)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    exp_sp = SpectralCube.read(d, hdr).spatial_smooth(k2).moment0()
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 5))
    plain = SpectralCube.read(d.copy(), hdr)
    got_sp = plain.spatial_smooth(k2).moment0()
    assert plain._dev is None
    assert_close(np.asarray(got_sp), np.asarray(exp_sp), atol=1e-12 * np.nanmax(np.abs(np.asarray(exp_sp))), what="streamed spatial_smooth -> moment0")
    sm = O.spectral_smooth(d, inc, k.array)
    e1 = O.moment(sm, inc, 1, res.spectral_axis - res.spectral_axis[0], res._pix_size_slice(0), world0=res.spectral_axis[0])
    s0 = O.moment(sm, inc, 0, res.spectral_axis - res.spectral_axis[0], 1.0)
    with np.errstate(invalid="ignore"):
        wc = np.abs(s0) > 5.0
    assert np.abs(np.asarray(got_s)[wc] - e1[wc]).max() <= 1e-5 * nz * abs(res._pix_size_slice(0))


def test_device_pixel_map_for_the_new_projections(gpu):
    """spc_wcs_pixel_map_f64 with SFL / CEA / MER / AIT / CAR on either side: against astropy (the Galactic AIT all-sky image
    onto an equatorial CAR grid, across frames) and, for every header of the fixture, against the numpy WCS (itself pinned
    against wcslib in test_host_logic) on a pixel map to and from a TAN grid; pixels beyond the edge of the sky -> -1e30."""
    from spectral_cube_amd.wcs import reproject_pixel_map
    g = golden("wcs_projections.npz")
    a, b = SimpleWCS(str(g["map_in"]), naxis=2), SimpleWCS(str(g["map_out"]), naxis=2)
    xs, ys = ops.wcs_pixel_map(a, b, g["map_xs"].shape)
    xs, ys = xs.get(), ys.get()
    ok = np.isfinite(g["map_xs"])
    assert np.abs(xs[ok] - g["map_xs"][ok]).max() <= 1e-9 and np.abs(ys[ok] - g["map_ys"][ok]).max() <= 1e-9
    assert np.all(xs[~ok] == -1e30)
    tan = SimpleWCS({"CTYPE1": "GLON-TAN", "CTYPE2": "GLAT-TAN", "CRVAL1": 118.0, "CRVAL2": 3.0, "CRPIX1": 30.0, "CRPIX2": 25.0,
                     "CDELT1": -0.5, "CDELT2": 0.5}, naxis=2)
    for i in range(int(g["n"])):
        w = SimpleWCS(str(g["hdr%d" % i]), naxis=2)
        for src, dst, shape in ((w, tan, (50, 60)), (tan, w, (48, 64))):
            exs, eys = reproject_pixel_map(src, dst, shape)
            xs, ys = ops.wcs_pixel_map(src, dst, shape)
            xs, ys = xs.get(), ys.get()
            fin = np.isfinite(exs) & np.isfinite(eys)
            assert fin.any()
            # (source coordinates reach hundreds of pixels off the image near the edge of an all-sky projection: relative)
            assert (np.abs(xs[fin] - exs[fin]) <= 1e-9 * (1.0 + np.abs(exs[fin]))).all(), (i, w.proj)
            assert (np.abs(ys[fin] - eys[fin]) <= 1e-9 * (1.0 + np.abs(eys[fin]))).all(), (i, w.proj)
            assert np.all(xs[~fin] == -1e30), (i, w.proj)


def test_out_of_core_cube_to_cube_operators(gpu, tmp_path, monkeypatch):
    """read -> per-spaxel operator -> write without the cube (or its result) ever fitting the HBM budget: the filled copy,
    spectral_smooth, spectral_interpolate and sigma_clip_spectrally of a streamed cube, written strip by strip to a FITS file
    (device byte swap, one pwrite per plane segment) or into a host array, equal the resident results bit for bit."""
    from spectral_cube_amd import Gaussian1DKernel, io_fits, streaming, synth
    nz, ny, nx = 64, 136, 40
    d = synth.gaussian_line_cube((nz, ny, nx), 33)
    d[:, 3:6, 2:5] = np.nan
    inc = synth.boolean_mask(d, 33).astype(bool)
    hdr = _c1_header()
    path = _write_cube(tmp_path, d, hdr)
    k = Gaussian1DKernel(2.0)
    grid_axis = None
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    res = SpectralCube.read(path)
    grid = np.linspace(res.spectral_axis[1], res.spectral_axis[-2], 2 * nz + 3)
    exp_fill = np.asarray(res.with_mask(inc).filled_data)
    exp_sm = np.asarray(res.with_mask(inc).spectral_smooth(k).filled_data)      # (the parent's mask on the parent's voxels)
    assert np.isnan(exp_sm[:, 3:6, 2:5]).all()
    exp_it = np.asarray(res.spectral_interpolate(grid, suppress_smooth_warning=True).filled_data)
    exp_cl = res.sigma_clip_spectrally(2.5)
    exp_cl = ops.fill_masked(exp_cl._device_data(), res._mask_spec(), np.nan).get()
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 6))
    big = SpectralCube.read(path)
    assert big._stream_source() is not None
    # (a) the filled copy of a masked streamed cube into a host array
    out = np.empty((nz, ny, nx), np.float32)
    assert big.with_mask(inc).stream_into(out) is out
    assert np.array_equal(out, exp_fill, equal_nan=True)
    # (b) spectral_smooth -> FITS
    p_sm = str(tmp_path / "sm.fits")
    big.with_mask(inc).spectral_smooth(k).write(p_sm)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    back = SpectralCube.read(p_sm)
    assert back.shape == (nz, ny, nx) and np.array_equal(np.asarray(back.unmasked_data), exp_sm, equal_nan=True)
    np.testing.assert_allclose(back.spectral_axis, res.spectral_axis)
    with pytest.raises(OSError):
        big.write(p_sm)                                   # exists, no overwrite
    # (c) spectral_interpolate (more channels out than in) -> FITS, (d) sigma clip -> host array
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 6))
    p_it = str(tmp_path / "it.fits")
    up = big.spectral_interpolate(grid, suppress_smooth_warning=True)
    up.write(p_it)
    out_cl = big.sigma_clip_spectrally(2.5).stream_into(np.empty((nz, ny, nx), np.float32))
    monkeypatch.setenv("SPC_MOMENTS_NSPLIT", "1")
    got_clm = np.asarray(big.sigma_clip_spectrally(2.5).moment1())       # pending operator -> reduction, strip by strip
    assert big._dev is None and up._dev is None
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    back = SpectralCube.read(p_it)
    assert back.shape == (len(grid), ny, nx) and np.array_equal(np.asarray(back.unmasked_data), exp_it, equal_nan=True)
    np.testing.assert_allclose(back.spectral_axis, grid, rtol=1e-12, atol=1e-9)
    assert np.array_equal(out_cl, exp_cl, equal_nan=True)
    assert np.array_equal(got_clm, np.asarray(res.sigma_clip_spectrally(2.5).moment1()), equal_nan=True)
    # and the astropy-independent reader agrees that the file is a valid FITS image
    img = io_fits.find_image(p_it)
    assert io_fits.cube_shape(img) == (len(grid), ny, nx) and os.path.getsize(p_it) % 2880 == 0


def test_preserve_unit_with_spectral_unit(gpu):
    """tests/test_moments.py:145-170 (test_preserve_unit / test_with_flux_unit): the moment cube read in m/s, switched to
    km/s with with_spectral_unit: moment 0 in K km/s, moment 1 in km/s, moment 2 in km2/s2 - the reference's golden table
    rescaled - and the unit strings follow; a change of KIND of axis is refused."""
    from test_oracle_golden import MOMENTS
    g = golden("moment_cube.npz")
    sc = SpectralCube.read(g["data"], str(g["header"]))
    assert sc.spectral_unit == "m/s"
    kms = sc.with_spectral_unit("km/s")
    np.testing.assert_allclose(kms.spectral_axis, np.asarray(sc.spectral_axis) / 1e3, rtol=1e-14)
    m0, m1 = kms.moment0(axis=0), kms.moment1(axis=0)
    np.testing.assert_allclose(m0, np.asarray(MOMENTS[0][0]) / 1e3, rtol=2e-7)
    np.testing.assert_allclose(m1, np.asarray(MOMENTS[1][0]) / 1e3, rtol=2e-7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        np.testing.assert_allclose(kms.moment2(axis=0), np.asarray(MOMENTS[2][0]) / 1e6, rtol=2e-7)
    assert "km/s" in m0.unit and m1.unit == "km/s"
    assert kms._dev is sc._dev or kms._data is sc._data            # the voxels are shared
    with pytest.raises(NotImplementedError):
        sc.with_spectral_unit("GHz")
    with pytest.raises(NotImplementedError):
        sc.with_spectral_unit("km/s", velocity_convention="radio")


def test_projection_reproject_2d_across_frames(gpu):
    """tests/test_regrid.py:409-434 of the reference (test_reproject_2D): a 5 x 5 Projection on the RA/DEC-SIN header
    reprojected onto GLON-SIN / GLAT-SIN, 5 x 4 pixels: shape, beam kept, the target's WCS; values = the oracle's bilinear
    resampler on astropy's cross-frame pixel map (tests/golden/wcs_frames.npz)."""
    from spectral_cube_amd.cube import Projection
    from spectral_cube_amd.beam import Beam
    g = golden("wcs_frames.npz")
    rng = np.random.default_rng(55)
    img = rng.random((5, 5)).astype(np.float32)
    w_in, w_out = SimpleWCS(str(g["in0"]), naxis=2), SimpleWCS(dict(SimpleWCS(str(g["out0"]), naxis=2).header, NAXIS1=4, NAXIS2=5), naxis=2)
    bm = Beam(1.0 / 3600)
    proj = Projection(img, unit="K", wcs=w_in, beam=bm)
    res = proj.reproject(w_out.header)
    assert res.shape == (5, 4) and res.beam == bm and res.wcs.frame == ("galactic",)
    np.testing.assert_allclose(res.wcs.crval, [134.37608, -31.939241], atol=1e-12)
    exp, foot = O.resample_bilinear(img[None], g["xs0"], g["ys0"])
    assert np.isfinite(exp).any()
    assert_close(np.asarray(res), exp[0].astype(np.float32), atol=1e-5 * np.nanmax(np.abs(exp)), what="Projection.reproject across frames")


def test_out_of_core_spatial_smooth_with_halo_rows(gpu, tmp_path, monkeypatch):
    """spatial_smooth of a streamed cube: strips carry the kernel's half width in extra rows on each side (clipped at the
    cube's edges), the smoothed halo rows - which saw an artificial edge - are dropped.  The written cube and the moments of
    the smoothed cube (masked data: NOT the algebraic shortcut) equal the resident results bit for bit; strips of 8 rows
    against a halo of 6 make every strip depend on two neighbours on each side."""
    from spectral_cube_amd import Gaussian2DKernel, synth
    monkeypatch.setenv("SPC_MOMENTS_NSPLIT", "1")
    nz, ny, nx = 24, 90, 70
    d = synth.gaussian_line_cube((nz, ny, nx), 34)
    d[:, 40:43, 10:13] = np.nan
    inc = synth.boolean_mask(d, 34).astype(bool)
    hdr = _c1_header()
    k2 = Gaussian2DKernel(1.5)                     # 13 x 13: halo 6
    assert k2.array.shape == (13, 13)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    res = SpectralCube.read(d, hdr).with_mask(inc)
    sm = res.spatial_smooth(k2)
    exp_cube = np.asarray(sm.filled_data)           # the parent's mask, on the parent's voxels: the NaN block stays excluded
    assert np.isnan(exp_cube[:, 40:43, 10:13]).all()
    exp_m0, exp_m1 = np.asarray(sm.moment0()), np.asarray(sm.moment1())
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 12))
    big = SpectralCube.read(d.copy(), hdr).with_mask(inc)
    assert big._stream_source() is not None
    out = big.spatial_smooth(k2).stream_into(np.empty((nz, ny, nx), np.float32))
    assert np.array_equal(out, exp_cube, equal_nan=True)
    p = str(tmp_path / "sp.fits")
    big.spatial_smooth(k2).write(p)
    got_m0, got_m1 = np.asarray(big.spatial_smooth(k2).moment0()), np.asarray(big.spatial_smooth(k2).moment1())
    assert big._dev is None
    for g_, e_ in ((got_m0, exp_m0), (got_m1, exp_m1)):
        assert np.array_equal(np.isnan(g_), np.isnan(e_))
        bad = ~np.isnan(e_) & (g_ != e_)
        assert not bad.any(), (int(bad.sum()), np.argwhere(bad)[:5].tolist(), float(np.nanmax(np.abs(g_ - e_) / np.abs(e_))))
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    assert np.array_equal(np.asarray(SpectralCube.read(p).unmasked_data), exp_cube, equal_nan=True)
    # against the oracle as well
    e = O.spatial_smooth(d, inc & np.isfinite(d), k2.array)
    ok = inc & np.isfinite(d) & np.isfinite(e)
    assert np.array_equal(np.isfinite(out), ok)
    assert np.abs(out[ok] - e[ok]).max() <= 1e-5 * np.abs(e[ok]).max()


def test_reference_reproject_3d_case_car_to_sin(gpu):
    """tests/test_regrid.py:511-587 (test_reproject_3D_memory) without its memory accounting: the cube of
    tests/utilities.py:14-37 (GLON-CAR / GLAT-CAR at 0, 0, 1 arcsec pixels, VRAD in km/s) reprojected onto GLON-SIN /
    GLAT-SIN at 0.001, 0.001 deg with CRPIX 2, half the image size; ``filled=False``, ``filled=True``, and a masked cube
    (``cube > 0.1``: masked voxels enter as NaN).  The reference asserts the result's CRVAL / CRPIX; the values here are
    held to the oracle resampler at the host pixel map (both projections pinned against wcslib,
    tests/golden/wcs_projections.npz)."""
    from spectral_cube_amd import synth
    from spectral_cube_amd.wcs import reproject_pixel_map
    nz, ny, nx = 40, 200, 200
    d = synth.gaussian_line_cube((nz, ny, nx), 77)
    px = 1.0 / 3600.0
    hdr = dict(NAXIS=3, NAXIS1=nx, NAXIS2=ny, NAXIS3=nz, CDELT1=-px, CDELT2=px, CRPIX1=nx / 2.0, CRPIX2=ny / 2.0, CRVAL1=0.0, CRVAL2=0.0,
               CTYPE1="GLON-CAR", CTYPE2="GLAT-CAR", CUNIT1="deg", CUNIT2="deg", CRVAL3=-20.0, CUNIT3="km/s", CDELT3=1.0, CRPIX3=1,
               CTYPE3="VRAD", BUNIT="K", BMAJ=3 * px, BMIN=3 * px, BPA=0.0)
    cube = SpectralCube.read(d, hdr)
    hout = dict(hdr, CTYPE1="GLON-SIN", CTYPE2="GLAT-SIN", CRVAL1=0.001, CRVAL2=0.001, CRPIX1=2.0, CRPIX2=2.0, NAXIS1=nx // 2, NAXIS2=ny // 2)
    xs, ys = reproject_pixel_map(SimpleWCS(hdr, naxis=2), SimpleWCS(hout, naxis=2), (ny // 2, nx // 2))
    exp, foot = O.resample_bilinear(d, xs, ys)
    assert foot.any() and np.isfinite(exp).any()
    for filled in (False, True):
        res = cube.reproject(hout, filled=filled)
        assert res.shape == (nz, ny // 2, nx // 2)
        assert res.wcs.crval[0] == 0.001 and res.wcs.crpix[0] == 2.0 and res.wcs.ctype[0].startswith("GLON-SIN")
        assert_close(res._device_data().get(), exp, atol=1e-5 * np.nanmax(np.abs(exp)), what="CAR -> SIN, filled=%s" % filled)
        np.testing.assert_allclose(res.spectral_axis, cube.spectral_axis, rtol=1e-12)
    mcube = cube.with_mask(cube > 0.1)
    assert mcube.mask.include().any() and not mcube.mask.include().all()
    res = mcube.reproject(hout, filled=True)
    expm, _ = O.resample_bilinear(np.where(d > 0.1, d, np.nan).astype(np.float32), xs, ys)
    got = res._device_data().get()
    assert np.array_equal(np.isnan(got), np.isnan(expm))
    assert_close(got, expm, atol=1e-5 * np.nanmax(np.abs(exp)), what="CAR -> SIN, masked")
    assert res.wcs.crval[0] == 0.001 and res.wcs.crpix[0] == 2.0


def test_out_of_core_slabs_of_planes(gpu, tmp_path, monkeypatch):
    """What needs whole image planes streams in slabs of channels (StripPipeline(axis=0)): moments / argmax / argmin /
    the nan-reductions / median / percentile / mad_std along y and x (the (nz, nx) / (nz, ny) maps grow slab by slab on the
    device) and reproject onto a celestial header (pending; write() / stream_into() run it slab by slab) - bit-identical to
    the resident results, the cube never resident.  Reductions along the spectral axis of a streamed cube go strip by strip."""
    from spectral_cube_amd import streaming, synth
    nz, ny, nx = 44, 72, 80
    d = synth.gaussian_line_cube((nz, ny, nx), 41)
    d[3:9, 10:14, 20:26] = np.nan
    inc = synth.boolean_mask(d, 41).astype(bool) | (np.random.default_rng(3).random(d.shape) < 0.3)
    hdr = dict(SimpleWCS(_c1_header()).header, NAXIS1=nx, NAXIS2=ny, NAXIS3=nz, CRPIX1=nx / 2.0, CRPIX2=ny / 2.0)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    res = SpectralCube.read(d, hdr).with_mask(inc)

    def everything(c):
        out = {}
        for ax in (1, 2):
            for order in (0, 1, 2, 3):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    out["moment%d_ax%d" % (order, ax)] = np.asarray(c.moment(order=order, axis=ax))
            out["argmax_ax%d" % ax] = np.asarray(c.argmax(axis=ax))
            out["argmin_ax%d" % ax] = np.asarray(c.argmin(axis=ax))
            out["median_ax%d" % ax] = np.asarray(c.median(axis=ax))
            out["p20_ax%d" % ax] = np.asarray(c.percentile(20.0, axis=ax))
            out["mad_ax%d" % ax] = np.asarray(c.mad_std(axis=ax))
        for ax in (0, 1, 2, (0, 1), (0, 2), (1, 2)):
            for op in ("sum", "mean", "std", "max", "min"):
                out["%s_%s" % (op, ax)] = np.asarray(getattr(c, op)(axis=ax))
        return out

    exp = everything(res)
    c, s_ = np.cos(np.radians(25)), np.sin(np.radians(25))
    target = {k: v for k, v in hdr.items() if not k.endswith("3")}
    target.update(NAXIS=2, NAXIS1=64, NAXIS2=60, CRPIX1=30.0, CRPIX2=28.0, PC1_1=c, PC1_2=-s_, PC2_1=s_, PC2_2=c)
    rp = res.reproject(target)
    exp_rp = np.asarray(rp.filled_data)
    assert np.isnan(exp_rp).any() and np.isfinite(exp_rp).any()
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 5))
    big = SpectralCube.read(d.copy(), hdr).with_mask(inc)
    assert big._stream_source() is not None
    got = everything(big)
    assert big._dev is None
    for k in exp:
        assert got[k].dtype == exp[k].dtype and np.array_equal(got[k], exp[k], equal_nan=True), k
    brp = big.reproject(target)
    assert brp.shape == rp.shape and brp._dev is None and np.array_equal(brp.mask.include(), rp.mask.include())
    out = brp.stream_into(np.empty(rp.shape, np.float32))
    assert np.array_equal(out, exp_rp, equal_nan=True)
    p = str(tmp_path / "rp.fits")
    brp.write(p)
    with pytest.raises(streaming.HugeCubeError):
        brp.filled_data
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    back = SpectralCube.read(p)
    assert np.array_equal(np.asarray(back.unmasked_data), exp_rp, equal_nan=True)
    np.testing.assert_allclose(back.wcs.pixel_scale_matrix, rp.wcs.pixel_scale_matrix, rtol=1e-12)
    np.testing.assert_allclose(back.spectral_axis, res.spectral_axis, rtol=1e-12)
    # convolve_to: one kernel for every channel (an elliptical, rotated one: the non-separable stencil), Jy/beam scaling
    from spectral_cube_amd.beam import Beam
    pix = abs(float(hdr["CDELT1"]))
    hb = dict(hdr, BMAJ=3.0 * pix, BMIN=2.0 * pix, BPA=20.0, BUNIT="Jy/beam")
    tgt = Beam(5.0 * pix, 3.5 * pix, 60.0)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    exp_cv = np.asarray(SpectralCube.read(d, hb).with_mask(inc).convolve_to(tgt).filled_data)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 5))
    bcv = SpectralCube.read(d.copy(), hb).with_mask(inc).convolve_to(tgt)
    assert bcv._dev is None and bcv.beam == tgt
    assert np.array_equal(bcv.stream_into(np.empty(d.shape, np.float32)), exp_cv, equal_nan=True)
    # ... and of a file with a BEAMS table (one kernel per run of channels; a streamed slab knows its first channel)
    from spectral_cube_amd import io_fits, VaryingResolutionSpectralCube
    pb = _write_cube(tmp_path, np.where(np.isnan(d), 0.0, d).astype(np.float32), hb, name="beams.fits")
    majs = (3.0 + 0.5 * (np.arange(nz) // 7 % 3)) * pix           # runs of 7 channels share a beam
    io_fits.append_beams_table(pb, majs, 0.7 * majs, 10.0 * (np.arange(nz) // 7 % 3))
    tgt2 = Beam(6.0 * pix, 5.0 * pix, 30.0)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(1 << 40))
    vres = SpectralCube.read(pb)
    assert isinstance(vres, VaryingResolutionSpectralCube)
    exp_v = np.asarray(vres.convolve_to(tgt2).filled_data)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 5))
    vbig = SpectralCube.read(pb)
    assert isinstance(vbig, VaryingResolutionSpectralCube) and vbig._stream_source() is not None
    vcv = vbig.convolve_to(tgt2)
    assert vcv._dev is None and vcv.beam == tgt2 and not isinstance(vcv, VaryingResolutionSpectralCube)
    assert np.array_equal(vcv.stream_into(np.empty(d.shape, np.float32)), exp_v, equal_nan=True)
    far = dict(target, CRVAL1=float(SimpleWCS(hdr).crval[0]) + 40.0)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 5))
    with pytest.raises(ValueError, match="All values in reprojected cube are nan"):
        big.reproject(far)
