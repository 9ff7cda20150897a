"""GPU: what round 4 added - the device pixel map with SIP / PV1 / polar-safe rotation (astropy fixtures), the ADVICE
round-3 fixes (chunked read-back ordered after the null stream, streamed fused smooth -> moment falling back when the
kernel is too wide to fuse, FITS -> the same FITS)."""
import os

import numpy as np
import pytest

import oracle_np as O
from conftest import assert_close, golden
from spectral_cube_amd import SpectralCube, SimpleWCS, _lib, ops
from spectral_cube_amd.device import DeviceArray

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["verdict_sip", "sip_source", "sip_both", "verdict_pv1", "polar", "polar_arc_zea"])
def test_device_pixel_map_sip_pv1_polar(gpu, name):
    """spc_wcs_pixel_map_f64 against astropy's all_pix2world -> all_world2pix (tests/golden/wcs_strict.npz): the round-3
    verdict's cases - SIP target (was 4 px off), PV1_1 / PV1_2 target (was 1014 px off), the polar pair (was 1e-5 px off) -
    plus SIP on the source side (the inverse) and on both sides; 1e-9 pixel"""
    g = golden("wcs_strict.npz")
    w_in, w_out = SimpleWCS(str(g["map_in_" + name]), naxis=2), SimpleWCS(str(g["map_out_" + name]), naxis=2)
    shape = tuple(int(v) for v in g["map_shape_" + name])
    xs, ys = (a.get() for a in ops.wcs_pixel_map(w_in, w_out, shape))
    yy, xx = g["map_yy_" + name], g["map_xx_" + name]
    assert np.abs(xs[yy, xx] - g["map_xs_" + name]).max() < 1e-9 and np.abs(ys[yy, xx] - g["map_ys_" + name]).max() < 1e-9
    # and the whole grid against the host map (numpy, the same arithmetic)
    from spectral_cube_amd.wcs import reproject_pixel_map
    hx, hy = reproject_pixel_map(w_in, w_out, shape)
    ok = np.isfinite(hx) & np.isfinite(hy)
    assert ok.all() and np.abs(xs - hx).max() < 1e-9 and np.abs(ys - hy).max() < 1e-9


def test_device_sip_inverse_without_solution_is_outside(gpu):
    """a source whose SIP polynomial folds over beyond the image: no pixel, not garbage"""
    base = {"CTYPE1": "RA---TAN-SIP", "CTYPE2": "DEC--TAN-SIP", "CRVAL1": 83.6, "CRVAL2": -5.4, "CRPIX1": 20.5, "CRPIX2": 15.5,
            "CDELT1": -2.0 / 3600, "CDELT2": 2.0 / 3600}
    src = SimpleWCS(dict(base, A_ORDER=2, B_ORDER=2, A_2_0=-2e-3, B_0_2=-2e-3), naxis=2)        # u + A(u) <= 125: nothing maps beyond
    dst = SimpleWCS({k: v for k, v in dict(base, CTYPE1="RA---TAN", CTYPE2="DEC--TAN", CRPIX1=-800.0).items()}, naxis=2)
    xs, ys = (a.get() for a in ops.wcs_pixel_map(src, dst, (30, 40)))
    assert np.all(xs == -1e30) and np.all(ys == -1e30)
    cube = SpectralCube.read(np.ones((3, 30, 40), dtype=np.float32), dict(src.header, CTYPE3="VRAD", CDELT3=1.0, CRVAL3=0.0, CRPIX3=1.0))
    with pytest.raises(ValueError, match="All values in reprojected cube are nan"):
        cube.reproject(dict(dst.header, NAXIS1=40, NAXIS2=30))


def test_reproject_onto_a_sip_header_end_to_end(gpu):
    """cube.reproject onto the verdict's RA---TAN-SIP header: the values follow astropy's map (fixture) through the
    oracle's bilinear restatement, 1e-5"""
    g = golden("wcs_strict.npz")
    name = "verdict_sip"
    shape = tuple(int(v) for v in g["map_shape_" + name])
    rng = np.random.default_rng(5)
    yy0, xx0 = np.mgrid[0:shape[0], 0:shape[1]]
    d = np.stack([np.sin(xx0 / 9.0) * np.cos(yy0 / 7.0) + 0.1 * k for k in range(3)]).astype(np.float32)
    hin = dict(SimpleWCS(str(g["map_in_" + name]), naxis=2).header, CTYPE3="VRAD", CDELT3=1.0, CRVAL3=0.0, CRPIX3=1.0, CUNIT3="km/s")
    cube = SpectralCube.read(d, hin)
    res = cube.reproject(str(g["map_out_" + name]))
    got = np.asarray(res.filled_data)
    yy, xx = g["map_yy_" + name], g["map_xx_" + name]
    exs, eys = g["map_xs_" + name], g["map_ys_" + name]
    full_x, full_y = np.full(shape, -1e30), np.full(shape, -1e30)
    full_x[yy, xx], full_y[yy, xx] = exs, eys
    exp, _ = O.reproject_separable(d.astype(np.float64), full_x, full_y, None)
    sel = np.zeros(shape, dtype=bool)
    sel[yy, xx] = True
    assert_close(got[:, sel], exp[:, sel].astype(np.float32), atol=1e-5 * np.abs(d).max(), what="reproject onto SIP")
    assert np.isfinite(got[:, sel]).mean() > 0.8


def test_chunked_readback_waits_for_the_null_stream(gpu):
    """DeviceArray.get() of a result above 256 MiB comes down in chunks on a stream of its own; it
    must see what kernels on the NULL stream (every resident cube path) wrote.  A long null-stream smoothing into a 512 MiB
    output, read back at once, with the pinned chunk buffers already warm."""
    nz, ny, nx = 512, 512, 512
    rng = np.random.default_rng(3)
    d = rng.standard_normal((nz, ny, nx)).astype(np.float32)
    dev = DeviceArray.from_numpy(d)
    warm = dev.get()                                   # pins the chunk buffers (the slow first call hid the race)
    assert np.array_equal(warm, d)
    from spectral_cube_amd import Gaussian1DKernel
    karr = Gaussian1DKernel(6.0).array                 # 49 taps: ~10 ms of kernel at this size
    for _ in range(3):
        out = ops.spectral_conv(dev, karr)             # launched on the null stream, no sync
        got = out.get()
        # spot-check whole planes at the END of the buffer (the last chunks: what a racing copy would fetch half-written)
        exp = O.spectral_smooth(d[:, -8:, :].astype(np.float64), np.ones((nz, 8, nx), bool), karr)
        assert_close(got[:, -8:, :], exp.astype(np.float32), atol=1e-5 * np.abs(exp).max(), what="chunked read-back")
        del out


def test_streamed_fused_smooth_moment_falls_back_for_wide_kernels(gpu, monkeypatch):
    """out-of-core parent, spectral_smooth with more taps than the fused rings take (> 33) on a
    cube with NaNs, then moment: the fused call raises HipUnsupported inside the strip loop; the streamed path now
    materialises the smoothed STRIP and reduces it (what the resident path always did) - same maps as the resident cube."""
    from spectral_cube_amd import Gaussian1DKernel
    nz, ny, nx = 96, 40, 64
    rng = np.random.default_rng(8)
    d = rng.standard_normal((nz, ny, nx)).astype(np.float32) + 2.0
    d[rng.random(d.shape) < 0.05] = np.nan
    hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3, "CDELT3": 0.5, "CUNIT3": "km/s",
           "CRPIX1": 1, "CRPIX2": 1, "CRPIX3": 1, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": -16.0}
    kern = Gaussian1DKernel(5.5)                      # 45 taps
    assert kern.array.size > 33
    res = SpectralCube.read(d, hdr)
    exp = {o: np.asarray(res.spectral_smooth(kern).moment(order=o)) for o in (0, 1)}
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 3))
    big = SpectralCube.read(d.copy(), hdr)
    assert big._stream_source() is not None
    for o in (0, 1):
        got = np.asarray(big.spectral_smooth(kern).moment(order=o))
        assert_close(got, exp[o], atol=1e-6 * np.nanmax(np.abs(exp[o])), what="streamed wide smooth -> moment %d" % o)


def test_streamed_write_onto_its_own_source_file(gpu, tmp_path, monkeypatch):
    """write(path, overwrite=True) of a streamed cube read FROM path used to truncate the input before
    the first strip was read.  The sink now fills a sibling file and renames it at the end."""
    from spectral_cube_amd import io_fits
    nz, ny, nx = 24, 32, 48
    rng = np.random.default_rng(9)
    d = rng.standard_normal((nz, ny, nx)).astype(np.float32)
    hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3, "CDELT3": 0.5, "CUNIT3": "km/s",
           "CRPIX1": 1, "CRPIX2": 1, "CRPIX3": 1, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": -16.0}
    path = os.path.join(tmp_path, "cube.fits")
    SpectralCube.read(d, hdr).write(path)
    monkeypatch.setenv("SPC_HBM_BUDGET", str(nz * 8 * nx * 4 * 3))
    cube = SpectralCube.read(path)
    assert cube._stream_source() is not None
    masked = cube.with_mask(cube > 0.0)
    masked.write(path, overwrite=True)                 # NaN where excluded, onto the file being read
    assert [f for f in os.listdir(tmp_path) if "spc-part" in f] == []
    monkeypatch.delenv("SPC_HBM_BUDGET")
    back = np.asarray(SpectralCube.read(path).unmasked_data)
    assert_close(back, np.where(d > 0.0, d, np.nan).astype(np.float32), what="file written onto itself")


@pytest.mark.parametrize("order", [2, 3])
def test_spline_resample_against_scipy_fixture(gpu, order):
    """spc_resample_spline_f32 (reproject order 'biquadratic' / 'bicubic') against scipy.ndimage.map_coordinates on the
    border-replicated cube (tests/golden/reproject_spline_scipy.npz): float32 results of float32 inputs, 1e-5 of the data
    range; NaN pattern (outside [-0.5, n - 0.5], NaN coordinates) identical; footprint identical"""
    g = golden("reproject_spline_scipy.npz")
    for n in range(int(g["n"])):
        d, xs, ys = g["data%d" % n].astype(np.float32), g["xs%d" % n], g["ys%d" % n]
        exp, efoot = O.resample_spline(d.astype(np.float64), xs, ys, order)
        # (the oracle on the float32-rounded samples; the fixture itself - float64 samples - within the rounding of the inputs)
        out, foot = ops.resample_spline(DeviceArray.from_numpy(d), np.where(np.isnan(xs), -1e30, xs), ys, order)
        got = out.get()
        assert_close(got, exp.astype(np.float32), atol=1e-5 * np.abs(d).max(), what="spline order %d case %d" % (order, n))
        assert_close(got, g["expected%d_%d" % (order, n)].astype(np.float32), atol=2e-5 * np.abs(d).max(), what="vs scipy fixture")
        assert np.array_equal(foot.get().astype(bool), efoot[0])


def test_spline_resample_large_planes_and_slabs(gpu):
    """planes wider than the causal-init horizon and the 64-column tiles, several slabs of coefficients"""
    rng = np.random.default_rng(12)
    nz, ny, nx = 7, 203, 331
    d = rng.standard_normal((nz, ny, nx)).astype(np.float32)
    yy, xx = np.mgrid[0:150, 0:170]
    xs = 0.9 * xx * nx / 170.0 + 0.31 * np.sin(yy / 9.0) - 0.4
    ys = 0.95 * yy * ny / 150.0 + 0.27 * np.cos(xx / 7.0) - 0.45
    for order in (2, 3):
        exp, _ = O.resample_spline(d.astype(np.float64), xs, ys, order)
        out, _ = ops.resample_spline(DeviceArray.from_numpy(d), xs, ys, order, slab_bytes=3 * (ny + 2) * (nx + 2) * 8)
        assert_close(out.get(), exp.astype(np.float32), atol=1e-5 * np.abs(d).max(), what="large spline order %d" % order)


def test_reproject_bicubic_and_biquadratic_end_to_end(gpu):
    """cube.reproject(order='bicubic' | 'biquadratic' | 3 | 2) (spectral_cube.py:2667-2676): values = the oracle's scipy-pinned
    spline on this package's pixel map; a masked / NaN sample anywhere makes the reference's result all NaN -> its ValueError"""
    from spectral_cube_amd.wcs import reproject_pixel_map
    rng = np.random.default_rng(4)
    nz, ny, nx = 3, 48, 56
    yy, xx = np.mgrid[0:ny, 0:nx]
    d = np.stack([np.sin(xx / 5.0) * np.cos(yy / 6.0) + 0.2 * k for k in range(nz)]).astype(np.float32)
    hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3, "CDELT3": 0.5, "CUNIT3": "km/s",
           "CRPIX1": 28.0, "CRPIX2": 24.0, "CRPIX3": 1, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": -16.0}
    c, s_ = np.cos(np.radians(20.0)), np.sin(np.radians(20.0))
    tgt = dict({k: v for k, v in hdr.items() if not k.endswith("3")}, PC1_1=c, PC1_2=-s_, PC2_1=s_, PC2_2=c, NAXIS=2, NAXIS1=50, NAXIS2=44)
    cube = SpectralCube(d, header=hdr)
    xs, ys = reproject_pixel_map(SimpleWCS(hdr).drop_spectral(), SimpleWCS(tgt, naxis=2), (44, 50))
    for order, code in (("bicubic", 3), ("biquadratic", 2), (3, 3)):
        res = cube.reproject(tgt, order=order)
        exp, foot = O.resample_spline(d.astype(np.float64), xs, ys, code)
        assert_close(np.asarray(res.filled_data), exp.astype(np.float32), atol=1e-5 * np.abs(d).max(), what="reproject %r" % (order,))
        assert foot[0].sum() > 800
    bad = d.copy()
    bad[1, 10, 10] = np.nan
    with pytest.raises(ValueError, match="All values in reprojected cube are nan"):
        SpectralCube.read(bad, hdr).reproject(tgt, order="bicubic")
    with pytest.raises(ValueError, match="order"):
        cube.reproject(tgt, order="quintic")


def test_device_pixel_map_fk4(gpu):
    """spc_wcs_pixel_map_f64 with the E-terms steps (ABI 4: 15 doubles of frame data) against astropy (wcs_fk4.npz), 1e-9 px"""
    g = golden("wcs_fk4.npz")
    for i in range(int(g["n"])):
        a, b = SimpleWCS(str(g["in%d" % i]), naxis=2), SimpleWCS(str(g["out%d" % i]), naxis=2)
        xs, ys = (v.get() for v in ops.wcs_pixel_map(a, b, g["xs%d" % i].shape))
        assert np.abs(xs - g["xs%d" % i]).max() <= 1e-9 and np.abs(ys - g["ys%d" % i]).max() <= 1e-9, i


def test_streamed_reproject_onto_a_cube_header_with_another_spectral_axis(gpu, monkeypatch):
    """VERDICT round 3, missing 3: reproject of a cube above the HBM budget onto a 3-axis header whose channels differ
    from the cube's raised HugeCubeError; it now runs in two streamed steps (source slabs -> resampled planes in a host
    array, its row strips -> output channels) and equals the resident result (one resample + one blend) bit for bit"""
    rng = np.random.default_rng(21)
    nz, ny, nx = 40, 64, 72
    d = rng.standard_normal((nz, ny, nx)).astype(np.float32)
    d[5, 10, 10] = np.nan
    hdr = {"NAXIS": 3, "NAXIS1": nx, "NAXIS2": ny, "NAXIS3": nz, "CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD",
           "CDELT1": -1e-3, "CDELT2": 1e-3, "CDELT3": 0.5, "CUNIT3": "km/s", "CRPIX1": 36.0, "CRPIX2": 32.0, "CRPIX3": 1,
           "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": -10.0}
    c, s_ = np.cos(np.radians(15)), np.sin(np.radians(15))
    tgt = dict(hdr, NAXIS1=60, NAXIS2=56, NAXIS3=55, CRPIX1=30.0, CRPIX2=28.0, PC1_1=c, PC1_2=-s_, PC2_1=s_, PC2_2=c,
               CDELT3=0.37, CRVAL3=-10.4)                       # the first channels lie below the cube: NaN planes
    res = SpectralCube.read(d, hdr)
    exp_cube = res.reproject(tgt)
    exp = np.asarray(exp_cube.filled_data)
    assert exp.shape == (55, 56, 60) and np.isnan(exp[0]).all() and np.isfinite(exp[5:50]).any()
    monkeypatch.setenv("SPC_HBM_BUDGET", str(d.nbytes // 4))
    big = SpectralCube.read(d.copy(), hdr)
    assert big._stream_source() is not None
    got_cube = big.reproject(tgt)
    assert got_cube.shape == exp_cube.shape and np.array_equal(got_cube.mask.include(), exp_cube.mask.include())
    monkeypatch.setenv("SPC_HBM_BUDGET", str(exp.nbytes // 3))   # the result does not fit either
    out = got_cube.stream_into(np.empty(exp.shape, np.float32))
    assert np.array_equal(out, exp, equal_nan=True)
    np.testing.assert_allclose(got_cube.spectral_axis, exp_cube.spectral_axis, rtol=1e-12)


DASK_CUBE_WORKER = r'''
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import numpy as np
import dask, dask.array as da
import oracle_np as O
from spectral_cube_amd.dask_adapter import DaskCubeOps
from spectral_cube_amd.kernels import Gaussian1DKernel, Gaussian2DKernel
rng = np.random.default_rng(6)
nz, ny, nx = 48, 72, 88
d = rng.standard_normal((nz, ny, nx)).astype(np.float32) + 1.0
d[5:9, 3, 7] = np.nan                                     # masked voxels arrive NaN-filled (FilledArrayHandler, dask_spectral_cube.py:205-230)
hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3, "CDELT3": 0.5, "CUNIT3": "km/s",
       "CRPIX1": 1, "CRPIX2": 1, "CRPIX3": 1, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": -3.0}
def close(a, b, scale, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)), what
    ok = np.isfinite(b)
    assert np.abs(a[ok] - b[ok]).max() <= 1e-5 * scale, (what, np.abs(a[ok] - b[ok]).max())
for dtype, chunks in ((np.float32, (-1, 24, 32)), (np.float64, (16, 36, -1))):
    arr = da.from_array(d.astype(dtype), chunks=chunks)
    ops = DaskCubeOps(arr, hdr)
    inc = np.isfinite(d)
    cen = np.arange(nz) * 0.5
    e = O.moments012(d, inc, cen, 0.5, -3.0)
    for got, exp, sc in zip(ops.moments012(), e, (np.nanmax(np.abs(e[0])), 24.0, np.nanmax(np.abs(e[2])))):
        close(got, exp, sc, "moments")
    close(ops.moment(order=1), e[1], 24.0, "moment1")
    k1, k2 = Gaussian1DKernel(1.5), Gaussian2DKernel(1.2)
    sm = ops.spectral_smooth(k1)
    assert isinstance(sm, da.Array) and sm.dtype == np.float32 and (dtype != np.float32 or sm.chunks == arr.chunks)
    exp = O.spectral_smooth(d, inc, k1.array)
    close(sm.compute(), np.where(inc, exp, np.nan), np.nanmax(np.abs(exp)), "spectral_smooth")      # (filled: the mask is kept)
    exp = O.spatial_smooth(d, inc, k2.array)
    close(ops.spatial_smooth(k2).compute(), np.where(inc, exp, np.nan), np.nanmax(np.abs(exp)), "spatial_smooth")
    clipped = ops.sigma_clip_spectrally(2.5).compute()
    expc = O.sigma_clip(d, inc, 2.5)
    assert np.mean(np.isnan(clipped) != np.isnan(expc)) < 1e-3
print("DASK_CUBE_OK", dask.__version__)
'''


def test_cube_level_dask_entry(gpu, tmp_path):
    """the operators of a dask-backed cube one level above the per-chunk seam - windows of the dask
    array go straight into the strip pipeline's pinned buffers (streaming.DaskSource), the strip kernels run, cube -> cube
    results come back as a dask array over the host sink.  dask lives in the image's conda interpreter only: subprocess."""
    import subprocess
    from conftest import REPO
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or subprocess.run([py, "-c", "import dask.array"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with dask on this box")
    script = tmp_path / "dask_cube_worker.py"
    script.write_text(DASK_CUBE_WORKER)
    env = dict(os.environ)
    sys_cxx = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(sys_cxx):
        env["LD_PRELOAD"] = sys_cxx
    r = subprocess.run([py, "-B", str(script), REPO], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DASK_CUBE_OK" in r.stdout, r.stdout + r.stderr


# ---- wide sources: the spectral moments in float64 ---------------------------------------------------
@pytest.mark.parametrize("name", ["f64", "i32"])
@pytest.mark.parametrize("source", ["file", "array"])
def test_float64_cube_gives_the_reference_float64_moments(gpu, tmp_path, name, source):
    """A BITPIX = -64 image (a 2 mK .. 1 K line on a 1000 K baseline: one float32 ulp there is 61 uK) and a BITPIX = 32 image
    with BSCALE / BZERO / BLANK, read as the reference reads them (float64, masks.py:225): moment 0 / 1 / 2 / 3, argmax /
    argmin along the spectral axis against the reference's own maps (tests/golden/moments_f64.npz, oracle/gen_golden.py::
    case_moments_f64), without and with a `cube > threshold` mask whose threshold float32 cannot represent: 1e-12, no
    PrecisionWarning on the way.  The float32 staging of the same cube misses these maps by orders of magnitude (below)."""
    import warnings as W
    from spectral_cube_amd import PrecisionWarning, io_fits
    g = golden("moments_f64.npz")
    path = str(tmp_path / (name + ".fits"))
    with open(path, "wb") as f:
        f.write(g[name + "_file"].tobytes())
    with W.catch_warnings():
        W.simplefilter("error", PrecisionWarning)
        if source == "file":
            cube = SpectralCube.read(path)
            assert cube._dev is None and cube._is_wide()          # nothing staged as float32
        else:
            cube = SpectralCube.read(g["data_" + name], io_fits.cube_header(io_fits.find_image(path)))
        for tag, c in (("u", cube), ("m", cube.with_mask(cube > float(g["thr_" + name])))):
            for backend in ("np", "dask"):
                for order in range(4):
                    exp = g["mom%d_%s_%s_%s" % (order, name, tag, backend)]
                    got = np.asarray(c.moment(order=order, axis=0))
                    assert got.dtype == np.float64 and np.array_equal(np.isnan(got), np.isnan(exp)), (tag, order)
                    ok = ~np.isnan(exp)
                    # (the two back-ends of the reference differ from each other by a few 1e-13 in the higher orders)
                    rtol = 1e-12 if order < 2 else 2e-12
                    assert np.abs(got[ok] - exp[ok]).max() <= rtol * np.abs(exp[ok]).max(), (tag, order, backend, np.abs(got[ok] - exp[ok]).max())
                assert np.array_equal(c.argmax(axis=0), g["argmax_%s_%s_%s" % (name, tag, backend)])
                assert np.array_equal(c.argmin(axis=0), g["argmin_%s_%s_%s" % (name, tag, backend)])
            m0, m1, m2 = c.moments012()
            assert np.array_equal(np.asarray(m0), np.asarray(c.moment0()), equal_nan=True)
            assert np.array_equal(np.asarray(m2), np.asarray(c.moment(order=2)), equal_nan=True)
        assert cube._dev is None and cube._data_id.dev64 is not None
    # any other operator stages float32 - and says so, once
    with pytest.warns(PrecisionWarning, match="narrowed to float32"):
        narrow = np.asarray(ops.moments(cube._device_data(), DeviceArray.from_numpy(np.arange(cube.shape[0], dtype=np.float64), 0), want=("m1",))["m1"].get())
    if name == "f64":
        wide = np.asarray(ops.moments_f64(cube._device_data64(), DeviceArray.from_numpy(np.arange(cube.shape[0], dtype=np.float64), 0), want=("m1",))["m1"].get())
        ok = np.isfinite(wide)
        assert np.abs(narrow[ok] - wide[ok]).max() > 1e-8           # what the narrowing costs on this cube (channels): 1e-7


@pytest.mark.parametrize("shape", [(33, 5, 7), (64, 6, 8), (9, 3, 130)])
def test_moments_f64_kernel_against_the_oracle(gpu, shape):
    """spc_moments_f64 / spc_moment_order_f64 through ops: odd and even widths (one / two spaxels per lane), a mask array and
    float64 thresholds, empty rays, ties in the extrema - the oracle in float64, 1e-13"""
    rng = np.random.default_rng(shape[2])
    nz, ny, nx = shape
    d = 5.0 + rng.standard_normal(shape)
    d[rng.random(shape) < 0.05] = np.nan
    d[:, 1, 2] = np.nan
    d[2, 0, 0] = d[7, 0, 0] = 99.0                             # a tie: the first index wins
    marr = rng.random(shape) < 0.8
    cen = np.linspace(-3.0, 4.0, nz)
    dev, d_cen = DeviceArray.from_numpy(d, 0), DeviceArray.from_numpy(cen, 0)
    for use_arr in (False, True):
        thr = 4.2000000001
        spec = ops.MaskSpec(_lib.MASK_FINITE | _lib.MASK_GT | (_lib.MASK_ARRAY if use_arr else 0), thr, 0.0,
                            DeviceArray.from_numpy(marr.view(np.uint8), 0) if use_arr else None)
        inc = np.isfinite(d) & (d > thr) & (marr if use_arr else True)
        r = ops.moments_f64(dev, d_cen, dv=0.5, m1_add=2.0, mask=spec, want=ops._WANT_F64)
        for order in range(3):
            exp = O.moment(d, inc, order, cen, 0.5, axis=0, world0=2.0)
            assert_close(r["m%d" % order].get(), exp, rtol=1e-13, atol=1e-13, what="f64 m%d" % order)
        o3 = ops.moment_order_f64(dev, d_cen, 3, r["mu"], r["s0"], mask=spec).get()
        assert_close(o3, O.moment(d, inc, 3, cen, 0.5, axis=0), rtol=1e-12, atol=1e-12, what="f64 m3")
        assert np.array_equal(r["argmax"].get(), O.argmax(d, inc)) and np.array_equal(r["argmin"].get(), O.argmin(d, inc))
        assert np.array_equal(r["nvalid"].get(), inc.sum(axis=0))
        filled = np.where(inc, d, np.nan)
        with np.errstate(all="ignore"):
            import warnings as W
            with W.catch_warnings():
                W.simplefilter("ignore")
                assert np.array_equal(r["vmax"].get(), np.nanmax(filled, axis=0), equal_nan=True)
                assert np.array_equal(r["vmin"].get(), np.nanmin(filled, axis=0), equal_nan=True)


# ---- the moment kernel's template space (round 4: predicate form, sums only / + count, waves per block, XCD grouping) ----
_MOM_MASKS = {
    "none": lambda d, arr: (0, 0.0, 0.0, None, ~np.isnan(d)),
    "isfinite": lambda d, arr: (_lib.MASK_FINITE, 0.0, 0.0, None, np.isfinite(d)),
    "array": lambda d, arr: (0, 0.0, 0.0, arr, arr & ~np.isnan(d)),
    "array + isfinite + (lo, hi]": lambda d, arr: (_lib.MASK_FINITE | _lib.MASK_GT | _lib.MASK_LE, -0.25, 3.0, arr,
                                                   arr & np.isfinite(d) & (np.nan_to_num(d) > np.float32(-0.25)) & (np.nan_to_num(d) <= np.float32(3.0))),
    "[lo, hi)": lambda d, arr: (_lib.MASK_GE | _lib.MASK_LT, 0.5, 2.5, None,
                                ~np.isnan(d) & (np.nan_to_num(d, nan=-9.0) >= np.float32(0.5)) & (np.nan_to_num(d, nan=9.0) < np.float32(2.5))),
}


@pytest.mark.parametrize("mask_kind", sorted(_MOM_MASKS))
@pytest.mark.parametrize("launch", [{}, {"SPC_MOMENTS_ZW": "8"}, {"SPC_MOMENTS_ZW": "8", "SPC_MOMENTS_U": "2", "SPC_MOMENTS_XCD": "1"},
                                    {"SPC_MOMENTS_ZW": "4", "SPC_MOMENTS_U": "4", "SPC_MOMENTS_XCD": "1", "SPC_MOMENTS_NSPLIT": "3"},
                                    {"SPC_MOMENTS_ZW": "1", "SPC_MOMENTS_NSPLIT": "1"}])
@pytest.mark.parametrize("shape", [(640, 24, 136), (77, 13, 52), (530, 9, 50)])
def test_moment_kernel_template_space(gpu, monkeypatch, shape, launch, mask_kind):
    """spc_moments_f32 with the three sums alone (EXT = 0: an any-valid bit instead of the count), with the count, for every
    predicate form (none / isfinite: one compare; thresholds: the canonical three), with and without a mask array, for 1 / 4 / 8
    waves per block, z splits with the combine kernel, XCD grouping with block counts that are no multiple of 8, 16-byte and
    2-lane rows (nx = 50): against the oracle (float64 sums: 1e-12 of the scale), NaN pattern and counts exact."""
    from spectral_cube_amd import synth
    for k, v in launch.items():
        monkeypatch.setenv(k, v)
    nz, ny, nx = shape
    rng = np.random.default_rng(nz + nx)
    d = synth.gaussian_line_cube(shape, 40 + nz)
    d[rng.random(shape) < 0.01] = np.nan
    if mask_kind in ("isfinite", "array + isfinite + (lo, hi]"):
        d[rng.random(shape) < 0.003] = np.inf
    d[:, 1, 3:9] = np.nan                                    # rays without a valid sample
    arr = rng.random(shape) < 0.7
    arr[:, 2, 10:20] = False
    flags, lo, hi, marr, inc = _MOM_MASKS[mask_kind](d, arr)
    spec = ops.MaskSpec(flags | (_lib.MASK_ARRAY if marr is not None else 0), lo, hi,
                        DeviceArray.from_numpy(marr.astype(np.uint8)) if marr is not None else None)
    v = synth.spectral_axis(nz)
    cen = v - v[0]
    cref = cen[nz // 2]
    dcube, dcen = DeviceArray.from_numpy(d), DeviceArray.from_numpy(cen - cref)
    e0, e1, e2 = O.moments012(d, inc, cen, 500.0, v[0])
    nval = inc.sum(axis=0)
    for want in (("m0", "m1", "m2"), ("m0", "m1", "m2", "nvalid"), ("m0",)):
        r = ops.moments(dcube, dcen, dv=500.0, m1_add=cref + v[0], mask=spec, want=want)
        with np.errstate(all="ignore"):
            m0 = r["m0"].get()
            assert np.array_equal(np.isnan(m0), nval == 0), (want, "all-bad rays are NaN, no others")
            assert_close(m0, e0, atol=1e-12 * np.nanmax(np.abs(e0)), what="m0")
            if "m1" in want:
                assert_close(r["m1"].get(), e1, atol=1e-9 * 500.0 * nz, what="m1")
                m2, okm = r["m2"].get(), np.isfinite(e2) & (np.abs(e0) > 1e-3 * np.nanmax(np.abs(e0)))
                assert np.array_equal(np.isnan(m2), np.isnan(e2))
                assert np.all(np.abs(m2[okm] - e2[okm]) <= 1e-8 * (500.0 * nz) ** 2)
        if "nvalid" in want:
            assert np.array_equal(r["nvalid"].get(), nval)


@pytest.mark.parametrize("shape", [(40, 96, 257), (64, 128, 256), (3, 50, 1366)])
@pytest.mark.parametrize("kind", ["none", "array", "array + thresholds", "isfinite"])
def test_stats_global_linear_groups_and_ragged_end(gpu, shape, kind):
    """statistics() of a contiguous cube through the round-4 kernel: whole 8-chunk groups (wave-wide popcount, four accumulator
    sets, NaN-substituted extrema), the last partial group (whole chunks, then a ragged chunk through the per-lane
    accumulators) and the scalar tail, with one compare (no threshold term) and with the canonical three: count and extrema
    exact, sums to 1e-12 (float64 sums of the same float32 samples in another order)."""
    rng = np.random.default_rng(shape[2])
    d = (rng.standard_normal(shape) * 3).astype(np.float32)
    d[rng.random(shape) < 0.01] = np.nan
    d[rng.random(shape) < 0.002] = -np.inf
    arr = rng.random(shape) < 0.6
    if kind == "none":
        spec, inc = None, ~np.isnan(d)
    elif kind == "isfinite":
        spec, inc = ops.MaskSpec(_lib.MASK_FINITE), np.isfinite(d)
    elif kind == "array":
        spec, inc = ops.MaskSpec(_lib.MASK_ARRAY, array=DeviceArray.from_numpy(arr.astype(np.uint8))), arr & ~np.isnan(d)
    else:
        spec = ops.MaskSpec(_lib.MASK_ARRAY | _lib.MASK_GE | _lib.MASK_LT, -2.0, 4.5, DeviceArray.from_numpy(arr.astype(np.uint8)))
        with np.errstate(invalid="ignore"):
            inc = arr & (d >= np.float32(-2.0)) & (d < np.float32(4.5))
    st = ops.stats_global(DeviceArray.from_numpy(d), mask=spec)
    sel = d[inc].astype(np.float64)
    assert st["npts"] == sel.size and st["min"] == sel.min() and st["max"] == sel.max()
    if np.isfinite(sel.sum()):
        assert st["sum"] == pytest.approx(sel.sum(), rel=1e-12, abs=1e-9) and st["sumsq"] == pytest.approx((sel * sel).sum(), rel=1e-12)
    else:
        assert st["sum"] == sel.sum()                        # -inf samples under a mask that keeps them: the sum is -inf, like numpy's


@pytest.mark.parametrize("shape", [(100, 131), (16, 64), (257, 70)])
@pytest.mark.parametrize("kind", ["gauss29", "gauss33 x gauss9", "row 1 x 9", "column 9 x 1", "rotated (not an outer product)", "gauss35 (too wide)"])
def test_map_conv2d_separable_form(gpu, monkeypatch, shape, kind):
    """spc_map_conv2d_f64 (the algebraic spatial_smooth -> moment path convolves the moment sums, not the planes): an outer-product
    kernel of up to 33 x 33 taps takes the two-pass LDS form; it must agree with the direct form (SPC_MAP_CONV_DIRECT=1) to float64
    rounding and with scipy's zero-boundary convolution of the normalised kernel; other kernels keep the direct form."""
    from scipy.signal import convolve2d
    rng = np.random.default_rng(shape[1])
    g = lambda s, n: np.exp(-0.5 * ((np.arange(n) - n // 2) / s) ** 2)      # noqa: E731
    if kind == "gauss29":
        k = np.outer(g(3.4, 29), g(3.4, 29))
    elif kind == "gauss33 x gauss9":
        k = np.outer(g(4.0, 33), g(1.2, 9) * (1 + 0.1 * np.arange(9)))      # asymmetric along x: the flip matters
    elif kind == "row 1 x 9":
        k = (g(1.5, 9) * (1 + 0.2 * np.arange(9)))[None, :]
    elif kind == "column 9 x 1":
        k = g(1.5, 9)[:, None]
    elif kind == "gauss35 (too wide)":
        k = np.outer(g(4.2, 35), g(4.2, 35))
    else:
        yy, xx = np.mgrid[-7:8, -7:8]
        k = np.exp(-0.5 * (((xx + 0.7 * yy) / 3.0) ** 2 + ((yy - 0.7 * xx) / 1.5) ** 2))
    m = rng.standard_normal(shape) * 100.0
    dm = DeviceArray.from_numpy(m)
    got = ops.map_conv2d(dm, k).get()
    monkeypatch.setenv("SPC_MAP_CONV_DIRECT", "1")
    direct = ops.map_conv2d(dm, k).get()
    exp = convolve2d(m, k / k.sum(), mode="same", boundary="fill", fillvalue=0.0)
    scale = np.abs(exp).max()
    assert np.abs(got - direct).max() <= 1e-13 * scale
    assert np.abs(got - exp).max() <= 1e-12 * scale
