"""GPU: what round 2 added to the contract - reproject with a cube header (spectral axis resampled too),
nearest neighbour, the all-NaN error on VALUES; moments of order > 2 along spatial axes; filled data on
write; masks that belong to another device-resident cube; the convolve= seam; and the boundary rule that
no entry point allocates or drains: two different stencils in flight on two streams."""
import os
import threading
import warnings

import numpy as np
import pytest

import oracle_np as O
from conftest import assert_close, golden
from spectral_cube_amd import (SpectralCube, SimpleWCS, Gaussian1DKernel, Gaussian2DKernel, BooleanArrayMask,
                               _lib, ops, synth)
from spectral_cube_amd import masks as M
from spectral_cube_amd.device import DeviceArray, Stream

pytestmark = pytest.mark.gpu


def _hdr(nz, ny, nx, **kw):
    h = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3,
         "CDELT3": 0.5, "CUNIT3": "km/s", "CRPIX1": nx / 2, "CRPIX2": ny / 2, "CRPIX3": 1, "CRVAL1": 10.0,
         "CRVAL2": 20.0, "CRVAL3": -16.0, "BUNIT": "K", "NAXIS": 3, "NAXIS1": nx, "NAXIS2": ny, "NAXIS3": nz}
    h.update(kw)
    return h


def _rot(h, deg):
    a = np.deg2rad(deg)
    return dict(h, PC1_1=np.cos(a), PC1_2=-np.sin(a), PC2_1=np.sin(a), PC2_2=np.cos(a))


def test_reproject_cube_header_resamples_the_spectral_axis(gpu):
    """spectral_cube.py:2705-2732: shape_out comes from NAXIS1..3 of the header and reproject_interp
    resamples all three axes; oracle = reproject_separable (pinned against scipy's trilinear call)."""
    rng = np.random.default_rng(3)
    nz, ny, nx = 9, 40, 36
    d = rng.standard_normal((nz, ny, nx)).astype(np.float32)
    d[4, 10:13, 8:11] = np.nan
    cube = SpectralCube.read(d, _hdr(nz, ny, nx))
    # target: rotated 25 degrees, 14 channels of 0.3 km/s starting half a channel before the cube's first
    tgt = _rot(_hdr(14, 44, 38, CDELT3=0.3, CRVAL3=-16.2, CRPIX1=19.0, CRPIX2=22.0), 25.0)
    out = cube.reproject(tgt)
    assert out.shape == (14, 44, 38)
    np.testing.assert_allclose(out.spectral_axis, -16.2 + 0.3 * np.arange(14))
    xs, ys = ops.wcs_pixel_map(cube.wcs, SimpleWCS(tgt), (44, 38))
    zs = ((-16.2 + 0.3 * np.arange(14)) - (-16.0)) / 0.5
    exp, foot = O.reproject_separable(np.where(np.isfinite(d), d, np.nan), xs.get(), ys.get(), zs)
    got = out._device_data().get()
    assert_close(got, exp, atol=1e-5 * np.nanmax(np.abs(exp)), what="3-D reproject")
    assert np.array_equal(out.mask.include(), np.broadcast_to(foot, got.shape))
    assert not foot[0].any() or zs[0] >= -0.5              # the first target channel lies -0.4 channels out: still inside
    # a header in m/s describing the same axis, same pixels: purely spatial, bit-identical to the 2-D form
    same = _rot(_hdr(nz, 44, 38, CUNIT3="m/s", CDELT3=500.0, CRVAL3=-16000.0, CRPIX1=19.0, CRPIX2=22.0), 25.0)
    a = cube.reproject(same)._device_data().get()
    two = {k: v for k, v in same.items() if not (k.endswith("3") or k == "NAXIS")}
    two["NAXIS"] = 2
    o2 = cube.reproject(two)
    assert np.array_equal(a, o2._device_data().get(), equal_nan=True)
    np.testing.assert_allclose(o2.spectral_axis, cube.spectral_axis)          # a 2-axis header keeps the cube's channels
    with pytest.raises(ValueError, match="cannot relate spectral units"):
        cube.reproject(dict(same, CUNIT3="Hz"))


def test_reproject_nearest_neighbour_and_value_check(gpu):
    g = golden("reproject_glue_scipy.npz")
    d = g["data"]
    out, foot = ops.resample_bilinear(DeviceArray.from_numpy(d), g["xs"], g["ys"], order=0)
    assert np.array_equal(out.get(), g["nearest"].astype(np.float32), equal_nan=True)
    assert np.array_equal(foot.get().astype(bool), g["footprint2d"])
    for env in ("0", "1"):                                  # gather kernel alone and LDS kernel give the same bits
        os.environ["SPC_BILINEAR_LDS"] = env
        o2, _ = ops.resample_bilinear(DeviceArray.from_numpy(d), g["xs"], g["ys"], order=0)
        assert np.array_equal(o2.get(), out.get(), equal_nan=True)
    os.environ.pop("SPC_BILINEAR_LDS")
    # through the cube API, against the oracle
    rng = np.random.default_rng(4)
    c = rng.standard_normal((3, 30, 28)).astype(np.float32)
    cube = SpectralCube.read(c, _hdr(3, 30, 28))
    tgt = _rot(_hdr(3, 33, 31, CRPIX1=15.0, CRPIX2=17.0), -40.0)
    r = cube.reproject(tgt, order="nearest-neighbor")
    xs, ys = ops.wcs_pixel_map(cube.wcs, SimpleWCS(tgt), (33, 31))
    exp, _ = O.resample_nearest(c, xs.get(), ys.get())
    assert np.array_equal(r._device_data().get(), exp.astype(np.float32), equal_nan=True)
    with pytest.raises(ValueError, match="order"):           # ('bicubic' / 'biquadratic' are built since round 4: test_gpu_round4.py)
        cube.reproject(tgt, order="quintic")
    # spectral_cube.py:2733-2739 looks at the VALUES: a non-empty footprint over all-NaN data raises too
    nan_cube = SpectralCube.read(np.full((3, 30, 28), np.nan, np.float32), _hdr(3, 30, 28))
    with pytest.raises(ValueError, match="All values in reprojected cube are nan"):
        nan_cube.reproject(tgt)
    masked = cube.with_mask(np.zeros(c.shape, bool))        # everything masked -> filled with NaN -> all NaN
    with pytest.raises(ValueError, match="All values in reprojected cube are nan"):
        masked.reproject(tgt)


@pytest.mark.parametrize("axis", [1, 2])
@pytest.mark.parametrize("order", [3, 4])
def test_moment_orders_above_two_along_spatial_axes(gpu, axis, order):
    """_moments.py:170-193 / dask_spectral_cube.py:1094-1099 for any axis."""
    shape = (7, 33, 52)
    d = synth.gaussian_line_cube(shape, 21) + 1.0
    d[2, 5:9, 7] = np.nan
    inc = np.random.default_rng(2).random(shape) > 0.2
    inc[3, :, 10] = False
    inc[4, 11, :] = False
    cube = SpectralCube.read(d.astype(np.float32), _hdr(*shape)).with_mask(inc)
    got = np.asarray(cube.moment(order=order, axis=axis))
    cen = cube._pix_cen_axis(axis)
    exp = O.moment(d.astype(np.float32), inc, order, cen[None], cube._pix_size_slice(axis), axis=axis)
    with np.errstate(all="ignore"):
        assert_close(got, exp, rtol=1e-9, atol=1e-9 * np.nanmax(np.abs(exp)), what="order %d axis %d" % (order, axis))
    assert np.isnan(got).any()                               # the fully masked rays


def test_write_stores_filled_data(gpu, tmp_path):
    """dask_spectral_cube.py:1405,1502: the HDU holds _get_filled_data(fill=self._fill_value)."""
    from spectral_cube_amd import io_fits
    rng = np.random.default_rng(6)
    d = rng.standard_normal((4, 6, 8)).astype(np.float32) + 3.0
    cube = SpectralCube.read(d, _hdr(4, 6, 8))
    m = cube.with_mask(cube > 3.0)
    m.write(tmp_path / "filled.fits")
    back, _ = io_fits.load_cube(str(tmp_path / "filled.fits"))
    assert np.array_equal(back.get(), np.where(d > 3.0, d, np.nan), equal_nan=True)
    m.with_fill_value(-1.0).write(tmp_path / "fill.fits")
    back, _ = io_fits.load_cube(str(tmp_path / "fill.fits"))
    assert np.array_equal(back.get(), np.where(d > 3.0, d, np.float32(-1.0)))
    m.write(tmp_path / "raw.fits", filled=False)
    back, _ = io_fits.load_cube(str(tmp_path / "raw.fits"))
    assert np.array_equal(back.get(), d)


def test_parent_mask_of_a_smoothed_cube_is_lowered_on_the_device(gpu, monkeypatch):
    """A smoothed cube keeps its parent's mask (dask_spectral_cube.py:836-840).  Once materialised, lowering
    that mask must not copy the parent cube to the host: the predicate runs on the parent's
    device data (spc_mask_include_u8)."""
    shape = (40, 12, 64)
    d = synth.gaussian_line_cube(shape, 14)
    d[5, 3, 3] = np.nan
    k = Gaussian1DKernel(1.5)
    parent = SpectralCube.read(d, _hdr(*shape))
    parent = parent.with_mask(parent > 0.4)
    sm = parent.spectral_smooth(k)
    sm._device_data()                                        # materialise: the fused route is off the table
    monkeypatch.setattr(SpectralCube, "_host_data", lambda self: (_ for _ in ()).throw(AssertionError("host copy")))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = [np.asarray(sm.moment(order=o)) for o in (0, 1, 2)]
    spec = sm._mask_spec()
    assert spec.flags == _lib.MASK_ARRAY and spec.array is not None
    inc = (d > np.float32(0.4)) & np.isfinite(d)
    assert np.array_equal(spec.array.get().astype(bool), inc)
    smo = O.spectral_smooth(d, inc, k.array)
    exp = O.moments012(smo, inc, parent._pix_cen_axis(0), parent._pix_size_slice(0), parent.spectral_axis[0])
    with np.errstate(all="ignore"):
        assert_close(got[0], exp[0], atol=1e-5 * np.nanmax(np.abs(exp[0])), what="m0")
        assert_close(got[1], exp[1], atol=1e-5 * 0.5 * shape[0], what="m1")


def test_nonfinite_thresholds_on_the_device(gpu):
    d = np.random.default_rng(9).standard_normal((6, 5, 8)).astype(np.float32)
    d[1, 1, 1] = -np.inf; d[2, 2, 2] = np.inf; d[3, 3, 3] = np.nan
    cube = SpectralCube(d, header=_hdr(6, 5, 8))
    with np.errstate(invalid="ignore"):
        for m, inc in ((cube > -np.inf, d > -np.inf), (cube < np.inf, d < np.inf), (cube > np.nan, np.zeros(d.shape, bool))):
            c = cube.with_mask(m)
            got = ops.mask_include(c._device_data(), c._mask_spec()).get().astype(bool)
            assert np.array_equal(got, inc)


def test_two_stencils_in_flight_on_two_streams(gpu):
    """No entry point allocates, frees or drains the device (include/spcube_hip.h): a spectral and a spatial
    stencil queued from two host threads on two streams, each with its own workspace, several rounds with
    changing data - both match the oracle every time."""
    shape = (48, 40, 128)
    k1, k2 = Gaussian1DKernel(2.0).array, Gaussian2DKernel(1.5).array
    rng = np.random.default_rng(31)
    datas = []
    for r in range(4):
        d = rng.standard_normal(shape).astype(np.float32)
        if r % 2:
            d[7, 5, 9] = np.nan                              # alternate clean / dirty tiles (speculative passes)
        datas.append(d)
    exp1 = [O.spectral_smooth(d, np.isfinite(d), k1) for d in datas]
    exp2 = [O.spatial_smooth(d, np.isfinite(d), k2) for d in datas]
    spec = ops.MaskSpec(_lib.MASK_FINITE)
    errors = []

    def worker(which):
        try:
            st = Stream(0)
            for r, d in enumerate(datas * 3):
                dev = DeviceArray.from_numpy(d)
                if which == 0:
                    out = ops.spectral_conv(dev, k1, mask=spec, stream=st)
                    exp = exp1[r % 4]
                else:
                    out = ops.spatial_conv(dev, k2, mask=spec, stream=st)
                    exp = exp2[r % 4]
                got = out.get(stream=st)
                assert np.array_equal(np.isnan(got), np.isnan(exp))
                ok = np.isfinite(exp)
                assert np.abs(got[ok] - exp[ok]).max() <= 1e-5 * np.abs(exp[ok]).max()
        except Exception as exc:                             # noqa: BLE001
            errors.append((which, repr(exc)))

    ts = [threading.Thread(target=worker, args=(w,)) for w in (0, 1)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors


DASK_WORKER = r'''
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import numpy as np
import dask, dask.array as da
import oracle_np as O
from spectral_cube_amd.dask_adapter import (SpectralSmoothChunk, SpatialSmoothChunk, MomentChunk, Moments012Chunk,
                                            SpectralInterpolateChunk, SigmaClipChunk)
from spectral_cube_amd.kernels import Gaussian1DKernel, Gaussian2DKernel
rng = np.random.default_rng(5)
d = rng.standard_normal((48, 40, 56)).astype(np.float32)
d[5:9, 3, 7] = np.nan                                     # masked voxels arrive NaN-filled (FilledArrayHandler)
k1, k2 = Gaussian1DKernel(1.5).array, Gaussian2DKernel(1.2).array
def close(a, b, what):
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(np.isnan(a), np.isnan(b)), what
    ok = np.isfinite(b)
    assert np.abs(a[ok] - b[ok]).max() <= 1e-5 * np.abs(b[ok]).max(), what
for sched in ("synchronous", "threads"):                  # dask_spectral_cube.py:259, 278-312
    with dask.config.set(scheduler=sched):
        # apply_function_parallel_spectral -> _map_blocks_to_cube: rechunk (-1, auto, auto), map_blocks(dtype=data.dtype)
        arr = da.from_array(d, chunks=(-1, 16, 24))
        out = da.map_blocks(SpectralSmoothChunk(k1), arr, dtype=arr.dtype).compute()
        close(out, O.spectral_smooth(d, None, k1), "spectral " + sched)
        # apply_function_parallel_spatial: rechunk (auto, -1, -1)
        arr = da.from_array(d, chunks=(10, -1, -1))
        out = da.map_blocks(SpatialSmoothChunk(k2), arr, dtype=arr.dtype).compute()
        close(out, O.spatial_smooth(d, None, k2), "spatial " + sched)
        # a reduced chunk function (tests/test_dask.py:144-169 pattern): drop_axis=[0]
        cen = np.arange(48.0) * 0.5
        arr = da.from_array(d, chunks=(-1, 20, 28))
        m1 = da.map_blocks(MomentChunk(1, cen, 0.5, world0=-3.0), arr, dtype=np.float64, drop_axis=[0],
                           chunks=(arr.chunks[1], arr.chunks[2])).compute()
        e1 = O.moment(d, None, 1, cen, 0.5, world0=-3.0)
        assert np.array_equal(np.isnan(m1), np.isnan(e1)) and np.nanmax(np.abs(m1 - e1)) <= 1e-5 * 24.0
        # round 3: moment 0, 1 and 2 from ONE staging of every chunk (drop_axis + new_axis: a (3, ny, nx) stack)
        m012 = da.map_blocks(Moments012Chunk(cen, 0.5, world0=-3.0), arr, dtype=np.float64, drop_axis=[0], new_axis=[0],
                             chunks=((3,), arr.chunks[1], arr.chunks[2])).compute()
        e012 = O.moments012(d, np.isfinite(d), cen, 0.5, -3.0)
        assert m012.shape == (3, 40, 56)
        for got, exp, sc in zip(m012, e012, (np.nanmax(np.abs(e012[0])), 24.0, np.nanmax(np.abs(e012[2])))):
            assert np.array_equal(np.isnan(got), np.isnan(exp)) and np.nanmax(np.abs(got - exp)) <= 1e-5 * sc
        assert np.array_equal(m012[1], m1, equal_nan=True)
        # float64 chunks (what a float64 FITS cube hands over): converted while they are staged
        out64 = da.map_blocks(SpectralSmoothChunk(k1), da.from_array(d.astype(np.float64), chunks=(-1, 16, 24)), dtype=np.float64).compute()
        assert out64.dtype == np.float64
        close(out64.astype(np.float32), O.spectral_smooth(d, None, k1), "spectral float64 " + sched)
        # interp_wrapper: the chunk grows along the spectral axis
        x = np.arange(48.0); grid = np.linspace(0.0, 47.0, 95)
        out = da.map_blocks(SpectralInterpolateChunk(x, grid), arr, dtype=arr.dtype,
                            chunks=((95,), arr.chunks[1], arr.chunks[2])).compute()
        exp, _ = O.spectral_interpolate(d, None, x, grid)
        close(out, exp.astype(np.float32), "interp " + sched)
print("DASK_OK", dask.__version__)
'''


def test_chunk_functions_under_dask_map_blocks(gpu, tmp_path):
    """The chunk functions under a REAL dask.array.map_blocks, the way _map_blocks_to_cube
    (dask_spectral_cube.py:816-844) calls them: spectral / spatial chunking, a reduced function with
    drop_axis, a chunk that grows along z, under the `synchronous` and the `threads` scheduler (re-entrancy:
    every worker thread gets its own device scratch).  dask lives in the image's conda interpreter only, so the
    test runs there as a subprocess; skipped where that interpreter or dask is missing."""
    import shutil
    import subprocess
    from conftest import REPO
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or subprocess.run([py, "-c", "import dask.array"], capture_output=True).returncode != 0:
        pytest.skip("no interpreter with dask on this box")
    script = tmp_path / "dask_worker.py"
    script.write_text(DASK_WORKER)
    # conda ships an older libstdc++ that scipy would load first: the system one must win for libspcube_hip.so
    env = dict(os.environ)
    sys_cxx = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(sys_cxx):
        env["LD_PRELOAD"] = sys_cxx
    r = subprocess.run([py, "-B", str(script), REPO], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DASK_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("shape", [(7, 5, 9), (40, 3, 70), (64, 2, 130), (100, 3, 200), (130, 2, 70), (260, 3, 67), (515, 2, 37), (1030, 2, 21), (2050, 1, 12), (4100, 1, 5)])
def test_percentile_axis0_every_kernel(gpu, shape, monkeypatch):
    """np.nanmedian / np.nanpercentile of masked rays through every selection kernel: the register-resident rays
    (32 / 16 / 8 spaxels per block and 16 - 128 keys per lane by ray length, lengths that are no multiple of the lanes per ray, tiles hanging
    over the row end; with the early ranking of the surviving candidates and - on the tie-heavy column - without),
    and - by switch - the streaming radix-16 and bisection descents that longer rays fall back to.  Medians are
    bit-exact; MAD (centre per spaxel) goes through the same kernels."""
    import warnings
    from spectral_cube_amd import ops, _lib
    from spectral_cube_amd.device import DeviceArray
    rng = np.random.default_rng(shape[0])
    d = rng.standard_normal(shape).astype(np.float32)
    d[rng.random(shape) < 0.05] = np.nan
    d[:, 0, 0] = np.nan                                             # an empty ray
    d[: shape[0] // 2, 0, 1] = 2.5                                  # ties
    inc = rng.random(shape) < 0.8
    fz = np.where(inc, d, np.nan).astype(np.float32)
    dd = DeviceArray.from_numpy(d)
    spec = ops.MaskSpec(_lib.MASK_FINITE | _lib.MASK_ARRAY, array=DeviceArray.from_numpy(inc.astype(np.uint8)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        emed = np.nanmedian(fz, axis=0)
        e30 = np.nanpercentile(fz.astype(np.float64), 30.0, axis=0)
        emad = np.nanmedian(np.abs(fz - emed[None]), axis=0)
        emed_nomask = np.nanmedian(d, axis=0)
    # (default: 512-thread blocks, descriptor loads; then 64-bit addresses, the 256-thread table with and without descriptors)
    for env in ({}, {"SPC_SELECT_DESC": "0"}, {"SPC_SELECT_SHORT": "0"}, {"SPC_SELECT_BT": "512"}, {"SPC_SELECT_BT": "256", "SPC_SELECT_DESC": "1"}, {"SPC_SELECT_BT": "256", "SPC_SELECT_DESC": "0"},
                {"SPC_SELECT_REG": "0"}, {"SPC_SELECT_REG": "0", "SPC_SELECT_RADIX16": "0"}):
        for k in ("SPC_SELECT_REG", "SPC_SELECT_RADIX16", "SPC_SELECT_BT", "SPC_SELECT_DESC", "SPC_SELECT_SHORT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        med = ops.percentile_axis0(dd, 50.0, mask=spec)
        assert np.array_equal(med.get(), emed, equal_nan=True), env
        assert np.array_equal(ops.percentile_axis0(dd, 50.0).get(), emed_nomask, equal_nan=True), env
        got30 = ops.percentile_axis0(dd, 30.0, mask=spec).get()
        assert np.array_equal(np.isnan(got30), np.isnan(e30))
        np.testing.assert_allclose(got30, e30, rtol=3e-6, atol=1e-7, equal_nan=True)
        mad = ops.percentile_axis0(dd, 50.0, mask=spec, center=med).get()
        assert np.array_equal(mad, emad, equal_nan=True), env


def test_order_statistics_mask_predicates_as_key_intervals(gpu):
    """The register-resident selection and the clip kernel test a sample's validity with ONE unsigned range test on its
    order-preserving key (sel_key_range: NaN, isfinite and every threshold comparison select a key interval).  Every
    comparison x thresholds at the awkward values (+-0, +-inf, NaN, a denormal, FLT_MAX) x data holding +-0, +-inf, NaN,
    denormals: the median of what numpy's comparison keeps (masks.py:670-758 semantics), bit for bit."""
    import warnings
    from spectral_cube_amd import ops, _lib
    from spectral_cube_amd.device import DeviceArray
    rng = np.random.default_rng(77)
    nz, ny, nx = 96, 3, 40
    d = rng.standard_normal((nz, ny, nx)).astype(np.float32)
    # (no +-FLT_MAX samples: numpy's float32 mean of two of them overflows to inf, the float64 interpolation here does not)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1e38, -1e38, 0.3, -0.3], np.float32)
    pick = rng.random(d.shape) < 0.4
    d[pick] = special[rng.integers(0, len(special), int(pick.sum()))]
    dd = DeviceArray.from_numpy(d)
    cmpf = {_lib.MASK_GT: np.greater, _lib.MASK_GE: np.greater_equal, _lib.MASK_LT: np.less, _lib.MASK_LE: np.less_equal}
    thr = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, 3.4028235e38, 0.3, -0.3]
    cases = [(f, t, None) for f in cmpf for t in thr]
    cases += [(_lib.MASK_GT | _lib.MASK_LT, -0.3, 0.3), (_lib.MASK_GE | _lib.MASK_LE, -0.0, 0.0), (_lib.MASK_GE | _lib.MASK_LT, 0.0, np.inf),
              (_lib.MASK_GT | _lib.MASK_LE | _lib.MASK_FINITE, -np.inf, np.inf), (_lib.MASK_FINITE, None, None), (0, None, None)]
    for flags, a, b in cases:
        inc = ~np.isnan(d)
        lo = hi = 0.0
        with np.errstate(invalid="ignore"):
            if flags & _lib.MASK_FINITE:
                inc &= np.isfinite(d)
            for f in (_lib.MASK_GT, _lib.MASK_GE):
                if flags & f:
                    lo = a
                    inc &= cmpf[f](d, np.float32(a))
            for f in (_lib.MASK_LT, _lib.MASK_LE):
                if flags & f:
                    hi = a if b is None else b
                    inc &= cmpf[f](d, np.float32(hi))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            exp = np.nanmedian(np.where(inc, d, np.nan).astype(np.float32), axis=0)
        got = ops.percentile_axis0(dd, 50.0, mask=ops.MaskSpec(flags, lo, hi)).get()
        assert np.array_equal(got, exp, equal_nan=True), (flags, a, b)
        # the clip kernel loads through the same test: nothing clipped (huge sigma) -> the filled cube
        if np.isfinite(d[inc]).all():
            clipped = ops.sigma_clip_axis0(dd, sigma=1e30, mask=ops.MaskSpec(flags, lo, hi), maxiters=1).get()
            assert np.array_equal(clipped, np.where(inc, d, np.nan).astype(np.float32), equal_nan=True), (flags, a, b)


@pytest.mark.parametrize("ntaps,sym", [(9, True), (17, True), (33, True), (33, False), (13, False)])
@pytest.mark.parametrize("shape", [(300, 5, 7), (70, 3, 130)])
def test_spectral_smooth_masked_table_denominators(gpu, shape, ntaps, sym):
    """The general spectral stencil (mask array + NaNs) takes its denominators from validity-bit tables: every ring
    size, symmetric and not, kernels narrower than their ring (13 taps in the 17 ring), tall cubes that are split
    along z (a slice starts mid-cube with an empty history), runs of invalid samples longer than the kernel (empty
    windows -> NaN) and windows hanging over both ends - against the oracle at the contract tolerance, materialised
    and fused with the moments."""
    from spectral_cube_amd import ops, _lib
    from spectral_cube_amd.device import DeviceArray
    rng = np.random.default_rng(ntaps * 7 + shape[0])
    d = rng.standard_normal(shape).astype(np.float32) + 3.0
    d[rng.random(shape) < 0.03] = np.nan
    inc = rng.random(shape) < 0.6
    inc[40:40 + 2 * ntaps, 1, 2] = False                              # an empty window in the middle
    inc[:ntaps, 0, 0] = False                                        # ... and at the start of the ray
    inc[:, 2, 3] = True
    k = np.exp(-0.5 * (np.arange(ntaps) - ntaps // 2) ** 2 / (ntaps / 6.0) ** 2)
    if not sym:
        k = k * np.linspace(0.5, 1.5, ntaps)
    exp = O.spectral_smooth(d, inc, k)
    dd = DeviceArray.from_numpy(d)
    spec = ops.MaskSpec(_lib.MASK_FINITE | _lib.MASK_ARRAY, array=DeviceArray.from_numpy(inc.astype(np.uint8)))
    got = ops.spectral_conv(dd, k, mask=spec).get()
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    ok = ~np.isnan(exp)
    np.testing.assert_allclose(got[ok], exp[ok], rtol=1e-6, atol=1e-6)
    # fused with the moments of the smoothed cube (same mask on the smoothed samples)
    cen = (np.arange(shape[0]) - shape[0] // 2) * 0.5
    r = ops.spectral_conv_moments(dd, k, DeviceArray.from_numpy(cen), mask=spec, cen_host=cen, want=("m0", "m1"))
    einc = inc & np.isfinite(d)             # the smoothed cube keeps the parent's mask: array AND isfinite(parent data)
    e0 = O.moment(exp.astype(np.float32), einc, 0, cen, 1.0)
    e1 = O.moment(exp.astype(np.float32), einc, 1, cen, 1.0)
    assert_close(r["m0"].get(), e0, atol=1e-5 * np.nanmax(np.abs(e0)), what="fused m0")
    with np.errstate(all="ignore"):
        wc = np.abs(e0) > 1e-3 * np.nanmax(np.abs(e0))
    assert_close(np.where(wc, r["m1"].get(), 0.0), np.where(wc, e1, 0.0), atol=1e-5 * np.abs(cen).max(), what="fused m1")


@pytest.mark.parametrize("ntaps", [35, 41, 49, 51, 65])
def test_spectral_smooth_wide_symmetric_rings(gpu, ntaps, monkeypatch):
    """35 - 65 symmetric taps (Gaussian1DKernel(4.25 ... 8)): the all-valid pass runs on the 49- / 65-tap rings, tiles
    holding a NaN are redone by the runs-of-16 kernel; both against the oracle at the contract tolerance and against
    each other (SPC_CONV_FAST=0: the runs-of-16 kernel alone) to an ulp of float32."""
    from spectral_cube_amd import ops, _lib
    from spectral_cube_amd.device import DeviceArray
    shape = (3 * ntaps + 5, 6, 300)
    rng = np.random.default_rng(ntaps)
    d = rng.standard_normal(shape).astype(np.float32) + 1.0
    d[ntaps, 2, 17] = np.nan                 # one dirty 128-column tile, the others stay on the ring
    d[0, 5, 299] = np.nan
    k = np.exp(-0.5 * ((np.arange(ntaps) - ntaps // 2) / (ntaps / 8.0)) ** 2)
    exp = O.spectral_smooth(d, np.isfinite(d), k)
    dd = DeviceArray.from_numpy(d)
    spec = ops.MaskSpec(_lib.MASK_FINITE)
    got = ops.spectral_conv(dd, k, mask=spec).get()
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=1e-6, equal_nan=True)
    monkeypatch.setenv("SPC_CONV_FAST", "0")
    ref = ops.spectral_conv(dd, k, mask=spec).get()
    monkeypatch.delenv("SPC_CONV_FAST")
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.all(np.abs(got[ok] - ref[ok]) <= np.spacing(np.abs(ref[ok]).astype(np.float32)))
    assert np.mean(got[ok] != ref[ok]) < 0.01          # different float64 summation order: rare float32 flips only


@pytest.mark.parametrize("shape", [(9, 4, 11), (60, 3, 70), (100, 3, 150), (200, 2, 70), (515, 2, 37), (1030, 2, 21), (2100, 2, 9)])
def test_sigma_clip_fused_kernel(gpu, shape, monkeypatch):
    """sigma clipping with the rays resident in registers (one kernel for all iterations) against the oracle's
    restatement of astropy.stats.sigma_clip(axis=0) and against the loop of separate kernels (the same clipped set:
    both carry the sums in float64 and the bounds in float32): median and mean centres, std and mad_std spreads, asymmetric sigmas, iteration
    caps, a mask array, rays that are empty, constant, or hold infinities."""
    from spectral_cube_amd import ops, _lib
    from spectral_cube_amd.device import DeviceArray
    rng = np.random.default_rng(shape[0] + 1)
    d = rng.standard_normal(shape).astype(np.float32)
    d[rng.random(shape) < 0.03] *= 12.0                              # outliers to clip
    d[rng.random(shape) < 0.02] = np.nan
    d[:, 0, 0] = np.nan                                              # empty ray
    d[:, 0, 1] = 4.25                                                # constant ray: std 0, nothing outside [c, c]
    d[0, 1, 2] = np.inf                                              # an infinity poisons the ray's std: nothing is clipped
    inc = rng.random(shape) < 0.85
    dd = DeviceArray.from_numpy(d)
    spec = ops.MaskSpec(_lib.MASK_ARRAY, array=DeviceArray.from_numpy(inc.astype(np.uint8)))
    for kw in (dict(), dict(maxiters=1), dict(maxiters=None), dict(sigma_lower=1.5, sigma_upper=4.0), dict(cenfunc="mean"),
               dict(stdfunc="mad_std"), dict(stdfunc="mad_std", cenfunc="mean", maxiters=3)):
        sig = 2.5
        monkeypatch.delenv("SPC_SIGMA_CLIP_FUSED", raising=False)
        got = ops.sigma_clip_axis0(dd, sigma=sig, mask=spec, **kw).get()
        for bt in ("256", "512"):                                    # (both block shapes, whatever the launcher would pick)
            monkeypatch.setenv("SPC_SIGMA_BT", bt)
            assert np.array_equal(ops.sigma_clip_axis0(dd, sigma=sig, mask=spec, **kw).get(), got, equal_nan=True), (kw, bt)
        monkeypatch.delenv("SPC_SIGMA_BT")
        monkeypatch.setenv("SPC_SIGMA_SHORT", "0")                   # (and without the few-lanes-per-ray table of short rays)
        assert np.array_equal(ops.sigma_clip_axis0(dd, sigma=sig, mask=spec, **kw).get(), got, equal_nan=True), kw
        monkeypatch.delenv("SPC_SIGMA_SHORT")
        monkeypatch.setenv("SPC_SIGMA_CLIP_FUSED", "0")
        ref = ops.sigma_clip_axis0(dd, sigma=sig, mask=spec, **kw).get()
        assert np.array_equal(got, ref, equal_nan=True), kw
        exp = O.sigma_clip(d, inc & ~np.isnan(d), sig, **kw)
        assert np.mean(np.isnan(got) != np.isnan(exp)) < 2e-4, kw    # float32-vs-float64 bounds: borderline samples only
        ok = ~np.isnan(got) & ~np.isnan(exp)
        assert np.array_equal(got[ok], exp[ok])


@pytest.mark.parametrize("shape", [(1, 68, 76), (7, 33, 100), (40, 9, 260)])
def test_fused_smooth_moments_rows_across_flag_tiles(gpu, shape):
    """smooth -> moments on a cube whose rows are no multiple of 128 spaxels, with a few NaNs: the algebraic pass works
    in row segments, the tile flags that hand spaxels to the stencil kernel are per 128 LINEAR spaxels - a segment that
    quits because of a NaN lies in up to two flag tiles and must flag both (a clean neighbouring tile used to keep
    whatever the output buffer held).  The output buffers are poisoned first."""
    from spectral_cube_amd import ops
    from spectral_cube_amd.device import DeviceArray
    rng = np.random.default_rng(shape[2])
    nz = shape[0]
    d = (rng.standard_normal(shape) * 3 + 1).astype(np.float32)
    d[rng.random(shape) < 0.02 / max(nz // 4, 1)] = np.nan
    k = np.abs(rng.standard_normal(15)) + 0.05
    cen = np.cumsum(rng.uniform(0.5, 1.5, nz))
    cref = cen[nz // 2]
    sm = O.spectral_smooth(d, None, k)
    e0 = O.moment(sm, None, 0, cen, 1.3)
    e1 = O.moment(sm, None, 1, cen, 1.3, world0=10.0)
    out = {"m0": DeviceArray.from_numpy(np.full(shape[1:], 1e30)), "m1": DeviceArray.from_numpy(np.full(shape[1:], 1e30))}
    r = ops.spectral_conv_moments(DeviceArray.from_numpy(d), k, DeviceArray.from_numpy(cen - cref), dv=1.3, m1_add=cref + 10.0,
                                  want=("m0", "m1"), cen_host=cen - cref, out=out)
    assert_close(r["m0"].get(), e0, atol=1e-5 * np.nanmax(np.abs(e0)), what="fused m0")
    with np.errstate(all="ignore"):
        wc = np.abs(e0) > 1e-2 * np.nanmax(np.abs(e0))
    assert_close(np.where(wc, r["m1"].get(), 0.0), np.where(wc, e1, 0.0), atol=1e-5 * max(cen[-1] - cen[0], 1.0), what="fused m1")


@pytest.mark.parametrize("taps", [(41, 41), (57, 57), (41, 35), (65, 65)])
def test_spatial_smooth_wide_rings(gpu, taps):
    """separable spatial kernels of 35 - 65 taps (the 49- and 65-tap rings): the all-valid pass (isotropic kernels
    only), a plane with a NaN (its strip is redone by the general kernel), a mask array, and an anisotropic pair of
    kernels that has no all-valid kernel - all against the oracle at the contract tolerance."""
    from spectral_cube_amd import ops, _lib
    from spectral_cube_amd.device import DeviceArray
    ny_t, nx_t = taps
    rng = np.random.default_rng(ny_t + nx_t)
    shape = (3, 90, 900)       # 900 columns: 3 strips of the 65-tap all-valid kernel, 2 of the 33-tap one (workspace bound)
    d = rng.standard_normal(shape).astype(np.float32) + 2.0
    d[1, 40, 300] = np.nan
    gy = np.exp(-0.5 * ((np.arange(ny_t) - ny_t // 2) / (ny_t / 8.0)) ** 2)
    gx = np.exp(-0.5 * ((np.arange(nx_t) - nx_t // 2) / (nx_t / 8.0)) ** 2)
    k2 = np.outer(gy, gx)
    dd = DeviceArray.from_numpy(d)
    exp = O.spatial_smooth(d, np.isfinite(d), k2)
    got = ops.spatial_conv(dd, k2, mask=ops.MaskSpec(_lib.MASK_FINITE)).get()
    assert_close(got, exp, atol=1e-5 * np.nanmax(np.abs(exp)), what="finite mask %s" % (taps,))
    inc = rng.random(shape) < 0.7
    expm = O.spatial_smooth(d, inc & np.isfinite(d), k2)
    gotm = ops.spatial_conv(dd, k2, mask=ops.MaskSpec(_lib.MASK_FINITE | _lib.MASK_ARRAY, array=DeviceArray.from_numpy(inc.astype(np.uint8)))).get()
    assert_close(gotm, expm, atol=1e-5 * np.nanmax(np.abs(expm)), what="array mask %s" % (taps,))


def test_nonseparable_stencil_all_valid_pass(gpu, monkeypatch):
    """rotated elliptical kernel (what convolve_to builds): the all-valid pass of the tiled 2-D stencil packs two columns
    per FMA and hands flagged tiles to the (num, den) kernel - a plane with one NaN (one 128 x 32 tile and the tiles
    whose halo reaches it are redone), a clean plane, image sizes that leave partial tiles; against the oracle and
    against the (num, den) kernel alone (SPC_CONV_FAST=0) to float32 rounding."""
    from spectral_cube_amd import ops, _lib
    from spectral_cube_amd.device import DeviceArray
    rng = np.random.default_rng(8)
    shape = (2, 75, 330)
    d = rng.standard_normal(shape).astype(np.float32) + 1.0
    d[1, 40, 200] = np.nan
    yy, xx = np.mgrid[-6:7, -6:7]
    kn = np.exp(-0.5 * (((xx + 0.5 * yy) / 2.0) ** 2 + (yy / 1.2) ** 2))
    dd = DeviceArray.from_numpy(d)
    exp = O.spatial_smooth(d, np.isfinite(d), kn)
    out = DeviceArray.from_numpy(np.full(shape, 1e30, np.float32))
    got = ops.spatial_conv(dd, kn, mask=ops.MaskSpec(_lib.MASK_FINITE), out=out).get()
    assert_close(got, exp, atol=1e-5 * np.nanmax(np.abs(exp)), what="two-pass 2-D stencil")
    monkeypatch.setenv("SPC_CONV_FAST", "0")
    ref = ops.spatial_conv(dd, kn, mask=ops.MaskSpec(_lib.MASK_FINITE)).get()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.nanmax(np.abs(got - ref)) <= 4e-7 * np.nanmax(np.abs(ref))


@pytest.mark.parametrize("shape", [(5, 7, 9), (3, 40, 515), (2, 9, 2050), (2, 3, 4100)])
def test_order_statistics_along_x_without_transpose(gpu, shape):
    """median / percentile / mad_std with axis=2: the rows are the rays (lanes walk along the contiguous samples, no
    transposed copy of the cube); rows of more than 4096 samples fall back to the transposed copy.  Against numpy on
    the masked cube, medians and MADs bit for bit."""
    import warnings
    from spectral_cube_amd import SpectralCube
    rng = np.random.default_rng(shape[2])
    d = rng.standard_normal(shape).astype(np.float32)
    d[rng.random(shape) < 0.05] = np.nan
    d[0, 0, :] = np.nan
    inc = rng.random(shape) < 0.8
    hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3, "CDELT3": 1.0,
           "CUNIT3": "km/s", "CRPIX1": 1, "CRPIX2": 1, "CRPIX3": 1, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": 0.0, "BUNIT": "K"}
    cube = SpectralCube.read(d, hdr).with_mask(inc)
    fz = np.where(inc, d, np.nan).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        emed = np.nanmedian(fz, axis=2)
        e30 = np.nanpercentile(fz.astype(np.float64), 30.0, axis=2)
        emad = np.nanmedian(np.abs(fz - emed[:, :, None]), axis=2) * np.float32(1.482602218505602)
        assert np.array_equal(np.asarray(cube.median(axis=2)), emed, equal_nan=True)
        np.testing.assert_allclose(np.asarray(cube.percentile(30.0, axis=2)), e30, rtol=3e-6, atol=1e-7, equal_nan=True)
        np.testing.assert_allclose(np.asarray(cube.mad_std(axis=2)), emad, rtol=3e-7, equal_nan=True)
