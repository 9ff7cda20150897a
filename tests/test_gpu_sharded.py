"""GPU: the sharded drivers of spectral_cube_amd.distributed with the ranks run ONE AFTER THE OTHER on
this GPU (8-GPU runs belong to the driver).  What is checked: the strips stitched in rank order equal
the unsharded result - bit for bit, since every strip runs the same kernels on the same numbers - and
that a rank only touches the rows it needs."""
import numpy as np
import pytest

import oracle_np as O
from conftest import assert_close, golden
from spectral_cube_amd import SpectralCube, SimpleWCS, Gaussian1DKernel, ops, synth
from spectral_cube_amd import distributed as D
from spectral_cube_amd.device import DeviceArray

pytestmark = pytest.mark.gpu


class ReplayComm:
    """stands in for the all-gather when the ranks run serially: pass 1 records every rank's strips,
    pass 2 hands the stitched maps back (the drivers are deterministic, so each runs twice)"""

    def __init__(self, world_size):
        self.world_size, self.rank = world_size, 0
        self.record, self.strips, self.calls = True, {}, 0

    def start(self, rank, record):
        self.rank, self.record, self.calls = rank, record, 0

    def allgather_rows(self, strip, ny_total):
        key, self.calls = self.calls, self.calls + 1
        if self.record:
            self.strips.setdefault(key, {})[self.rank] = np.array(strip)
            return np.full((ny_total,) + strip.shape[1:], np.nan)
        return np.concatenate([self.strips[key][r] for r in range(self.world_size)], axis=0)[:ny_total]


def _hdr(nz, ny, nx):
    return {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3,
            "CDELT3": 0.5, "CUNIT3": "km/s", "CRPIX1": nx / 2, "CRPIX2": ny / 2, "CRPIX3": 1, "CRVAL1": 10.0,
            "CRVAL2": 20.0, "CRVAL3": -16.0, "BUNIT": "K", "NAXIS1": nx, "NAXIS2": ny, "NAXIS3": nz}


def _strip_cube(d, inc, hdr, y0, y1):
    h = dict(hdr, CRPIX2=hdr["CRPIX2"] - y0, NAXIS2=y1 - y0)
    c = SpectralCube.read(np.ascontiguousarray(d[:, y0:y1]), h)
    return c.with_mask(np.ascontiguousarray(inc[:, y0:y1])) if inc is not None else c


@pytest.mark.parametrize("masked", [False, True])
@pytest.mark.parametrize("ws", [1, 2, 3])
def test_sharded_spectral_smooth_moment_c3(gpu, ws, masked):
    """configs[2] on row strips: fused spectral_smooth -> moment per strip + one stitch."""
    shape = (96, 22, 64)
    d = synth.gaussian_line_cube(shape, 41)
    d[10:14, 5, 7] = np.nan
    inc = synth.boolean_mask(d, 41).astype(bool) if masked else None
    hdr = _hdr(*shape)
    k = Gaussian1DKernel(2.0)
    whole = _strip_cube(d, inc, hdr, 0, shape[1]).spectral_smooth(k)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = {o: np.asarray(whole.moment(order=o)) for o in (0, 1, 2)}
    comm = ReplayComm(ws)
    got = None
    for record in (True, False):
        for r in range(ws):
            y0, y1 = D.strip_bounds(shape[1], ws, r)
            comm.start(r, record)
            res = D.sharded_spectral_smooth_moment(_strip_cube(d, inc, hdr, y0, y1), k, shape[1], comm, orders=(0, 1, 2))
            if not record:
                if got is None:
                    got = res
                for o in (0, 1, 2):                     # every rank holds the same stitched maps
                    assert np.array_equal(res[o], got[o], equal_nan=True)
    for o in (0, 1, 2):
        assert np.array_equal(got[o], exp[o], equal_nan=True), "order %d" % o
    # and against the oracle at the contract tolerance
    einc = inc if masked else np.isfinite(d)       # (the smoothed cube keeps the parent's isfinite mask)
    sm = O.spectral_smooth(d, einc, k.array)
    ref = whole
    e1 = O.moment(sm, einc, 1, ref._pix_cen_axis(0), ref._pix_size_slice(0), world0=ref.spectral_axis[0])
    with np.errstate(all="ignore"):
        assert_close(got[1], e1, atol=1e-5 * 0.5 * shape[0], what="C3 sharded moment1")


@pytest.mark.parametrize("ws", [1, 3, 4])
def test_sharded_interpolate_then_reproject_c5(gpu, ws):
    """configs[4] on output-row strips: every rank loads only the source rows its strip of the rotated
    grid touches, interpolates them spectrally (per spaxel, no exchange) and resamples; the stitched
    strips equal the unsharded spectral_interpolate -> reproject bit for bit (both in one pass: the interpolation folded
    into the resampling kernel; the two-pass strips agree to float32 rounding)."""
    g = golden("wcs.npz")
    rng = np.random.default_rng(12)
    nz, ny, nx = 12, 96, 80
    d = rng.standard_normal((nz, ny, nx)).astype(np.float32)
    d[3, 40:44, 30:33] = np.nan
    hin = SimpleWCS(str(g["rp_hdr_in"])).header
    hin = dict(hin, NAXIS1=nx, NAXIS2=ny, NAXIS3=nz, CRPIX1=nx / 2, CRPIX2=ny / 2)
    cube = SpectralCube.read(d, hin)
    hout = {k: v for k, v in SimpleWCS(str(g["rp_hdr_out"])).header.items() if not (k.endswith("3") or k == "NAXIS")}
    hout = dict(hout, NAXIS=2, NAXIS1=70, NAXIS2=90, CRPIX1=35.0, CRPIX2=45.0)       # celestial target: channels kept
    grid = np.linspace(cube.spectral_axis[0], cube.spectral_axis[-1], 2 * nz)
    whole = cube.spectral_interpolate(grid, suppress_smooth_warning=True).reproject(hout)
    exp, efoot = whole._device_data().get(), whole._footprint
    assert np.isfinite(exp).any() and not efoot.all()
    outs, outs2, foots, loaded = [], [], [], []
    lo, t, inv, _, _, _ = ops.lerp_plan(cube.spectral_axis, grid)
    for r in range(ws):
        r0, r1 = D.reproject_source_rows(cube.wcs, cube.shape, hout, r, ws)
        loaded.append(r1 - r0)
        src = _strip_cube(d, None, hin, r0, r1)                        # "load" rows [r0, r1) only
        out, foot, (y0, y1) = D.sharded_reproject(src._device_data(), r0, cube.wcs, cube.shape, hout, r, ws,
                                                  mask=src._mask_spec(), fill=np.nan, lerp=(lo, t, inv))
        assert out.shape == (2 * nz, y1 - y0, 70)
        outs.append(out.get()); foots.append(foot.get())
        up = D.sharded_spectral_interpolate(src, grid, suppress_smooth_warning=True)
        out2, foot2, _ = D.sharded_reproject(up._device_data(), r0, cube.wcs, cube.shape, hout, r, ws,
                                             mask=up._mask_spec(), fill=np.nan)
        assert np.array_equal(foot2.get(), foot.get())
        outs2.append(out2.get())
    got, gfoot = np.concatenate(outs, axis=1), np.concatenate(foots, axis=0).astype(bool)
    assert np.array_equal(gfoot, efoot)
    assert np.array_equal(got, exp, equal_nan=True)
    two = np.concatenate(outs2, axis=1)
    assert np.array_equal(np.isnan(two), np.isnan(exp))
    assert np.nanmax(np.abs(two - exp)) <= 2e-6 * np.nanmax(np.abs(exp))
    if ws > 1:
        assert max(loaded) < ny, "a strip of a 30-degree rotated grid does not need every source row"


def test_reproject_source_rows_margins(gpu):
    """identity mapping: the strip of output rows [y0, y1) needs source rows [y0 - 1, y1 + 1) (one row of
    margin for the +1 neighbour and so that a strip edge is not taken for the image border)"""
    hdr = _hdr(4, 60, 50)
    w = SimpleWCS(hdr)
    for r in range(3):
        y0, y1 = D.strip_bounds(60, 3, r)
        r0, r1 = D.reproject_source_rows(w, (4, 60, 50), hdr, r, 3)
        assert r0 <= max(0, y0 - 1) and r1 >= min(60, y1 + 1) and (r1 - r0) <= (y1 - y0) + 4
    far = dict(hdr, CRVAL1=200.0)
    assert D.reproject_source_rows(w, (4, 60, 50), far, 0, 2) == (0, 0)
    out, foot, _ = D.sharded_reproject(DeviceArray((4, 0, 50), np.float32), 0, w, (4, 60, 50), far, 0, 2)
    assert np.isnan(out.get()).all() and not foot.get().any()


class ThreadTransport:
    """the rendezvous protocol between threads of this process (every rank is a thread driving the same GPU)"""

    def __init__(self, world_size):
        import threading
        self.world_size = world_size
        self._slots = {}
        self._cv = threading.Condition()

    def view(self, rank):
        outer = self

        class _View:
            world_size = outer.world_size

            def __init__(self):
                self.rank, self._seq = rank, 0

            def allgather_bytes(self, payload):
                seq, self._seq = self._seq, self._seq + 1
                with outer._cv:
                    outer._slots.setdefault(seq, {})[self.rank] = bytes(payload)
                    outer._cv.notify_all()
                    assert outer._cv.wait_for(lambda: len(outer._slots[seq]) == outer.world_size, timeout=60)
                    return [outer._slots[seq][r] for r in range(outer.world_size)]
        return _View()


@pytest.mark.parametrize("ws", [1, 2, 3])
def test_sharded_percentile_whole_cube(gpu, ws):
    """median / percentile / MAD of a cube whose row strips belong to different ranks: every rank histograms its
    strip on the device (spc_key_histogram_f32), the counters are added through the transport, all ranks return
    the value np.nanpercentile gives for the whole cube (median and MAD bit-exact)."""
    import threading
    import warnings
    from spectral_cube_amd import ops, _lib
    rng = np.random.default_rng(5)
    shape = (33, 41, 57)
    d = rng.standard_normal(shape).astype(np.float32)
    d[rng.random(shape) < 0.02] = np.nan
    d[3, 4, 5], d[6, 7, 8] = np.inf, -np.inf
    inc = rng.random(shape) < 0.7
    fz = np.where(inc, d, np.nan).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exp = {q: float(np.nanpercentile(fz.astype(np.float64), q)) for q in (0.0, 30.0, 50.0, 100.0)}
        med = float(np.nanmedian(fz))
        emad = float(np.nanmedian(np.abs(fz - np.float32(med))))
    tt = ThreadTransport(ws)
    results, errors = [None] * ws, []

    def rank_main(r):
        try:
            tr = tt.view(r)
            y0, y1 = D.strip_bounds(shape[1], ws, r)
            strip = DeviceArray.from_numpy(np.ascontiguousarray(d[:, y0:y1]))
            spec = ops.MaskSpec(_lib.MASK_ARRAY, array=DeviceArray.from_numpy(np.ascontiguousarray(inc[:, y0:y1]).astype(np.uint8)))
            out = {q: D.sharded_percentile(strip, q, tr, mask=spec) for q in exp}
            out["mad"] = D.sharded_percentile(strip, 50.0, tr, mask=spec, center=out[50.0])
            results[r] = out
        except Exception as e:            # noqa: BLE001 - reported by the main thread
            errors.append(e)
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(ws)]
    for t in threads: t.start()
    for t in threads: t.join(120)
    assert not errors, errors
    for r in range(ws):
        np.testing.assert_equal(results[r], results[0])          # (NaN == NaN: the lerp between two infinities)
    assert results[0][50.0] == med and results[0]["mad"] == emad
    for q, e in exp.items():
        np.testing.assert_allclose(results[0][q], e, rtol=2e-7, atol=0)
    # against the one-GPU entry point
    whole = DeviceArray.from_numpy(d)
    wspec = ops.MaskSpec(_lib.MASK_ARRAY, array=DeviceArray.from_numpy(inc.astype(np.uint8)))
    assert ops.percentile_global(whole, 30.0, mask=wspec) == results[0][30.0]


def test_chunked_moments_hide_the_stitch(gpu):
    """distributed.ChunkedMoments: the rank's rows in blocks, the grouped RCCL all-gather of a block's three maps on a
    second stream under the kernel of the next block.  With one rank (RCCL initialises on a 1-GPU box) the gathered
    maps must equal the one-launch moments of the whole strip bit for bit; called twice (events are reused)."""
    from spectral_cube_amd import ops, _lib
    from spectral_cube_amd.device import Stream
    from spectral_cube_amd.rendezvous import SingleProcess
    shape = (24, 64, 40)
    d = synth.gaussian_line_cube(shape, 7)
    m = synth.boolean_mask(d, 7)
    dd, dm = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m)
    cen = np.arange(shape[0]) * 500.0
    cref = cen[shape[0] // 2]
    d_cen = DeviceArray.from_numpy(cen - cref)
    whole = ops.moments(dd, d_cen, dv=500.0, m1_add=cref - 3.0, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=dm), want=("m0", "m1", "m2"))
    try:
        comm = D.RcclComm(0, SingleProcess())
    except Exception as exc:                       # pragma: no cover - RCCL missing on the box
        pytest.skip("RCCL did not initialise: %s" % exc)
    try:
        cm = D.ChunkedMoments(dd, dm, d_cen, 500.0, cref - 3.0, comm, chunks=4)
        s1, s2 = Stream(0), Stream(0)
        for _ in range(2):
            maps = cm(s1, s2)
            s1.synchronize()
            for k in ("m0", "m1", "m2"):
                assert np.array_equal(maps[k].get(), whole[k].get(), equal_nan=True), k
        assert cm.global_rows(0, 2) == (32, 48)
        with pytest.raises(ValueError):
            D.ChunkedMoments(dd, dm, d_cen, 500.0, 0.0, comm, chunks=5)
    finally:
        comm.close()


def test_rccl_collectives_with_one_rank_on_this_gpu(gpu):
    """The device side of the stitch for real - RCCL communicator, spc_allgather_rows, the grouped all-gathers of
    ChunkedMoments on a second stream behind stream events - with the one rank a 1-GPU box allows (RCCL refuses two ranks on
    one device): rendezvous-carried unique id, comm init, in-order map assembly, sharded_moments through RcclComm; all
    equal to the unsharded kernels, bit for bit.  (Multi-rank ordering is the driver's 8-GPU tier.)"""
    from spectral_cube_amd import _lib, ops
    from spectral_cube_amd.device import Stream
    from spectral_cube_amd.rendezvous import SingleProcess
    shape = (128, 256, 96)
    d = synth.gaussian_line_cube(shape, 43)
    d[5:9, 17, 3] = np.nan
    inc = synth.boolean_mask(d, 43)
    comm = D.RcclComm(0, SingleProcess())
    try:
        cube, maskd = DeviceArray.from_numpy(d), DeviceArray.from_numpy(inc)
        v = synth.spectral_axis(shape[0])
        cen = v - v[0]
        cref = cen[shape[0] // 2]
        d_cen = DeviceArray.from_numpy(cen - cref)
        mspec = ops.MaskSpec(_lib.MASK_ARRAY, array=maskd)
        ref = ops.moments(cube, d_cen, dv=500.0, m1_add=cref + v[0], mask=mspec, want=("m0", "m1", "m2"))
        ch = D.ChunkedMoments(cube, maskd, d_cen, 500.0, cref + v[0], comm, chunks=4)
        s1, s2 = Stream(0), Stream(0)
        for _ in range(3):                                   # back to back: the events of one call are reused by the next
            maps = ch(s1, s2)
        s1.synchronize()
        for k in ("m0", "m1", "m2"):
            assert np.array_equal(maps[k].get(), ref[k].get(), equal_nan=True), k
        assert [ch.global_rows(0, c) for c in range(4)] == [(0, 64), (64, 128), (128, 192), (192, 256)]
        # the plain all-gather and the driver that uses it
        strip = ref["m1"]
        got = comm.allgather_rows(strip, shape[1]).get()
        assert np.array_equal(got, ref["m1"].get(), equal_nan=True)
        sc = SpectralCube.read(d, _hdr(*shape)).with_mask(inc.astype(bool))
        sm = D.sharded_moments(sc, shape[1], comm)
        e = sc.moments012()
        for o in (0, 1, 2):
            assert np.array_equal(sm[o], np.asarray(e[o]), equal_nan=True)
    finally:
        comm.close()
