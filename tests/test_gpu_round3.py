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
    with pytest.raises(NotImplementedError):
        ops.wcs_pixel_map(SimpleWCS(dict(SimpleWCS(str(g["in1"]), naxis=2).header, EQUINOX=1950.0), naxis=2),
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
