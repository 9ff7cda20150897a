"""GPU: masked separable spatial_smooth with the denominator on the matrix cores, fused with moment 0
(spc_spatial_conv_sep_mfma_f32).  Oracle: oracle_np.spatial_smooth (astropy semantics, float64) and its nansum."""
import numpy as np
import pytest

import oracle_np as O
from conftest import assert_close
from spectral_cube_amd import Gaussian2DKernel, _lib, ops
from spectral_cube_amd.device import DeviceArray

pytestmark = pytest.mark.gpu


def _case(shape, seed, valid=0.8, nan_frac=0.0):
    rng = np.random.default_rng(seed)
    d = (rng.standard_normal(shape) + 2.0).astype(np.float32)
    m = (rng.random(shape) < valid)
    if nan_frac:
        d[rng.random(shape) < nan_frac] = np.nan
    return d, m


@pytest.mark.parametrize("form", [3, 2, 1])
@pytest.mark.parametrize("shape, fwhm, valid", [((3, 40, 64), 8.0, 0.8), ((5, 33, 130), 8.0, 0.5), ((2, 70, 962), 8.0, 0.05),
                                                 ((4, 16, 480), 4.0, 0.9), ((3, 50, 1000), 8.0, 1.0), ((5, 33, 144), 8.0, 0.5),
                                                 ((2, 70, 976), 8.0, 0.05), ((3, 97, 1040), 8.0, 1.0)])
def test_mfma_smoothed_cube_against_the_oracle(gpu, shape, fwhm, valid, form, monkeypatch):
    # form 3 (round 5: every product on the fp16 matrix instruction) takes widths that are multiples of 4, form 2 (numerator
    # on the float32 matrix instruction) multiples of 16; form 1 (numerator on the vector ALU) any even width - and is what
    # the entry point falls back to
    monkeypatch.setenv("SPC_SPATIAL_MFMA_FORM", str(form))
    """the smoothed cube itself (d_out): 1e-5 of the data range, NaN pattern identical; includes partial strips, bands
    that cross both plane edges, a sparse mask (windows with ONE valid far-tail sample: the fp16 hi/lo denominator must
    hold 1e-6 there) and an all-valid mask array"""
    d, m = _case(shape, 3, valid=valid)
    k2 = Gaussian2DKernel(fwhm / 2.3548200450309493).array
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    out, _ = ops.spatial_conv_mfma(cube, k2, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk))
    exp = O.spatial_smooth(d, m, k2)
    assert_close(out.get(), exp.astype(np.float32), atol=1e-5 * np.nanmax(np.abs(exp)), what="mfma smooth %s" % (shape,))


@pytest.mark.parametrize("form, nx", [(3, 528), (3, 532), (2, 528), (1, 530), (1, 528)])
def test_mfma_fused_moment0_against_the_oracle(gpu, form, nx, monkeypatch):
    """d_m0 without d_out: dv * nansum over channels of the smoothed cube under the ORIGINAL mask; all-masked spaxels NaN;
    NaN samples under a true mask bit are interpolated over by the convolution AND counted by the (array-only) mask"""
    monkeypatch.setenv("SPC_SPATIAL_MFMA_FORM", str(form))
    shape = (37, 45, nx)
    d, m = _case(shape, 9, valid=0.7, nan_frac=0.01)
    m[:, 3:6, 10:14] = False
    k2 = Gaussian2DKernel(8 / 2.3548200450309493).array
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    for flags, inc in ((_lib.MASK_ARRAY, m), (_lib.MASK_ARRAY | _lib.MASK_FINITE, m & np.isfinite(d))):
        _, m0 = ops.spatial_conv_mfma(cube, k2, mask=ops.MaskSpec(flags, array=mk), want_cube=False, want_m0=True, dv=500.0)
        sm = O.spatial_smooth(d, m, k2)                      # (NaN samples are invalid for the convolution either way)
        filled = np.where(inc, sm, np.nan)
        exp = 500.0 * np.nansum(filled, axis=0)
        exp[np.all(np.isnan(filled), axis=0)] = np.nan
        assert np.isnan(exp).sum() >= 12
        assert_close(m0.get(), exp, atol=1e-5 * np.nanmax(np.abs(exp)), what="mfma moment0 flags %d" % flags)


def test_mfma_many_channels_several_chunks(gpu):
    """more channels than a chunk holds: float32 sums inside a chunk, float64 across chunks"""
    shape = (150, 16, 96)
    d, m = _case(shape, 5, valid=0.6)
    k2 = Gaussian2DKernel(2.0).array                         # 17 taps: padded to the 29-tap ring
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    out, m0 = ops.spatial_conv_mfma(cube, k2, mask=ops.MaskSpec(_lib.MASK_ARRAY, array=mk), want_cube=True, want_m0=True, dv=2.0)
    sm = O.spatial_smooth(d, m, k2)
    assert_close(out.get(), sm.astype(np.float32), atol=1e-5 * np.nanmax(np.abs(sm)), what="mfma smooth 17 taps")
    exp = 2.0 * np.nansum(np.where(m, sm, np.nan), axis=0)
    assert_close(m0.get(), exp, atol=1e-5 * np.nanmax(np.abs(exp)), what="mfma moment0 chunks")


def test_mfma_refuses_what_it_does_not_do(gpu):
    d, m = _case((2, 16, 32), 1)
    cube, mk = DeviceArray.from_numpy(d), DeviceArray.from_numpy(m.astype(np.uint8))
    spec = ops.MaskSpec(_lib.MASK_ARRAY, array=mk)
    with pytest.raises(_lib.HipUnsupported):
        ops.spatial_conv_mfma(cube, Gaussian2DKernel(10.0).array, mask=spec)                 # 81 taps (the split form takes up to 65)
    with pytest.raises(_lib.HipUnsupported):
        ops.spatial_conv_mfma(cube, Gaussian2DKernel(2.0).array, mask=ops.MaskSpec(_lib.MASK_ARRAY | _lib.MASK_GT, 0.5, 0.0, mk))
    neg = np.outer([-0.1, 1.0, -0.1], [0.2, 1.0, 0.2])
    with pytest.raises(_lib.HipUnsupported):
        ops.spatial_conv_mfma(cube, neg, mask=spec)


def test_cube_level_masked_spatial_smooth_moment0_runs_fused(gpu, monkeypatch):
    """SpectralCube.spatial_smooth(...).moment0() with a boolean mask array takes the fused kernel (the smoothed cube is
    never formed) and equals the materialised path (SPC_SPATIAL_MFMA_FORM / a 41-tap kernel force it) and the oracle"""
    from spectral_cube_amd import SpectralCube
    shape = (20, 64, 160)
    d, m = _case(shape, 17, valid=0.6)
    m[:, 10:13, 20:24] = False
    hdr = {"CTYPE1": "RA---TAN", "CTYPE2": "DEC--TAN", "CTYPE3": "VRAD", "CDELT1": -1e-3, "CDELT2": 1e-3, "CDELT3": 0.5, "CUNIT3": "km/s",
           "CRPIX1": 1, "CRPIX2": 1, "CRPIX3": 1, "CRVAL1": 10.0, "CRVAL2": 20.0, "CRVAL3": -16.0, "BUNIT": "K"}
    cube = SpectralCube.read(d, hdr).with_mask(m)
    k = Gaussian2DKernel(8 / 2.3548200450309493)
    calls = []
    real = ops.spatial_conv_mfma
    monkeypatch.setattr(ops, "spatial_conv_mfma", lambda *a, **kw: (calls.append(1), real(*a, **kw))[1])
    got = np.asarray(cube.spatial_smooth(k).moment0())
    assert calls, "the fused kernel was not used"
    sm = O.spatial_smooth(d, m, k.array)
    filled = np.where(m, sm, np.nan)
    exp = 0.5 * np.nansum(filled, axis=0)
    exp[np.all(np.isnan(filled), axis=0)] = np.nan
    assert np.isnan(exp).sum() == 12
    assert_close(got, exp, atol=1e-5 * np.nanmax(np.abs(exp)), what="cube-level fused smooth -> moment0")
    # the materialised path (what runs for kernels the fused form refuses) gives the same map
    sm_dev = ops.spatial_conv(cube._device_data(), k.array, mask=cube._mask_spec())
    mat = np.asarray(SpectralCube.from_device(sm_dev, header=hdr).with_mask(m).moment0())
    assert_close(got, mat, atol=1e-5 * np.nanmax(np.abs(exp)), what="fused vs materialised")
