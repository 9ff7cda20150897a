"""Host-side mirror of the reference's operator interface for the hot path.

``SpectralCube`` keeps the reference's method names, argument meaning and
error behaviour for

    moment / moment0 / moment1 / moment2 / linewidth_sigma / linewidth_fwhm
    argmax / argmin / max / min (axis=0)
    spectral_smooth / spatial_smooth / spectral_interpolate / reproject
    with_mask / with_fill_value / filled_data / spectral_axis

(reference: spectral_cube/spectral_cube.py:1614-1763, 793-819, 2649-2842,
3186-3332; spectral_cube/dask_spectral_cube.py:880-993, 1031-1132, 1250-1373)
while every voxel is touched only by the HIP kernels behind the C ABI
(include/spcube_hip.h).  There is no CPU fallback: without the library or a
GPU the compute methods raise ``HipLibraryError``.

Differences by design: results of cube->cube operations are
float32 device-resident cubes (the Dask class keeps float32 too; only
spectral_interpolate returns float64 there); 2-D maps are float64 like the
reference; units are plain strings unless astropy is importable.  The 1e8-voxel
``warn_slow`` guard (utils.py:41-75) is kept where BOTH reference classes have it
(``reproject``, ``convolve_to``): set ``cube.allow_huge_operations = True`` for
cubes above the threshold, exactly as with the reference.
"""
import functools
import math
import operator
import os
import warnings

import numpy as np

from . import _lib, masks as M, ops
from .beam import Beam, BeamError
from .device import DeviceArray
from .kernels import kernel_array
from .wcs import (SimpleWCS, check_same_spectral_kind, join_celestial_spectral, pix_cen_spatial, pix_size, reproject_pixel_map,
                  spectral_unit_scale)

SIGMA2FWHM = 2. * np.sqrt(2. * np.log(2.))      # spectral_cube.py:66


class VarianceWarning(UserWarning):
    pass


class PrecisionWarning(UserWarning):
    """the source holds samples float32 cannot represent exactly (a float64 array, a BITPIX = -64 / 32 / 64 FITS image): the
    reference keeps such a cube in float64 (``np.result_type(dtype, 0.0)``, masks.py:225).  The moments along the spectral
    axis (moment 0 / 1 / 2 / N, argmax / argmin, moments012) of a resident cube are computed from the float64 samples
    (spc_moments_f64); every other operator stages the cube as float32 (sums are carried in float64, the SAMPLES are
    rounded to 24 bits: ~6e-8 relative per sample, inside the 1e-5 contract of the float32 configurations, but not the
    reference's float64 result).  Raised once per source, the first time its samples are narrowed."""


_NARROWED = "%s samples are narrowed to float32 on their way to HBM (the reference would keep float64, masks.py:225): results " \
            "agree with a float64 computation to ~1e-7 relative, not to float64 precision"


def _is_wide_dtype(dtype):
    dt = np.dtype(dtype)
    return (dt.kind == "f" and dt.itemsize > 4) or (dt.kind in "iu" and dt.itemsize >= 4)


def _warn_if_narrowed(dtype=None, bitpix=None, stacklevel=3):
    if dtype is not None:
        wide, what = _is_wide_dtype(dtype), str(np.dtype(dtype))
    else:
        wide = bitpix in (-64, 32, 64)
        what = "BITPIX = %s" % bitpix
    if wide:
        warnings.warn(_NARROWED % what, PrecisionWarning, stacklevel=stacklevel)


class _DataToken:
    """identity of a cube's voxel values, shared by the cubes derived from it without touching them (with_mask,
    with_fill_value, with_spectral_unit): lazy masks compare it (``_is_same_data``), and a wide source keeps what its
    float64 path needs here - the FITS image the samples come from, the float64 device copy, whether the narrowing to
    float32 has been announced."""
    __slots__ = ("wide_file", "dev64", "warned", "lazy64", "derived64")

    def __init__(self):
        self.wide_file = None        # (path, hdu, bitpix) of a resident BITPIX = -64 / 32 / 64 image
        self.dev64 = None            # float64 DeviceArray, staged by the first spectral moment
        self.warned = False
        self.lazy64 = None           # pending float64 operator of a cube DERIVED from a wide one (spectral_smooth, ...)
        self.derived64 = False       # the values exist as a float64 DeviceArray only (dev64 / lazy64): no host array, no file


def _token_dev64(tok):
    """the float64 DeviceArray of a DERIVED wide cube's data token (pending operator run once)"""
    if tok.dev64 is None:
        _lib.require_gpu()
        tok.dev64 = tok.lazy64()
        tok.lazy64 = None
    return tok.dev64


class _WideView:
    """what a mask is lowered against for the float64 kernels: the cube's identity (lazy masks bound to the cube stay
    device terms), ``_wide`` (comparison thresholds keep their float64 value: numpy compares a float64 cube in float64),
    and the float64 samples for the terms that are evaluated on the host"""
    _wide = True

    def __init__(self, cube):
        self._cube, self._data_id = cube, cube._data_id

    def _host_data(self):
        c = self._cube
        if c._data_id.wide_file is None and c._data is not None:
            return c._data
        return c._device_data64().get()


_ALL_NAN = ("All values in reprojected cube are nan.  This can be caused"
            " by an error in which coordinates do not 'round-trip'.  Try "
            "setting ``roundtrip_coords=False``.  You might also check "
            "whether the WCS transformation produces valid pixel->world "
            "and world->pixel coordinates in each axis.")


class SmoothingWarning(UserWarning):
    pass


class UnitsError(ValueError):
    """stands in for astropy.units.UnitsError"""


class WCSCelestialError(ValueError):
    """spectral_cube.utils.WCSCelestialError (utils.py:95): an image without two celestial axes was asked for a
    celestial operation (lower_dimensional_structures.py:450-538)."""


class BeamUnitsError(Exception):
    """spectral_cube.utils.BeamUnitsError (base_class.py:134-137)"""


_VARIANCE_MSG = ("Note that the second moment returned will be a "
                 "variance map. To get a linewidth map, use the "
                 "SpectralCube.linewidth_fwhm() or "
                 "SpectralCube.linewidth_sigma() methods instead.")


MEMORY_THRESHOLD = 1e8        # voxels: cube_utils.py:266-274


def warn_slow(function):
    """utils.py:41-75: operations the reference refuses on huge cubes unless
    ``allow_huge_operations`` is set (they load the whole cube there; here they need the whole cube
    resident in HBM, plus an output of the same size).  Same switch, same ValueError."""
    @functools.wraps(function)
    def wrapper(self, *args, **kwargs):
        if self._is_huge and not self.allow_huge_operations:
            raise ValueError("This function ({0}) requires loading the entire "
                             "cube into memory, and the cube is large ({1} "
                             "pixels), so by default we disable this operation. "
                             "To enable the operation, set "
                             "`cube.allow_huge_operations=True` and try again.  "
                             "See {2} for details.".format(str(function), self.size,
                                                           "https://spectral-cube.readthedocs.io/en/latest/big_data.html"))
        return function(self, *args, **kwargs)
    return wrapper


def _check_convolve(convolve):
    """The ``convolve=`` seam of spectral_smooth / spatial_smooth / convolve_to
    (dask_spectral_cube.py:883, 963; spectral_cube.py:2810, 3188, 3335).  The device stencils ARE
    astropy's normalised, NaN-interpolating convolution, so astropy's own ``convolve`` (the default of
    the smoothing methods) and ``convolve_fft`` (the default of convolve_to; same mathematics through
    an FFT) are accepted as naming it; any other callable cannot run on the device."""
    if convolve is None:
        return
    mod, name = getattr(convolve, "__module__", "") or "", getattr(convolve, "__name__", "")
    if mod.startswith("astropy.convolution") and name in ("convolve", "convolve_fft"):
        return
    raise NotImplementedError("custom `convolve` callables cannot run on the device; the HIP stencils "
                              "implement astropy.convolution.convolve (pass that, convolve_fft, or nothing)")


def _unit_mul(a, b):
    a, b = (a or "").strip(), (b or "").strip()
    return (a + " " + b).strip()


def _unit_pow(a, n):
    a = (a or "").strip()
    if not a or n == 1:
        return a
    return "(%s)%d" % (a, n) if " " in a or "/" in a else "%s%d" % (a, n)


class Projection(np.ndarray):
    """2-D result map (stands in for lower_dimensional_structures.Projection
    :246-292): an ndarray carrying unit, wcs and meta."""

    def __new__(cls, value, unit="", wcs=None, meta=None, beam=None, device=0):
        obj = np.asarray(value).view(cls)
        obj.unit = unit
        obj.wcs = wcs
        obj.meta = dict(meta or {})
        obj.beam = beam if beam is not None else obj.meta.get("beam")
        obj._spc_device = device
        return obj

    def __array_finalize__(self, obj):
        if obj is None:
            return
        self.unit = getattr(obj, "unit", "")
        self.wcs = getattr(obj, "wcs", None)
        self.meta = getattr(obj, "meta", {})
        self.beam = getattr(obj, "beam", None)
        self._spc_device = getattr(obj, "_spc_device", 0)

    def _celestial(self):
        if self.ndim != 2 or self.wcs is None or getattr(self.wcs, "naxis", 0) < 2:
            raise WCSCelestialError("WCS does not contain two spatial axes.")      # _raise_wcs_no_celestial
        return self.wcs

    def convolve_to(self, beam, convolve=None, **kwargs):
        """Convolve the image to *beam* (lower_dimensional_structures.py:450-494): the kernel
        ``beam.deconvolve(self.beam).as_kernel(pixscale)`` through the same NaN-aware 2-D stencils as
        the cube (one channel), astropy's normalised 'interpolate' treatment."""
        treat = kwargs.pop("nan_treatment", "interpolate")
        _check_convolve(convolve)
        if treat not in ("interpolate", "fill") or kwargs.pop("fill_value", 0.0) != 0.0 or kwargs:
            raise NotImplementedError("the device stencil is astropy.convolution.convolve with boundary='fill', "
                                      "fill_value=0, normalize_kernel=True and nan_treatment 'interpolate' or 'fill'")
        w = self._celestial()
        if self.beam is None:
            raise ValueError("No beam is contained in Projection.meta.")
        if beam == self.beam:
            warnings.warn("The given beam is identical to the current beam. Skipping convolution.")
            return self
        psm = w.pixel_scale_matrix
        pixscale = math.sqrt(abs(psm[0, 0] * psm[1, 1] - psm[0, 1] * psm[1, 0]))
        karr = beam.deconvolve(self.beam).as_kernel(pixscale)
        _lib.require_gpu()
        pix = np.asarray(self, dtype=np.float32)
        if treat == "fill":                  # astropy: NaN -> fill_value (0), plain normalised convolution
            pix = np.where(np.isnan(pix), np.float32(0.0), pix)
        img = DeviceArray.from_numpy(pix[None], self._spc_device)
        out = ops.spatial_conv(img, karr).get()[0].astype(self.dtype if self.dtype.kind == "f" else np.float32)
        return Projection(out, unit=self.unit, wcs=self.wcs, meta=dict(self.meta, beam=beam), beam=beam, device=self._spc_device)

    def reproject(self, header, order="bilinear"):
        """Reproject the image onto the celestial WCS of *header* (lower_dimensional_structures.py:496-538):
        bilinear, NaN outside the footprint - the cube's resampling kernel on one channel."""
        if order not in ("bilinear", 1):
            raise NotImplementedError("only order='bilinear' is built on the GPU path")
        w = self._celestial()
        newwcs = header if isinstance(header, SimpleWCS) else SimpleWCS(header, naxis=2)
        hdr = newwcs.header
        ny_out, nx_out = (int(hdr["NAXIS2"]), int(hdr["NAXIS1"])) if ("NAXIS1" in hdr and "NAXIS2" in hdr) else self.shape
        _lib.require_gpu()
        xs, ys = ops.wcs_pixel_map(w, newwcs, (ny_out, nx_out), self._spc_device)
        img = DeviceArray.from_numpy(np.asarray(self, dtype=np.float32)[None], self._spc_device)
        dev, _ = ops.resample_bilinear(img, xs, ys, fill=np.nan, want_footprint=False)
        out = dev.get()[0].astype(self.dtype if self.dtype.kind == "f" else np.float32)
        return Projection(out, unit=self.unit, wcs=newwcs, meta=dict(self.meta), beam=self.beam, device=self._spc_device)

    @property
    def value(self):
        return np.asarray(self)

    def quantity(self):
        """astropy Quantity when astropy is importable."""
        from astropy import units as u
        return u.Quantity(np.asarray(self), u.Unit(self.unit))


class SpectralCube:
    """(nz, ny, nx) float32 cube with a lazily lowered mask, device resident."""

    def __init__(self, data=None, wcs=None, mask=None, meta=None, fill_value=np.nan,
                 header=None, unit=None, device=0, allow_huge_operations=False, _dev=None,
                 _lazy=None, _shape=None, _data_id=None, _source=None):
        if data is None and _dev is None and _lazy is None and _source is None:
            raise ValueError("data is required")
        if data is not None:
            data = np.asarray(data)
            if data.ndim != 3:
                raise ValueError("SpectralCube needs a 3-D (spectral, y, x) array")
        self._data = data                 # host ndarray or None
        self._dev = _dev                  # DeviceArray float32 or None
        self._lazy = _lazy                # pending (op, parent, args) - see spectral_smooth
        self._source = _source            # streaming.FitsSource / NdarraySource of an out-of-core cube, or None
        self._shape = tuple(data.shape) if data is not None else (
            tuple(_dev.shape) if _dev is not None else tuple(_shape))
        # (strict=False: a header with celestial keywords the minimal WCS does not model still gives a cube - moments along the
        # spectral axis need no celestial transform; the first celestial USE raises, naming the keyword: wcs.py)
        if wcs is None and header is not None:
            wcs = SimpleWCS(header, strict=False)
        elif wcs is not None and not isinstance(wcs, SimpleWCS):
            wcs = SimpleWCS(wcs, strict=False)
        self._wcs = wcs
        self._header = dict(wcs.header) if wcs is not None else {}
        self._mask = mask
        self._meta = dict(meta or {})
        self._fill_value = fill_value
        if unit is None:
            unit = str(self._header.get("BUNIT", "")).strip()
        self._unit = unit
        self.device = device
        self.allow_huge_operations = allow_huge_operations
        self._mask_cache = None
        self._cen_cache = {}
        # cubes that share the same voxel values (with_mask, with_fill_value) share
        # this token; lazy masks use it to decide whether their predicate may be
        # evaluated by the kernel on the data it is reading
        self._data_id = _data_id if _data_id is not None else _DataToken()
        self._mask64_cache = None

    # ---- construction helpers ------------------------------------------------
    @classmethod
    def read(cls, data, header=None, device=0, hdu=None, **kw):
        """``SpectralCube.read``: a FITS file name (streamed to HBM through pinned staging
        buffers and decoded on the device, ``io_fits.load_cube``), or an in-memory array +
        header.  Like the FITS reader (spectral_cube/io/fits.py:171-260) it attaches
        ``LazyMask(np.isfinite)`` and copies ``BUNIT`` into ``meta``."""
        if isinstance(data, (str, os.PathLike)):
            from . import io_fits, streaming
            img = io_fits.find_image(os.fspath(data), hdu)
            fshape = tuple(io_fits.cube_shape(img))
            _lib.require_gpu()
            if 4 * int(np.prod(fshape, dtype=np.int64)) > streaming.hbm_budget(device):
                _warn_if_narrowed(bitpix=img.bitpix)         # the strips are staged as float32
                # larger than the HBM budget: the cube stays in the file and goes through the device in row
                # strips (streaming.py; the role of _moments.py:89-125 / cube_utils.py:277-301 in the reference)
                src = streaming.FitsSource(os.fspath(data), hdu)
                hdr = io_fits.cube_header(img)
                meta = dict(kw.pop("meta", None) or {})
                if "BUNIT" in hdr:
                    meta["BUNIT"] = hdr["BUNIT"]
                table = cls._beams_table_for(os.fspath(data), fshape[0]) if "beams" not in kw else None
                if table is not None:            # a BEAMS extension makes it a varying-resolution cube (io/fits.py:216-228)
                    cls, kw = VaryingResolutionSpectralCube, dict(kw, beam_table=table)
                cube = cls(None, header=hdr, device=device, meta=meta, _source=src, _shape=fshape, **kw)
                finite = M.LazyMask(np.isfinite, cube=cube)
                beam_mask = cube._mask if isinstance(cube, VaryingResolutionSpectralCube) else None
                cube._mask = finite if beam_mask is None else (finite & beam_mask)
                return cube
            wide = img.bitpix in (-64, 32, 64)
            if wide:
                # samples float32 cannot hold: nothing is staged yet - the spectral moments read the image as float64
                # (_device_data64), any other operator as float32 (with a PrecisionWarning), whichever comes first
                dev, hdr = None, io_fits.cube_header(img)
            else:
                dev, hdr = io_fits.load_cube(os.fspath(data), device=device, hdu=hdu)
            meta = dict(kw.pop("meta", None) or {})
            if "BUNIT" in hdr:
                meta["BUNIT"] = hdr["BUNIT"]
            table = cls._beams_table_for(os.fspath(data), fshape[0]) if "beams" not in kw else None
            if table is not None:            # a BEAMS extension makes it a varying-resolution cube (io/fits.py:216-228)
                cls, kw = VaryingResolutionSpectralCube, dict(kw, beam_table=table)
            if wide:
                path_, hdu_ = os.fspath(data), hdu
                cube = cls(None, header=hdr, device=device, meta=meta, _shape=fshape,
                           _lazy=lambda: io_fits.load_cube(path_, device=device, hdu=hdu_)[0], **kw)
                cube._data_id.wide_file = (path_, hdu_, img.bitpix)
            else:
                cube = cls(None, header=hdr, device=device, _dev=dev, meta=meta, **kw)
        else:
            cube = cls(np.asarray(data), header=header, device=device, **kw)
        finite = M.LazyMask(np.isfinite, cube=cube)
        beam_mask = cube._mask if isinstance(cube, VaryingResolutionSpectralCube) else None
        cube._mask = finite if beam_mask is None else (finite & beam_mask)
        return cube

    @staticmethod
    def _beams_table_for(path, nchan):
        """the file's BEAMS table when it has one row per channel (io/fits.py:216-228), else None"""
        from . import io_fits
        table = io_fits.read_beams_table(path)
        if table is not None and len(table["BMAJ"]) != nchan and "POL" in table:
            # full-polarisation tables list every channel once per Stokes plane (cube_utils._split_stokes
            # hands each component its rows); this path reads one plane: keep the first polarisation's rows
            keep = table["POL"] == table["POL"][0]
            table = {k: v[keep] for k, v in table.items()}
        if table is not None and len(table["BMAJ"]) != nchan:
            warnings.warn("BEAMS table with %d rows does not match the %d channels of the cube: ignored"
                          % (len(table["BMAJ"]), nchan), BeamWarning)
            table = None
        return table

    def write(self, filename, overwrite=False, format=None, filled=True):
        """Write the cube as FITS (io/fits.py:262-294; device byte swap, pinned read-back, no host
        arithmetic).  Like the reference's ``hdu`` (dask_spectral_cube.py:1405,1502:
        ``_get_filled_data(fill=self._fill_value)``) what goes to disk is the FILLED data: excluded
        voxels carry the fill value (NaN by default).  ``filled=False`` writes the raw voxels."""
        from . import io_fits
        if self._runs_wide():
            # a float64 cube goes to disk as BITPIX = -64 (the reference writes the array it holds, io/fits.py:262-294)
            path = os.fspath(filename)
            if os.path.exists(path) and not overwrite:
                raise OSError("File %r already exists (use overwrite=True)" % path)
            # the float64 samples explicitly: _host_data() returns the float32-narrowed copy once a float32-only operator has
            # staged one, and the written precision must not depend on the call history
            if self._data is not None and getattr(self._data, "dtype", None) == np.float64:
                data = self._data
            else:
                data = self._device_data64().get()
            if filled and self._mask is not None:
                data = self._mask._filled(data, fill=self._fill_value)
            hdr = {k: v for k, v in self._header.items() if k != "WCSAXES"}
            io_fits.write_fits(path, np.asarray(data, dtype=np.float64), header=hdr)
            return
        plan = self._strip_plan(filled)
        if plan is not None:             # out of core: strips in, operator, strips out (streaming.map_strips / map_slabs)
            from . import streaming
            sink = streaming.FitsSink(os.fspath(filename), self._header, tuple(self._shape), overwrite=overwrite)
            self._run_plan(plan, sink)
            return
        dev = self._device_data()
        if filled and self._mask is not None:
            dev = ops.fill_masked(dev, self._mask_spec(), self._fill_value)
        io_fits.save_cube(os.fspath(filename), dev, header=self._header, overwrite=overwrite)

    def _strip_plan(self, filled=True):
        """(streamed source cube, fn(strip, mask spec, stream) -> result strip, result channels) when this cube is - or
        is a pending per-spaxel operator (spectral_smooth, spectral_interpolate, sigma_clip_spectrally) on - a cube that
        does not fit the HBM budget; None otherwise."""
        fill = self._fill_value
        if self._stream_source() is not None:
            if filled and self._mask is not None:
                return self, (lambda dev, mspec, stream: ops.fill_masked(dev, mspec, fill, stream)), self._shape[0]
            # (the strip buffer itself is recycled by the pipeline: hand out a copy)
            return self, (lambda dev, mspec, stream: ops.fill_masked(dev, None, fill, stream)), self._shape[0]
        lz = self._lazy
        parent = getattr(lz, "parent", None)
        if (self._dev is None and lz is not None and parent is not None and getattr(lz, "slab_fn", None) is not None
                and parent._stream_source() is not None):
            if filled and self._mask is not None and getattr(lz, "keeps_mask", False) and self._mask is parent._mask:
                def slab_filled(dev, mspec, stream):     # (convolve_to: the parent's mask, on the parent's voxels)
                    from . import streaming
                    keep = streaming.original_include(parent, dev, mspec, stream)
                    return ops.fill_masked(lz.slab_fn(dev, mspec, stream), keep, fill, stream)
                return parent, slab_filled, self._shape[0]
            return parent, lz.slab_fn, self._shape[0]        # (reprojection: NaN outside the footprint is its own fill)
        if self._dev is not None or lz is None or parent is None or getattr(lz, "strip_fn", None) is None:
            return None
        if parent._stream_source() is None:
            return None
        keeps_parent_mask = self._mask is parent._mask
        if filled and self._mask is not None and keeps_parent_mask:
            def fn(dev, mspec, stream):      # the parent's mask, evaluated on the parent's voxels (as _mask_spec does)
                from . import streaming
                keep = streaming.original_include(parent, dev, mspec, stream)
                return ops.fill_masked(lz.strip_fn(dev, mspec, stream), keep, fill, stream)
        else:
            fn = lz.strip_fn             # (spectral_interpolate: the new mask is ~isnan(result), the data are their own fill)
        return parent, fn, self._shape[0]

    def stream_into(self, out):
        """Out-of-core counterpart of ``filled_data``: fill the float32 (nz, ny, nx) host array / memory map *out* with
        the filled data of this cube, strip by strip (the cube, or the pending per-spaxel operator on a cube, larger than
        the HBM budget).  Returns *out*."""
        from . import streaming
        plan = self._strip_plan(True)
        if plan is None:
            raise ValueError("this cube fits the device: use filled_data")
        self._run_plan(plan, streaming.NdarraySink(out))
        return out

    def _run_plan(self, plan, sink):
        from . import streaming
        src, fn, nz_out = plan
        if getattr(self._lazy, "slab_fn", None) is not None and self._dev is None:
            streaming.map_slabs(src, fn, tuple(self._shape[1:]), sink)        # per-channel operator: slabs of whole planes
        else:
            streaming.map_strips(src, fn, nz_out, sink, halo=self._strip_halo())

    def _strip_halo(self):
        return int(getattr(self._lazy, "halo", 0)) if (self._lazy is not None and self._dev is None) else 0

    @classmethod
    def from_device(cls, dev, wcs=None, header=None, mask=None, **kw):
        if dev.dtype != np.float32 or len(dev.shape) != 3:
            raise TypeError("device cube must be float32 (nz, ny, nx)")
        return cls(None, wcs=wcs, header=header, mask=mask, device=dev.device, _dev=dev, **kw)

    def _new_cube_with(self, data=None, dev=None, wcs=None, mask=None, meta=None, fill_value=None,
                       lazy=None, shape=None, unit=None, same_data=False):
        return SpectralCube(data, wcs=self._wcs if wcs is None else wcs,
                            mask=self._mask if mask is None else mask,
                            meta=self._meta if meta is None else meta,
                            fill_value=self._fill_value if fill_value is None else fill_value,
                            unit=self._unit if unit is None else unit, device=self.device,
                            allow_huge_operations=self.allow_huge_operations, _dev=dev,
                            _lazy=lazy, _shape=shape, _data_id=self._data_id if same_data else None,
                            _source=self._source if same_data else None)

    # ---- identity used by lazy masks -------------------------------------------
    def _host_data(self):
        if self._data is None:
            if self._data_id.derived64 or (self._data_id.wide_file is not None and self._dev is None and self._wide_resident()):
                # the result of a float64 operator / a resident BITPIX = -64 / 32 / 64 image: float64 on the host as well
                # (what the reference holds, masks.py:225 - host-evaluated mask terms compare these samples)
                self._data = self._device_data64().get()
            else:
                self._data = self._device_data().get()
        return self._data

    def _stream_source(self):
        """streaming source of a cube that is NOT resident and larger than the HBM budget (a FITS file, a memory
        map, a host array), else None.  Such a cube runs its spectral-axis reductions strip by strip
        (streaming.py) and refuses the operators that need it whole."""
        if self._dev is not None or self._lazy is not None:
            return None
        if self._source is not None:
            return self._source
        if self._data is not None:
            from . import streaming
            if 4 * self._data.size > streaming.hbm_budget(self.device):
                self._source = streaming.NdarraySource(self._data)
        return self._source

    def _is_same_data(self, data):
        return getattr(data, "_data_id", None) is self._data_id

    # ---- basic properties --------------------------------------------------------
    @property
    def shape(self):
        return self._shape

    @property
    def ndim(self):
        return 3

    @property
    def size(self):
        return int(np.prod(self._shape, dtype=np.int64))

    @property
    def _is_huge(self):
        return self.size > MEMORY_THRESHOLD                  # cube_utils.is_huge

    @property
    def unit(self):
        return self._unit

    @property
    def wcs(self):
        return self._wcs

    @property
    def header(self):
        return self._header

    @property
    def meta(self):
        return self._meta

    @property
    def mask(self):
        return self._mask

    @property
    def fill_value(self):
        return self._fill_value

    @property
    def spectral_axis(self):
        """world coordinate of every channel, in CUNIT3 (base_class.py:276-281)."""
        return self._wcs.spectral_pix2world(np.arange(self._shape[0]))

    @property
    def spectral_unit(self):
        return self._wcs.spectral_unit if self._wcs is not None else ""

    def with_spectral_unit(self, unit, velocity_convention=None, rest_value=None):
        """The same cube with its spectral axis in another unit OF THE SAME KIND (m/s <-> km/s, Hz <-> GHz, ...:
        spectral_cube.py:1345-1388 for that case; base_class.py `with_spectral_unit`): the WCS is rescaled, the voxels and the mask
        are shared.  Moments then come out in the new unit (tests/test_moments.py:145-155).  Changing the KIND of axis
        (frequency <-> velocity, another velocity convention or rest value) is the reference's spectral_axis machinery, outside
        the dense path: NotImplementedError."""
        unit = str(getattr(unit, "to_string", lambda: unit)()).replace(" ", "")
        if velocity_convention is not None or rest_value is not None:
            raise NotImplementedError("changing the velocity convention / rest value is not built: only same-kind unit changes are")
        try:
            scale = spectral_unit_scale(self.spectral_unit, unit)
        except ValueError as exc:
            raise NotImplementedError(str(exc))
        w = self._wcs
        newwcs = w.with_spectral(w.crval[2] * scale, w.cdelt[2] * w.pc[2, 2] * scale, w.crpix[2], cunit=unit)
        out = self._new_cube_with(data=self._data, dev=self._dev, wcs=newwcs, lazy=self._lazy, shape=self._shape, same_data=True)
        out._mask_cache = self._mask_cache
        return out

    def with_mask(self, mask, inherit_mask=True):
        """spectral_cube.py:1390-1441: AND the new mask with the existing one."""
        if isinstance(mask, np.ndarray):
            np.broadcast_shapes(mask.shape, self._shape)
            mask = M.BooleanArrayMask(mask, self._wcs, shape=self._shape)
        if self._mask is not None and inherit_mask:
            mask = self._mask & mask
        out = self._new_cube_with(data=self._data, dev=self._dev, mask=mask, lazy=self._lazy,
                                  shape=self._shape, same_data=True)
        return out

    def with_fill_value(self, fill_value):
        out = self._new_cube_with(data=self._data, dev=self._dev, fill_value=fill_value,
                                  lazy=self._lazy, shape=self._shape, same_data=True)
        out._mask_cache = self._mask_cache
        return out

    def _cmp(self, op, value):
        value = getattr(value, "value", value)
        return M.LazyComparisonMask(op, value, cube=self)

    def __gt__(self, value):
        return self._cmp(operator.gt, value)

    def __ge__(self, value):
        return self._cmp(operator.ge, value)

    def __lt__(self, value):
        return self._cmp(operator.lt, value)

    def __le__(self, value):
        return self._cmp(operator.le, value)

    # ---- data access -----------------------------------------------------------------
    def _device_data(self):
        """float32 DeviceArray of the cube values (uploads / materialises lazily)."""
        if self._dev is None:
            self._note_narrowed()
            if self._lazy is not None:
                self._dev = self._lazy()
                self._lazy = None
            else:
                _lib.require_gpu()
                if self._stream_source() is not None:
                    from . import streaming
                    nbytes = 4 * int(np.prod(self._shape, dtype=np.int64))
                    raise streaming.HugeCubeError(
                        "this operation needs the whole cube in HBM: %.2f GiB against a budget of %.2f GiB (SPC_HBM_BUDGET). "
                        "Out-of-core cubes stream moments, argmax / argmin, the nan-reductions and median / percentile / mad_std along any "
                        "axis, statistics(), and spectral_smooth / spatial_smooth / spectral_interpolate / sigma_clip_spectrally / reproject "
                        "(celestial) into write(), stream_into() or a following moment"
                        % (nbytes / 2**30, streaming.hbm_budget(self.device) / 2**30))
                self._dev = DeviceArray.from_numpy(self._data, self.device, dtype=np.float32)
        return self._dev

    # ---- wide sources (float64 arrays, BITPIX = -64 / 32 / 64 images): the spectral moments in their own precision ------
    def _is_wide(self):
        if self._data_id.wide_file is not None or self._data_id.derived64:
            return True
        return self._data is not None and _is_wide_dtype(self._data.dtype)

    def _note_narrowed(self, stacklevel=4):
        """PrecisionWarning, once per source, when a wide source is about to be used as float32"""
        tok = self._data_id
        if tok.warned or not self._is_wide():
            return
        tok.warned = True
        if tok.wide_file is not None:
            _warn_if_narrowed(bitpix=tok.wide_file[2], stacklevel=stacklevel)
        elif tok.derived64:
            _warn_if_narrowed(dtype=np.float64, stacklevel=stacklevel)
        else:
            _warn_if_narrowed(dtype=self._data.dtype, stacklevel=stacklevel)

    def _wide_resident(self):
        """True when the spectral moments of this cube run on its float64 samples: a wide source whose values this cube
        still is (no pending operator) and whose float64 copy fits the HBM budget"""
        if self._data_id.derived64:
            return True
        if not self._is_wide() or (self._lazy is not None and self._data_id.wide_file is None):
            return False
        if self._data_id.dev64 is not None:
            return True
        from . import streaming
        return 8 * int(np.prod(self._shape, dtype=np.int64)) <= streaming.hbm_budget(self.device)

    def _device_data64(self):
        """float64 DeviceArray of a wide source (uploaded / decoded from the FITS image once, shared by the cubes that share
        the data)"""
        tok = self._data_id
        if tok.dev64 is None:
            _lib.require_gpu()
            if tok.lazy64 is not None:
                tok.dev64 = tok.lazy64()
                tok.lazy64 = None
            elif tok.wide_file is not None:
                from . import io_fits
                path, hdu, _ = tok.wide_file
                tok.dev64 = io_fits.load_cube(path, device=self.device, hdu=hdu, dtype=np.float64)[0]
            else:
                tok.dev64 = DeviceArray.from_numpy(self._data, self.device, dtype=np.float64)
        return tok.dev64

    def _mask_spec64(self):
        """the mask lowered for the float64 kernels: thresholds in float64, host-evaluated terms on the float64 samples"""
        if self._mask64_cache is None:
            owner = M.foreign_owner(self._mask) if self._mask is not None else None
            if (owner is not None and not owner._is_same_data(self) and tuple(owner._shape) == tuple(self._shape)
                    and getattr(owner, "_wide_resident", lambda: False)() and self._mask._device_terms(_WideView(owner)) is not None):
                # every lazy term belongs to ANOTHER float64 cube's data (a smoothed cube keeps its parent's mask):
                # evaluated there, on the device, on the float64 samples (as _mask_spec does for float32 cubes)
                flags, lo, hi, arr = M.lower_mask(self._mask, _WideView(owner), self._shape)
                darr = DeviceArray.from_numpy(arr, self.device) if arr is not None else None
                inc = ops.mask_include_f64(owner._device_data64(), ops.MaskSpec(flags, lo, hi, darr),
                                           nan_excluded=M.contains(self._mask, M.NotNaNMask))
                self._mask64_cache = ops.MaskSpec(_lib.MASK_ARRAY, 0.0, 0.0, inc)
            else:
                flags, lo, hi, arr = M.lower_mask(self._mask, _WideView(self), self._shape)
                darr = DeviceArray.from_numpy(arr, self.device) if arr is not None else None
                self._mask64_cache = ops.MaskSpec(flags, lo, hi, darr)
        return self._mask64_cache

    def _runs_wide(self):
        """True when a cube -> cube operator of this cube runs on float64 samples: a wide cube that is resident (or fits).
        Decided without a device where it can be (a float32 cube never asks); no device at all = the float32 plan, which
        raises when it is run"""
        if not self._is_wide():
            return False
        try:
            return self._stream_source() is None and self._wide_resident()
        except _lib.HipLibraryError:
            return False

    def _new_wide_cube(self, fn64, shape=None, wcs=None, mask=None, plain=False):
        """the result of a float64 operator on this (wide, resident) cube: a cube whose values exist as a float64 DeviceArray
        (pending until first used; the Dask class keeps the chunk dtype, dask_spectral_cube.py:829).  Its spectral moments,
        reductions, statistics() and further smoothing / interpolation run in float64; an operator without a float64 form
        narrows it with a PrecisionWarning, like a float64 source."""
        # the narrowing thunk holds the result's DATA TOKEN, never the cube: out._lazy -> thunk -> out was a reference cycle
        # that kept a dropped float64 cube (8 B / voxel of HBM) and its parent alive until a generation-2 collection
        tokens = []
        narrow = _Thunk(lambda: ops.narrow_f64(_token_dev64(tokens[0])))
        make = (lambda **kw: SpectralCube._new_cube_with(self, **kw)) if plain else self._new_cube_with      # plain: not the subclass's (beams)
        out = make(lazy=narrow, shape=tuple(shape) if shape is not None else self._shape, wcs=wcs, mask=mask)
        out._data_id.lazy64 = fn64
        out._data_id.derived64 = True
        tokens.append(out._data_id)
        return out

    def _mask_spec(self):
        """lower the mask tree once and keep the uint8 array resident in HBM."""
        if self._mask_cache is None:
            owner = M.foreign_owner(self._mask) if self._mask is not None else None
            if (owner is not None and not owner._is_same_data(self) and tuple(owner._shape) == tuple(self._shape)
                    and self._mask._device_terms(owner) is not None):
                # every lazy term belongs to ANOTHER cube's data (a smoothed cube keeps its parent's mask):
                # evaluate it there, on the device - no host copy of either cube
                flags, lo, hi, arr = M.lower_mask(self._mask, owner, self._shape)
                darr = DeviceArray.from_numpy(arr, self.device) if arr is not None else None
                inc = ops.mask_include(owner._device_data(), ops.MaskSpec(flags, lo, hi, darr),
                                       nan_excluded=M.contains(self._mask, M.NotNaNMask))
                self._mask_cache = ops.MaskSpec(_lib.MASK_ARRAY, 0.0, 0.0, inc)
            else:
                flags, lo, hi, arr = M.lower_mask(self._mask, self, self._shape)
                darr = DeviceArray.from_numpy(arr, self.device) if arr is not None else None
                self._mask_cache = ops.MaskSpec(flags, lo, hi, darr)
        return self._mask_cache

    @property
    def filled_data(self):
        """host copy with excluded voxels replaced by fill_value
        (base_class.py:389-417 / masks.py:197-237)."""
        d = self._host_data()
        if self._mask is None:
            return d
        return self._mask._filled(d, fill=self._fill_value)

    unitless_filled_data = filled_data

    @property
    def unmasked_data(self):
        return self._host_data()

    # ---- coordinates fed to the kernels -------------------
    def _pix_size_slice(self, axis):
        return pix_size(self._wcs, axis)

    def _pix_cen_axis(self, axis):
        if axis not in self._cen_cache:
            if axis == 0:
                spec = self.spectral_axis
                self._cen_cache[0] = spec - spec[0]          # spectral_cube.py:1473-1475
            else:
                y, x = pix_cen_spatial(self._wcs, self._shape)
                self._cen_cache[1], self._cen_cache[2] = y, x
        return self._cen_cache[axis]

    # ---- moments -----------------------------------------------------------------------
    def _moment_device(self, want, fused_kernel=None):
        nz = self._shape[0]
        cen = self._pix_cen_axis(0)
        cref = cen[nz // 2]
        spec0 = self.spectral_axis[0]
        d_cen = DeviceArray.from_numpy(cen - cref, self.device)
        dv = self._pix_size_slice(0)
        if fused_kernel is None and self._wide_resident():
            # a float64 source: its own precision (the reference's maps are float64 sums of float64 samples)
            return ops.moments_f64(self._device_data64(), d_cen, dv=dv, m1_add=cref + spec0, mask=self._mask_spec64(), want=want)
        if fused_kernel is None and self._is_wide():
            self._note_narrowed()
        if fused_kernel is not None and fused_kernel[0]._stream_source() is not None:
            from . import streaming                     # out-of-core parent: the fused kernels, strip by strip
            try:
                return streaming.moments(fused_kernel[0], want, d_cen, dv, cref + spec0, kernel=fused_kernel[1], cen_host=cen - cref)
            except _lib.HipUnsupported:
                # a kernel wider than the rings on a strip that holds an invalid sample (or an extremum requested): the
                # resident path materialises the smoothed cube; here the smoothed STRIP is materialised, then reduced.
                # The maps are rebuilt from the first strip on (a strip's rows are written whole, nothing is accumulated).
                lz = self._lazy
                return streaming.moments(lz.parent, want, d_cen, dv, cref + spec0, pre=lz.strip_fn,
                                         halo=int(getattr(lz, "halo", 0)))
        if fused_kernel is None and self._stream_source() is not None:
            from . import streaming
            return streaming.moments(self, want, d_cen, dv, cref + spec0)
        lz = self._lazy
        if (fused_kernel is None and self._dev is None and lz is not None and getattr(lz, "strip_fn", None) is not None
                and getattr(lz, "parent", None) is not None and lz.parent._stream_source() is not None
                and lz.parent._mask is self._mask and tuple(lz.parent._shape) == tuple(self._shape)):
            # out-of-core parent and a pending cube -> cube operator that keeps the parent's mask (spatial_smooth outside the
            # all-valid algebraic case: strip + halo rows, the strip's own rows reduced; sigma_clip_spectrally; a
            # spectral_smooth too wide to fuse): operator and reduction strip by strip
            from . import streaming
            return streaming.moments(lz.parent, want, d_cen, dv, cref + spec0, pre=lz.strip_fn, halo=int(getattr(lz, "halo", 0)))
        if fused_kernel is not None:
            parent, karr = fused_kernel
            try:
                return ops.spectral_conv_moments(parent._device_data(), karr, d_cen, dv=dv,
                                                 m1_add=cref + spec0, mask=parent._mask_spec(), want=want,
                                                 cen_host=cen - cref)
            except _lib.HipUnsupported:
                pass        # e.g. a wide kernel on a cube with invalid samples: materialise, then reduce
        return ops.moments(self._device_data(), d_cen, dv=dv, m1_add=cref + spec0,
                           mask=self._mask_spec(), want=want)

    def _fusable(self):
        """pending spectral_smooth whose parent shares this cube's mask object:
        smooth -> moment can run fused (dask_spectral_cube.py:836-840 keeps the
        original mask, so the semantics are identical)."""
        lz = self._lazy
        if (lz is not None and getattr(lz, "op", None) == "spectral_smooth" and self._dev is None
                and lz.parent._mask is self._mask and lz.fusable):
            return (lz.parent, lz.kernel)
        return None

    def _moments_of_spatially_smoothed(self, want):
        """spatial_smooth -> moment without smoothing nz planes.  With every voxel valid astropy's
        convolution is linear (zero fill, division by the kernel sum: dask_spectral_cube.py:
        962-993), so it commutes with the sums along the spectral axis (:1083-1104):
        S_n' = conv2d(S_n).  One pass over the UNSMOOTHED cube gives S0, S1, S2 (float64), three
        small map convolutions give the moments of the smoothed cube.  Returns None when the
        shortcut does not apply (mask terms other than isfinite, any invalid voxel, S0 == 0) -
        the caller then materialises the smoothed cube."""
        lz = self._lazy
        if not (lz is not None and getattr(lz, "op", None) == "spatial_smooth" and self._dev is None
                and lz.parent._mask is self._mask):
            return None
        parent = lz.parent
        spec = parent._mask_spec()
        split_ok = getattr(lz, "arithmetic", None) != "f32"        # (asked for float32 multiply-adds: the fused forms are split-form kernels)
        if not split_ok and spec.array is not None:
            return None
        if tuple(want) == ("m0",) and parent._stream_source() is None and (spec.array is not None or (spec.flags & ~_lib.MASK_FINITE) == 0):
            fused = self._fused_masked_smooth_moment0(parent, spec, lz.kernel)
            if fused is not None:
                return {"m0": fused}
        elif (set(want) <= {"m0", "m1", "m2"} and parent._stream_source() is None and spec.array is not None
              and self._prefer_fused_higher_moments()):
            fused = self._fused_masked_smooth_moments(parent, spec, lz.kernel, want)
            if fused is not None:
                return fused
        if spec.array is not None or (spec.flags & ~_lib.MASK_FINITE):
            return None
        nz = self._shape[0]
        higher = ("m1" in want) or ("m2" in want)
        # moment 0 alone: the kernel scales by the channel width (m0 = dv * S0), the convolved map is the result
        r = parent._moment_device((("s0", "mu", "m2") if higher else ("m0",)) + ("nvalid",))
        if not higher:
            # every spaxel has nz valid voxels, and no NaN / Inf sum anywhere (cannot happen when the mask keeps finite samples
            # only): both checked on the device - four bytes come back instead of the count map and a host reduction of the
            # result (17.5 -> 14 ms at C4)
            c0d = ops.map_conv2d(r["m0"], lz.kernel)             # the sums never leave the device
            if ops.map_check(r["nvalid"], nz, None if (spec.flags & _lib.MASK_FINITE) else c0d):
                return None
            return {"m0": c0d.get()}
        if ops.map_check(r["nvalid"], nz):
            return None
        # S1 = mu * S0 and S2 = (m2 + mu^2) * S0 about the reference channel are smoothed like S0; the moments of the
        # smoothed cube are their ratios.  All of it on the device: only the requested maps come back.  A spaxel whose
        # spectrum sums to zero (0/0 in mu), a zero smoothed sum or a non-finite sample shows up as a non-finite value
        # in the result: not this path.
        s0, mu, m2 = r["s0"], r["mu"], r["m2"]
        c0d = ops.map_conv2d(s0, lz.kernel)
        c1d = ops.map_conv2d(ops.map_arith(_lib.MAP_MUL, mu, s0), lz.kernel)
        cen = self._pix_cen_axis(0)
        cref = cen[nz // 2]
        out = {}
        if "m1" in want:
            out["m1"] = ops.map_arith(_lib.MAP_DIV_ADD, c1d, c0d, s=cref + self.spectral_axis[0]).get()
        if "m2" in want:
            c2d = ops.map_conv2d(ops.map_arith(_lib.MAP_SECOND_MOMENT_SUM, m2, mu, s0), lz.kernel)
            mupd = ops.map_arith(_lib.MAP_DIV_ADD, c1d, c0d, s=0.0)
            out["m2"] = ops.map_arith(_lib.MAP_DIV_SUB_SQ, c2d, c0d, mupd).get()
        if "m0" in want:
            out["m0"] = self._pix_size_slice(0) * c0d.get()
        if not all(np.isfinite(m.sum()) for m in out.values()):
            return None
        return out

    def _prefer_fused_higher_moments(self):
        """Measured at 4096 x 2048^2 + uint8 mask (round 6): the fused moments 0 / 1 / 2 take 55 ms (eight channel-parallel
        waves per block, bands of 96 rows; round 5: 71 ms with 16-row wave regions) against 39 + 14 ms for smoothing into a
        second cube (whole-column march) and reducing that.  So the fused form remains for cubes whose smoothed copy does not
        fit beside them in HBM (SPC_FUSED_SMOOTH_MOMENTS=1 / 0 forces / forbids it)."""
        env = os.environ.get("SPC_FUSED_SMOOTH_MOMENTS")
        if env is not None:
            return env == "1"
        from .device import device_info
        nz, ny, nx = self._shape
        return device_info(self.device)["free_mem"] < 1.05 * 4 * nz * ny * nx

    def _fused_masked_smooth_moments(self, parent, spec, kernel2d, want):
        """masked spatial_smooth -> moment 1 / 2 (and 0) in ONE kernel, like _fused_masked_smooth_moment0: the split form of
        spc_spatial_conv_sep_mfma_moments_f32 keeps sum v, sum c v, sum c^2 v of the smoothed values per spaxel
        (dask_spectral_cube.py:962-993 then :1083-1104).  None when the kernel does not take the case."""
        nz = self._shape[0]
        cen = self._pix_cen_axis(0)
        cref = cen[nz // 2]
        d_cen = DeviceArray.from_numpy(cen - cref, self.device)
        try:
            _, maps = ops.spatial_conv_mfma_moments(parent._device_data(), kernel2d, d_cen, dv=self._pix_size_slice(0),
                                                    m1_add=cref + self.spectral_axis[0], mask=spec, want=tuple(want))
        except _lib.HipUnsupported:
            return None
        return {k: v.get() for k, v in maps.items()}

    def _fused_masked_smooth_moment0(self, parent, spec, kernel2d):
        """masked spatial_smooth -> moment0 in ONE kernel that never writes the smoothed cube (the lazy Dask graph of the
        reference does not materialise it either: dask_spectral_cube.py:962-993 then :1083-1104): numerator and denominator of
        the NaN-aware convolution on the matrix cores, moment sums in registers / LDS (spc_spatial_conv_sep_mfma_f32).
        None when the kernel does not take the case (more than 29 taps per axis, non-separable or negative kernels, mask
        terms other than the array / isfinite): the caller materialises.  Only tried where it pays: a mask ARRAY (an
        all-valid cube takes the algebraic path below, which is cheaper still)."""
        if spec.array is None:
            return None
        try:
            _, m0 = ops.spatial_conv_mfma(parent._device_data(), kernel2d, mask=spec, want_cube=False, want_m0=True,
                                          dv=self._pix_size_slice(0))
        except _lib.HipUnsupported:
            return None
        return m0.get()

    def moment(self, order=0, axis=0, how="auto", **kwargs):
        """Compute moments along an axis (spectral_cube.py:1614-1720;
        dask_spectral_cube.py:1031-1132).  ``how`` is accepted for interface
        compatibility; every strategy maps onto the same one-pass HIP kernel."""
        if how not in ("auto", "cube", "slice", "ray"):
            raise ValueError("Invalid how. Must be in %r" % sorted(["auto", "cube", "slice", "ray"]))
        if axis not in (0, 1, 2):
            raise ValueError("Cubes have 3 axes.")
        order = int(order)
        if order < 0:
            raise ValueError("order must be >= 0")
        if axis == 0 and order == 2:
            warnings.warn(_VARIANCE_MSG, VarianceWarning)
        if axis == 0:
            key = {0: "m0", 1: "m1", 2: "m2"}.get(order)
            alg = self._moments_of_spatially_smoothed((key,)) if key is not None else None
            if alg is not None:
                out = alg[key]
            elif key is not None:
                out = self._moment_device((key,), self._fusable())[key].get()
            else:
                r = self._moment_device(("mu", "s0"))
                cen = self._pix_cen_axis(0)
                cref = cen[self._shape[0] // 2]
                d_cen = DeviceArray.from_numpy(cen - cref, self.device)
                if self._wide_resident():
                    out = ops.moment_order_f64(self._device_data64(), d_cen, order, r["mu"], r["s0"], mask=self._mask_spec64()).get()
                else:
                    out = ops.moment_order(self._device_data(), d_cen, order, r["mu"], r["s0"], mask=self._mask_spec()).get()
                # dv cancels: sum(I dv (c-M1)^N) / sum(I dv)
            axunit = self.spectral_unit
        else:
            cen = self._pix_cen_axis(axis)
            d_cen = DeviceArray.from_numpy(cen, self.device)
            size = self._pix_size_slice(axis)
            key = {0: "m0", 1: "m1", 2: "m2"}.get(order)
            if self._stream_source() is not None:
                # out of core: the image planes are whole in a slab of channels, the (nz, nx) / (nz, ny) map grows slab by slab
                from . import streaming
                width, f64 = self._shape[2 if axis == 1 else 1], {"m0": np.float64, "m1": np.float64, "m2": np.float64, "o": np.float64}
                first = key or "m1"
                mu = streaming.slab_maps(self, lambda dev, ms, st, o, z0, z1: ops.moments_spatial(
                    dev, d_cen, axis, size, mask=ms, want=(first,), stream=st, out=o), (first,), width, f64)[first]
                if key is None:
                    mu = streaming.slab_maps(self, lambda dev, ms, st, o, z0, z1: ops.moment_order_spatial(
                        dev, d_cen, axis, order, streaming._rows_view(mu, z0, z1), mask=ms, stream=st, out=o["o"]), ("o",), width, f64)["o"]
                out = mu.get()
            elif key is None:          # order > 2: second pass about the first moment (_moments.py:185-193)
                mu = ops.moments_spatial(self._device_data(), d_cen, axis, size, mask=self._mask_spec(),
                                         want=("m1",))["m1"]
                out = ops.moment_order_spatial(self._device_data(), d_cen, axis, order, mu,
                                               mask=self._mask_spec()).get()
            else:
                out = ops.moments_spatial(self._device_data(), d_cen, axis, size, mask=self._mask_spec(),
                                          want=(key,))[key].get()
            axunit = self._wcs.cunit[2 - axis] if self._wcs is not None else ""
        if order == 0:
            unit = _unit_mul(self._unit, axunit)
        else:
            unit = _unit_pow(axunit, max(order, 1))
        meta = {"moment_order": order, "moment_axis": axis}
        meta.update(self._meta)
        new_wcs = self._wcs.drop_spectral() if (self._wcs is not None and axis == 0) else None
        beam = None if isinstance(self, VaryingResolutionSpectralCube) else self.beam
        return Projection(out, unit=unit, wcs=new_wcs, meta=meta, beam=beam, device=self.device)

    def moment0(self, axis=0, how="auto"):
        return self.moment(axis=axis, order=0, how=how)

    def moment1(self, axis=0, how="auto"):
        return self.moment(axis=axis, order=1, how=how)

    def moment2(self, axis=0, how="auto"):
        return self.moment(axis=axis, order=2, how=how)

    def moments012(self, keep_on_device=False):
        """moment 0, 1 and 2 along the spectral axis from ONE pass over the cube
        (the reference needs three graph executions, dask_spectral_cube.py
        :1090,1097,1104)."""
        r = self._moment_device(("m0", "m1", "m2"), self._fusable())
        if keep_on_device:
            return r["m0"], r["m1"], r["m2"]
        return tuple(self._wrap0(r[k].get(), o) for o, k in enumerate(("m0", "m1", "m2")))

    def _wrap0(self, arr, order):
        unit = _unit_mul(self._unit, self.spectral_unit) if order == 0 else _unit_pow(self.spectral_unit, max(order, 1))
        meta = {"moment_order": order, "moment_axis": 0}
        meta.update(self._meta)
        beam = None if isinstance(self, VaryingResolutionSpectralCube) else self.beam
        return Projection(arr, unit=unit, wcs=self._wcs.drop_spectral() if self._wcs is not None else None,
                          meta=meta, beam=beam, device=self.device)

    def linewidth_sigma(self, how="auto"):
        """sqrt(moment 2), no VarianceWarning (spectral_cube.py:1746-1753)."""
        with np.errstate(invalid="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore", VarianceWarning)
            m2 = self.moment(order=2, axis=0, how=how)
            out = np.sqrt(m2)
        return Projection(out, unit=self.spectral_unit, wcs=m2.wcs, meta=m2.meta)

    def linewidth_fwhm(self, how="auto"):
        """linewidth_sigma * sqrt(8 ln 2) (spectral_cube.py:1755-1763)."""
        s = self.linewidth_sigma(how=how)
        return Projection(np.asarray(s) * SIGMA2FWHM, unit=s.unit, wcs=s.wcs, meta=s.meta)

    def _extremum(self, key, axis, how):
        if axis == 0:
            return self._moment_device((key,))[key].get()
        if axis in (1, 2):
            if self._stream_source() is not None:
                from . import streaming
                return streaming.slab_maps(self, lambda dev, ms, st, o, z0, z1: ops.argextrema_axis(dev, axis, mask=ms, want=(key,), stream=st, out=o),
                                           (key,), self._shape[2 if axis == 1 else 1], {key: np.int64})[key].get()
            return ops.argextrema_axis(self._device_data(), axis, mask=self._mask_spec(), want=(key,))[key].get()
        if axis is not None:
            raise ValueError("axis must be None, 0, 1 or 2")
        # whole cube: flat C-order index of the first extreme voxel, from the per-spaxel extreme and
        # its first channel (one pass of the fused kernel)
        vkey = "vmax" if key == "argmax" else "vmin"
        r = self._moment_device((key, vkey))
        val, idx = r[vkey].get().astype(np.float64), r[key].get()
        fill = -np.inf if key == "argmax" else np.inf
        val = np.where(np.isnan(val), fill, val)               # rays without an included sample
        best = val.max() if key == "argmax" else val.min()
        if best == fill:
            return np.int64(0)
        nz, ny, nx = self._shape
        yy, xx = np.nonzero(val == best)
        return np.int64(np.min(idx[yy, xx] * (ny * nx) + yy * nx + xx))

    def argmax(self, axis=None, how="auto", **kwargs):
        """index of the maximum along *axis* (spectral_cube.py:793-804: nanargmax of the data filled
        with -inf); first index on ties, 0 for fully masked rays; int64.  ``axis=None``: flat index
        into the cube."""
        return self._extremum("argmax", axis, how)

    def argmin(self, axis=None, how="auto", **kwargs):
        """spectral_cube.py:806-819 (fill +inf)"""
        return self._extremum("argmin", axis, how)

    # ---- statistics / nan-reductions -----------------------------
    def _reduce(self, op, axis, ddof=0):
        """sum / mean / std / max / min (dask_spectral_cube.py:641-767): every statistic of a
        call comes out of ONE pass over the cube (the reference: one pass per statistic, two
        for nanstd)."""
        need = {"sum": ("count", "sum"), "mean": ("count", "sum"), "std": ("count", "sum", "sumsq"),
                "max": ("count", "max"), "min": ("count", "min")}[op]
        if axis is None:
            if self._stream_source() is not None:
                from . import streaming
                st = streaming.statistics(self)
            elif self._wide_resident():
                st = ops.stats_global_f64(self._device_data64(), mask=self._mask_spec64())
            else:
                st = ops.stats_global(self._device_data(), mask=self._mask_spec())
            n = st["npts"]
            vals = {"count": np.float64(n), "sum": np.float64(st["sum"]), "sumsq": np.float64(st["sumsq"]),
                    "max": np.float64(st["max"]), "min": np.float64(st["min"])}
        elif isinstance(axis, (tuple, list)):
            # two axes at once (e.g. the mean spectrum, axis=(1, 2)): one device pass removes the first
            # of them, the small (5 statistics x one map) remainder is finished on the host
            axes = sorted(set(int(a) for a in axis))
            if len(axes) == 3:
                return self._reduce(op, None, ddof)
            if len(axes) == 1:
                return self._reduce(op, axes[0], ddof)
            if len(axes) != 2 or any(a not in (0, 1, 2) for a in axes):
                raise ValueError("axis must be None, 0, 1, 2 or a tuple of these")
            if axes == [1, 2] and not (self._stream_source() is None and self._wide_resident()):
                # per channel (spectra): one dedicated pass, nz records (a float64 cube: its rows on the device, the rest here)
                if self._stream_source() is not None:
                    from . import streaming
                    vals = streaming.stats_planes(self)
                else:
                    vals = ops.stats_planes(self._device_data(), mask=self._mask_spec())
                axis = (1, 2)
                return self._finish_reduce(op, vals, axis, ddof)
            first, second = axes[1], axes[0]            # drop the higher axis on the device, the lower one here
            r = self._stats_axis_maps(first, need)
            part = {k: r[k].get().astype(np.float64) for k in need}
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                vals = {"count": part["count"].sum(axis=second)}
                for k in need:
                    if k in ("sum", "sumsq"):
                        vals[k] = np.nansum(part[k], axis=second)
                    elif k == "max":
                        vals[k] = np.nanmax(np.where(part["count"] > 0, part[k], -np.inf), axis=second)
                    elif k == "min":
                        vals[k] = np.nanmin(np.where(part["count"] > 0, part[k], np.inf), axis=second)
            axis = tuple(axes)
        else:
            if axis not in (0, 1, 2):
                raise ValueError("axis must be None, 0, 1 or 2")
            r = self._stats_axis_maps(axis, need)
            vals = {k: r[k].get().astype(np.float64) for k in need}
        return self._finish_reduce(op, vals, axis, ddof)

    def _stats_axis_maps(self, axis, need):
        if self._stream_source() is not None:          # out of core: strips (axis 0) or slabs of planes (axis 1 / 2)
            from . import streaming
            return streaming.stats_axis(self, axis, need)
        if self._wide_resident():
            return ops.stats_axis_f64(self._device_data64(), axis, mask=self._mask_spec64(), want=need)
        return ops.stats_axis(self._device_data(), axis, mask=self._mask_spec(), want=need)

    def _finish_reduce(self, op, vals, axis, ddof=0):
        n = vals["count"]
        with np.errstate(invalid="ignore", divide="ignore"):
            if op == "sum":
                out = np.where(n > 0, vals["sum"], np.nan)            # nansum_allbadtonan
            elif op == "mean":
                out = np.where(n > 0, vals["sum"] / n, np.nan)
            elif op == "std":
                var = (vals["sumsq"] - vals["sum"] * vals["sum"] / n) / (n - ddof)
                out = np.where((n > 0) & (n - ddof > 0), np.sqrt(np.maximum(var, 0.0)), np.nan)
            else:
                out = np.where(n > 0, vals[op], np.nan)
        if axis is None:
            return float(out)
        wcs = self._wcs.drop_spectral() if (axis == 0 and self._wcs is not None) else None
        return Projection(out, unit=self._unit, wcs=wcs, meta=dict(self._meta))

    def sum(self, axis=None, how="auto", **kwargs):
        """nansum_allbadtonan of the masked data (dask_spectral_cube.py:641-647)."""
        return self._reduce("sum", axis)

    def mean(self, axis=None, how="auto", **kwargs):
        """nanmean (dask_spectral_cube.py:649-655)."""
        return self._reduce("mean", axis)

    def std(self, axis=None, how="auto", ddof=0, **kwargs):
        """nanstd with ddof (dask_spectral_cube.py:695-709)."""
        return self._reduce("std", axis, ddof=ddof)

    def max(self, axis=None, how="auto", **kwargs):
        """nanmax (dask_spectral_cube.py:733-739)."""
        return self._reduce("max", axis)

    def min(self, axis=None, how="auto", **kwargs):
        """nanmin (dask_spectral_cube.py:741-747)."""
        return self._reduce("min", axis)

    def _order_stat(self, q, axis, what, center=None, scale=1.0):
        if axis not in (0, 1, 2):
            raise ValueError("axis must be None, 0, 1 or 2")
        if axis in (0, 1) and self._shape[axis] <= 4096 and self._runs_wide():
            # a float64 cube: sorted rays of float64 keys (spc_percentile_axis0_f64), a float64 map
            d64, ms = self._device_data64(), self._mask_spec64()
            if axis == 1:
                d64, ms = d64.swap01(), ms.swap01()
            return ops.percentile_axis0_f64(d64, q, mask=ms, center=center, scale=scale)
        if axis in (1, 2) and self._stream_source() is not None:
            # out of core: the rays along y / x are whole in a slab of channels; row z of the (nz, nx) / (nz, ny) map per slab row
            from . import streaming

            def one(dev, ms, st, o, z0, z1):
                cen = streaming._rows_view(center, z0, z1) if center is not None else None
                if axis == 1:
                    ops.percentile_axis0(dev.swap01(), q, mask=ms.swap01() if ms is not None else None, center=cen, scale=scale, stream=st, out=o["q"])
                    return
                try:
                    ops.percentile_axis2(dev, q, mask=ms, center=cen, scale=scale, stream=st, out=o["q"])
                except _lib.HipUnsupported:
                    flipped = ops.fill_masked_transposed(dev, ms, np.nan, stream=st)
                    ops.percentile_axis0(flipped.swap01(), q, center=cen, scale=scale, stream=st, out=o["q"])
                    st.synchronize()
            return streaming.slab_maps(self, one, ("q",), self._shape[2 if axis == 1 else 1], {"q": np.float32})["q"]
        if axis == 1:            # rays along y: the same kernels on a view with the first two axes exchanged
            return ops.percentile_axis0(self._device_data().swap01(), q, mask=self._mask_spec().swap01(),
                                        center=center, scale=scale)
        if axis == 2:            # rays along x: the rows themselves (contiguous samples, lanes walk along the ray) ...
            try:
                return ops.percentile_axis2(self._device_data(), q, mask=self._mask_spec(), center=center, scale=scale)
            except _lib.HipUnsupported:
                pass             # ... or, for rows of more than 4096 samples, a NaN-filled transposed copy, then as along y
            flipped = ops.fill_masked_transposed(self._device_data(), self._mask_spec(), np.nan)
            return ops.percentile_axis0(flipped.swap01(), q, center=center, scale=scale)
        if axis != 0:
            raise ValueError("axis must be None, 0, 1 or 2")
        if self._stream_source() is not None:       # out of core: per spaxel, so strip by strip
            from . import streaming
            return streaming.percentile_axis0(self, q, center=center, scale=scale)
        return ops.percentile_axis0(self._device_data(), q, mask=self._mask_spec(), center=center, scale=scale)

    def _order_wcs(self, axis):
        return self._wcs.drop_spectral() if (axis == 0 and self._wcs is not None) else None

    def _order_stat_global(self, q, center=None):
        """whole-cube statistic (axis=None): float32 like numpy's result for a float32 cube"""
        return np.float32(ops.percentile_global(self._device_data(), q, mask=self._mask_spec(), center=center))

    def median(self, axis=None, **kwargs):
        """nanmedian along an axis or of the whole cube (dask_spectral_cube.py:657-671): digit
        descent on order-preserving keys, no sort (csrc/spc_select.hip)."""
        if axis is None:
            return float(self._order_stat_global(50.0))
        return Projection(self._order_stat(50.0, axis, "median").get(), unit=self._unit,
                          wcs=self._order_wcs(axis), meta=dict(self._meta))

    def percentile(self, q, axis=None, **kwargs):
        """np.nanpercentile along an axis or of the whole cube (dask_spectral_cube.py:673-693)."""
        if axis is None:
            return float(self._order_stat_global(float(q)))
        return Projection(self._order_stat(float(q), axis, "percentile").get(), unit=self._unit,
                          wcs=self._order_wcs(axis), meta=dict(self._meta))

    def mad_std(self, axis=None, ignore_nan=True, **kwargs):
        """astropy mad_std along the spectral axis (dask_spectral_cube.py:711-731): two selections
        (median, then median of |x - median|) without leaving the device."""
        if axis is None:
            med = self._order_stat_global(50.0)
            return float(np.float32(self._order_stat_global(50.0, center=med) * np.float32(ops.MAD_TO_STD)))
        med = self._order_stat(50.0, axis, "mad_std")
        out = self._order_stat(50.0, axis, "mad_std", center=med, scale=ops.MAD_TO_STD)
        return Projection(out.get(), unit=self._unit, wcs=self._order_wcs(axis), meta=dict(self._meta))

    def sigma_clip_spectrally(self, threshold, **kwargs):
        """astropy's sigma clipper along the spectral axis, clipped (and excluded) values -> NaN
        (dask_spectral_cube.py:851-878); kwargs: maxiters, cenfunc ('median' | 'mean'), stdfunc
        ('std' | 'mad_std'), sigma_lower, sigma_upper.  Runs on the device (ops.sigma_clip_axis0)."""
        allowed = {"maxiters", "cenfunc", "stdfunc", "sigma_lower", "sigma_upper"}
        extra = set(kwargs) - allowed
        if extra:
            raise NotImplementedError("sigma_clip options not built on the device path: %s" % sorted(extra))
        if self._stream_source() is not None:
            # out of core: a pending operator; write() / stream_into() run it strip by strip
            parent, sig = self, float(threshold)
            thunk = _Thunk(lambda: ops.sigma_clip_axis0(parent._device_data(), sigma=sig, mask=parent._mask_spec(), **kwargs))
            thunk.parent = parent
            thunk.strip_fn = lambda dev, mspec, stream: ops.sigma_clip_axis0(dev, sigma=sig, mask=mspec, stream=stream, **kwargs)
            return self._new_cube_with(lazy=thunk, shape=self._shape)
        if self._shape[0] <= 4096 and self._runs_wide():
            parent, kw = self, dict(kwargs)
            return self._new_wide_cube(lambda: ops.sigma_clip_axis0_f64(parent._device_data64(), sigma=float(threshold),
                                                                        mask=parent._mask_spec64(), **kw))
        dev = ops.sigma_clip_axis0(self._device_data(), sigma=float(threshold), mask=self._mask_spec(), **kwargs)
        return self._new_cube_with(dev=dev)

    def statistics(self):
        """global basic statistics in ONE pass (dask_spectral_cube.py:769-814): npts, min, max,
        sum, sumsq, mean, sigma (the reference's textbook formula), rms."""
        if self._stream_source() is not None:
            from . import streaming
            return streaming.statistics(self)           # per-strip records, combined (same formulae)
        if self._wide_resident():
            st = ops.stats_global_f64(self._device_data64(), mask=self._mask_spec64())
        else:
            st = ops.stats_global(self._device_data(), mask=self._mask_spec())
        n = st["npts"]
        with np.errstate(invalid="ignore", divide="ignore"):
            st["mean"] = st["sum"] / n if n else np.nan
            st["sigma"] = float(np.sqrt((st["sumsq"] - st["sum"] ** 2 / n) / (n - 1))) if n > 1 else np.nan
            st["rms"] = float(np.sqrt(st["sumsq"] / n)) if n else np.nan
        return st

    # ---- smoothing ---------------------------------------------------------------------------
    def spectral_smooth(self, kernel, convolve=None, **kwargs):
        """Smooth along the spectral axis; the mask is left unchanged
        (dask_spectral_cube.py:880-917).  Lazy like the Dask class: a following
        ``moment`` runs fused with the stencil, any other access materialises."""
        karr = kernel_array(kernel, 1)
        _check_convolve(convolve)
        parent = self
        if self._runs_wide():
            # a float64 cube stays float64 (the Dask class keeps the chunk dtype, dask_spectral_cube.py:829)
            return self._new_wide_cube(lambda: ops.spectral_conv_f64(parent._device_data64(), karr, mask=parent._mask_spec64()))

        class _Lazy:
            op = "spectral_smooth"
            fusable = True

            def __init__(self):
                self.parent, self.kernel = parent, karr

            def __call__(self):
                return ops.spectral_conv(parent._device_data(), karr, mask=parent._mask_spec())

            def strip_fn(self, dev, mspec, stream):          # one row strip of an out-of-core parent
                return ops.spectral_conv(dev, karr, mask=mspec, stream=stream)

        return self._new_cube_with(lazy=_Lazy(), shape=self._shape)

    def spatial_smooth(self, kernel, convolve=None, raise_error_jybm=True, arithmetic=None, **kwargs):
        """Smooth every channel with a 2-D kernel (dask_spectral_cube.py:962-993).  Extra keyword
        arguments are accepted and not used - exactly what the Dask class does with them (its
        convolve_wrapper only receives ``kernel``, :990-993; spectral_smooth likewise, :912-917).
        ``arithmetic`` (this package's own): None, "f16-split" or "f32" - which arithmetic the MASKED separable stencil of a
        float32 cube runs in (ops.masked_spatial_arithmetic; DESIGN section 5: both are inside the 1e-5 contract, the
        reference computes in float64)."""
        self.check_jybeam_smoothing(raise_error_jybm=raise_error_jybm)
        karr = kernel_array(kernel, 2)
        _check_convolve(convolve)
        if arithmetic is not None and arithmetic not in ops.MASKED_SPATIAL_ARITHMETIC:
            raise ValueError("arithmetic must be one of %r, got %r" % (ops.MASKED_SPATIAL_ARITHMETIC, arithmetic))
        parent = self
        if self._runs_wide():
            return self._new_wide_cube(lambda: ops.spatial_conv_f64(parent._device_data64(), karr, mask=parent._mask_spec64()))

        class _Lazy:
            op = "spatial_smooth"
            fusable = False

            def __init__(self):
                self.parent, self.kernel, self.arithmetic = parent, karr, arithmetic

            def __call__(self):
                return ops.spatial_conv(parent._device_data(), karr, mask=parent._mask_spec(), arithmetic=arithmetic)

            halo = karr.shape[0] // 2        # rows a strip of an out-of-core parent needs from its neighbours

            def strip_fn(self, dev, mspec, stream):
                return ops.spatial_conv(dev, karr, mask=mspec, stream=stream, arithmetic=arithmetic)

            # cube -> cube (write / stream_into) of an out-of-core parent goes in slabs of whole planes instead: no halo rows
            # to read twice (29 x 29 taps, 8 GiB host array: 8.3 GB/s each way in halo strips, the link's rate in slabs); the
            # halo strips remain for a following moment, which needs whole spaxels
            keeps_mask = True

            def slab_fn(self, dev, mspec, stream):
                return ops.spatial_conv(dev, karr, mask=mspec, stream=stream, arithmetic=arithmetic)

        return self._new_cube_with(lazy=_Lazy(), shape=self._shape)

    @property
    def beam(self):
        """the restoring beam from BMAJ / BMIN / BPA (None when the header has no beam)."""
        from .beam import Beam
        if getattr(self, "_beam", None) is None:
            self._beam = Beam.from_header(self._header)
        return self._beam

    def with_beam(self, beam, raise_error_jybm=True):
        new = self._new_cube_with(data=self._data, dev=self._dev, lazy=self._lazy, shape=self._shape, same_data=True)
        new._beam = beam
        new._header = dict(self._header, BMAJ=beam.major, BMIN=beam.minor, BPA=beam.pa)
        return new

    @warn_slow
    def convolve_to(self, beam, convolve=None, **kwargs):
        """Convolve every channel to *beam* (dask_spectral_cube.py:1412-1464): kernel = beam.deconvolve(
        self.beam).as_kernel(pixscale) through the 2-D stencils (separable when the kernel is, else the
        LDS-tiled direct kernel), Jy/beam data scaled by the ratio of the beam areas."""
        _check_convolve(convolve)
        if self.beam is None:
            raise ValueError("cube has no beam (BMAJ / BMIN / BPA) to convolve from")
        if beam == self.beam:
            warnings.warn("The given beam is identical to the current beam. Skipping convolution.")
            return self
        if self._wcs is None:
            raise ValueError("convolve_to needs the celestial pixel scale of a WCS")
        psm = self._wcs.pixel_scale_matrix                     # proj_plane_pixel_area(wcs.celestial) ** 0.5
        pixscale = math.sqrt(abs(psm[0, 0] * psm[1, 1] - psm[0, 1] * psm[1, 0]))
        karr = beam.deconvolve(self.beam).as_kernel(pixscale)
        is_jybm = str(self._unit).replace(" ", "").upper() in ("JY/BEAM", "JYBEAM-1", "JY/BM")
        ratio = beam.sr / self.beam.sr if is_jybm else 1.0
        if self._runs_wide():
            parent = self

            def run64():
                res = ops.spatial_conv_f64(parent._device_data64(), karr, mask=parent._mask_spec64())
                return ops.scale_inplace_f64(res, ratio) if ratio != 1.0 else res
            return self._new_wide_cube(run64).with_beam(beam, raise_error_jybm=False)
        if self._stream_source() is not None:
            # out of core: one kernel for every channel, so slabs of whole planes; pending until write() / stream_into()
            parent = self

            def slab(dev, mspec, stream):
                res = ops.spatial_conv(dev, karr, mask=mspec, stream=stream)
                if ratio != 1.0:
                    ops.scale_inplace(res, ratio, stream=stream)
                return res
            thunk = _Thunk(lambda: parent._device_data())           # (never resident: raises HugeCubeError with the budget)
            thunk.parent, thunk.slab_fn, thunk.keeps_mask = parent, slab, True
            return self._new_cube_with(lazy=thunk, shape=self._shape).with_beam(beam, raise_error_jybm=False)
        dev = ops.spatial_conv(self._device_data(), karr, mask=self._mask_spec())
        if ratio != 1.0:
            ops.scale_inplace(dev, ratio)
        new = self._new_cube_with(dev=dev)
        return new.with_beam(beam, raise_error_jybm=False)

    def check_jybeam_smoothing(self, raise_error_jybm=True):
        """base_class.py:116-140"""
        if str(self._unit).replace(" ", "").upper() in ("JY/BEAM", "JYBEAM-1", "JY/BM"):
            if raise_error_jybm:
                raise BeamUnitsError("Attempting to change the spatial resolution of a cube with "
                                     "Jy/beam units. To ignore this error, set "
                                     "`raise_error_jybm=False`.")

    # ---- regridding -----------------------------------------------------------------------------
    def spectral_interpolate(self, spectral_grid, suppress_smooth_warning=False, fill_value=None,
                             force_rechunk=True):
        """Resample onto a linear spectral grid, Dask semantics
        (dask_spectral_cube.py:1250-1373): NaN outside the input range (or
        ``fill_value``), NaN when either bracketing channel is NaN/masked,
        new mask = ~isnan(result)."""
        grid = np.asarray(getattr(spectral_grid, "value", spectral_grid), dtype=np.float64)
        inaxis = self.spectral_axis
        lo, t, inv_dx, rin, rout, fill = ops.lerp_plan(inaxis, grid, fill_value)
        g_sorted = grid[::-1] if rout else grid
        in_sorted = inaxis[::-1] if rin else inaxis
        indiff = np.mean(np.diff(in_sorted))
        outdiff = np.mean(np.diff(g_sorted))
        if outdiff > 2 * indiff and not suppress_smooth_warning:
            warnings.warn("Input grid has too small a spacing. The data should "
                          "be smoothed prior to resampling.", SmoothingWarning)
        nz = self._shape[0]
        if rin:          # cubedata[::-1]: channel k of the flipped cube is nz-1-k
            lo_dev = np.where(lo >= 0, nz - 2 - lo, -1).astype(np.int32)
            # bracketing pair (lo, lo+1) of the flipped axis = (nz-1-lo, nz-2-lo) here:
            # interpolate from the upper neighbour downwards
            t_dev = (in_sorted[np.clip(lo, 0, nz - 2) + 1] - g_sorted)
            plan = (lo_dev, np.where(lo >= 0, t_dev, 0.0), inv_dx)
        else:
            plan = (lo, t, inv_dx)
        if rout:
            plan = tuple(p[::-1].copy() for p in plan)
        parent = self

        def run():
            return ops.spectral_lerp(parent._device_data(), plan[0], plan[1], plan[2], fill,
                                     mask=parent._mask_spec())

        crval = g_sorted[0] if not rout else g_sorted[-1]
        cdelt = outdiff if not rout else -outdiff
        newwcs = self._wcs.with_spectral(crval, cdelt, 1.0)
        if self._runs_wide():
            out = self._new_wide_cube(lambda: ops.spectral_lerp_f64(parent._device_data64(), plan[0], plan[1], plan[2], fill,
                                                                    mask=parent._mask_spec64()),
                                      shape=(len(grid),) + self._shape[1:], wcs=newwcs, mask=False)
            out._mask = M.NotNaNMask(out)
            return out
        thunk = _Thunk(run)
        thunk.parent = parent
        thunk.lerp = (plan, fill)          # (a following reproject folds the interpolation into its resampling kernel)
        thunk.strip_fn = lambda dev, mspec, stream: ops.spectral_lerp(dev, plan[0], plan[1], plan[2], fill, mask=mspec, stream=stream)
        out = self._new_cube_with(lazy=thunk, shape=(len(grid),) + self._shape[1:], wcs=newwcs,
                                  mask=False)
        out._mask = M.NotNaNMask(out)
        return out

    @warn_slow
    def reproject(self, header, order="bilinear", use_memmap=False, filled=True, **kwargs):
        """Reproject onto the WCS of *header* (spectral_cube.py:2649-2746 -> reproject_interp).

        The celestial plane of every channel is resampled at the source positions of the target pixels
        (``order`` 'bilinear' / 1 or 'nearest-neighbor' / 0; the pixel map is formed on the device by
        this package's WCS, validated against astropy.wcs).  Like the reference, a cube header also
        carries the output's spectral axis: ``NAXIS3`` channels at the target's spectral coordinates,
        linearly interpolated between the source channels (reproject_interp's one trilinear call for a
        separable cube WCS, restated as oracle_np.reproject_separable and pinned against scipy).  A
        2-axis header keeps this cube's spectral axis.  Masked voxels enter as ``fill_value`` when
        ``filled``; the new mask is the footprint; an output without a single non-NaN value raises the
        reference's ValueError."""
        order = {"nearest-neighbor": 0, "bilinear": 1, "biquadratic": 2, "bicubic": 3}.get(order, order)
        if order not in (0, 1, 2, 3):
            raise ValueError("order %r: 'nearest-neighbor' (0), 'bilinear' (1), 'biquadratic' (2) or 'bicubic' (3)" % (order,))
        newwcs = header if isinstance(header, SimpleWCS) else SimpleWCS(header)
        if self._wcs is not None:
            self._wcs._require_celestial()     # (a cube is read with a lenient WCS: keywords it does not model are refused here)
        newwcs._require_celestial()
        hdr = newwcs.header
        if "NAXIS1" in hdr and "NAXIS2" in hdr:
            ny_out, nx_out = int(hdr["NAXIS2"]), int(hdr["NAXIS1"])
        else:
            ny_out, nx_out = self._shape[1:]
        nz = self._shape[0]
        zs = None
        if newwcs.naxis >= 3 and self._wcs is not None and self._wcs.naxis >= 3:
            nz_out = int(hdr.get("NAXIS3", nz))
            check_same_spectral_kind(self._wcs, newwcs)
            scale = spectral_unit_scale(newwcs.spectral_unit or self.spectral_unit, self.spectral_unit)
            zs = self._wcs.spectral_world2pix(newwcs.spectral_pix2world(np.arange(nz_out)) * scale)
            if nz_out == nz and np.all(np.abs(zs - np.arange(nz)) <= 1e-9 * max(nz, 1)):
                zs = None                         # the same channels: a purely spatial reprojection
        elif self._wcs is not None and self._wcs.naxis >= 3:
            newwcs = join_celestial_spectral(newwcs, self._wcs, nz)
        if zs is not None and (order != 1 or nz < 2):
            raise NotImplementedError("resampling the spectral axis as well needs order='bilinear' and at least two channels "
                                      "(spectral_interpolate first, then reproject with order %r)" % (order,))
        if order >= 2 and self._stream_source() is not None:
            from . import streaming
            raise streaming.HugeCubeError("order 'biquadratic' / 'bicubic' of a cube above the HBM budget (SPC_HBM_BUDGET) is not built: "
                                          "the spline prefilter couples every sample of the cube")
        if os.environ.get("SPC_WCS_HOST_MAP", "0") == "1":         # cross-check: the numpy map (0.8 s per 1024^2 pixels)
            xs, ys = reproject_pixel_map(self._wcs, newwcs, (ny_out, nx_out))
            xs = np.where(np.isfinite(xs), xs, -1e30)
            ys = np.where(np.isfinite(ys), ys, -1e30)
        else:
            _lib.require_gpu()
            xs, ys = ops.wcs_pixel_map(self._wcs, newwcs, (ny_out, nx_out), self.device)
        if self._stream_source() is not None:
            # out of core: every channel is resampled on its own, so the cube goes through the device in slabs of whole
            # planes; the result stays pending until write() / stream_into() (it does not fit the budget either)
            from . import streaming
            if zs is not None:
                return self._reproject_streamed_with_spectral_axis(newwcs, zs, (ny_out, nx_out), order, filled)
            probe = DeviceArray.from_numpy(np.zeros((1,) + tuple(self._shape[1:]), np.float32), self.device)
            _, foot = ops.resample_bilinear(probe, xs, ys, fill=np.nan, order=order)
            footprint = foot.get().astype(bool)
            if not footprint.any():
                raise ValueError("All values in reprojected cube are nan.  This can be caused"
                                 " by an error in which coordinates do not 'round-trip'.  Try "
                                 "setting ``roundtrip_coords=False``.  You might also check "
                                 "whether the WCS transformation produces valid pixel->world "
                                 "and world->pixel coordinates in each axis.")
            parent, fill = self, float(self._fill_value)
            thunk = _Thunk(lambda: parent._device_data())           # (never resident: raises HugeCubeError with the budget)
            thunk.parent = parent
            thunk.slab_fn = lambda dev, mspec, stream: ops.resample_bilinear(dev, xs, ys, fill=fill, mask=mspec if filled else None,
                                                                            order=order, stream=stream, want_footprint=False)[0]
            shape = (nz, ny_out, nx_out)
            out = self._new_cube_with(lazy=thunk, wcs=newwcs, mask=False, shape=shape)
            out._mask = M.BooleanArrayMask(footprint[None], newwcs, shape=shape)
            out._footprint = footprint
            return out
        if order in (0, 1) and self._runs_wide():
            # a float64 cube: float64 weights and results (reproject_interp computes in float64); a cube header's own spectral
            # axis is blended from the resampled float64 planes afterwards
            flag = DeviceArray((1,), np.uint32, self.device)
            mask64 = self._mask_spec64() if filled else None
            fillv = float(self._fill_value)
            if filled and not np.isnan(fillv) and self._mask is not None and M.contains(self._mask, M.NotNaNMask) \
                    and not M.contains(self._mask, M.InvertedMask):
                inc = ops.mask_include_f64(self._device_data64(), mask64, nan_excluded=True)
                mask64 = ops.MaskSpec(_lib.MASK_ARRAY, 0.0, 0.0, inc)
            dev64, foot = ops.resample_bilinear_f64(self._device_data64(), xs, ys, fill=fillv, mask=mask64, order=order, any_valid=flag)
            footprint = foot.get().astype(bool)
            valid3d = footprint[None]
            nothing = int(flag.get()[0]) == 0
            if zs is not None:
                inside = (zs >= -0.5) & (zs <= nz - 0.5)
                zc = np.clip(np.where(inside, zs, 0.0), 0.0, nz - 1.0)
                z0 = np.minimum(np.floor(zc).astype(np.int64), nz - 2)
                dev64 = ops.spectral_lerp_f64(dev64, np.where(inside, z0, -1).astype(np.int32), zc - z0, np.ones(len(zs)), np.nan)
                if not inside.all():
                    valid3d = footprint[None] & inside[:, None, None]
                nothing = nothing or not inside.any()
            if nothing:
                raise ValueError(_ALL_NAN)
            out = self._new_wide_cube(lambda: dev64, shape=dev64.shape, wcs=newwcs, mask=False)
            out._mask = M.BooleanArrayMask(valid3d, newwcs, shape=dev64.shape)
            out._footprint = footprint
            return out
        folded = self._pending_interpolation(order) if zs is None else None
        if folded is not None:
            # spectral_interpolate(...).reproject(...): both are linear interpolations with NaN propagation, so they commute -
            # every INPUT plane is resampled once and the output channels are blended from neighbouring resampled planes in
            # the same kernel (ops.resample_bilinear_lerp): one read of the parent, one write of the result, the
            # interpolated cube is never formed
            parent, plan = folded
            flag = DeviceArray((1,), np.uint32, self.device)
            dev, foot = ops.resample_bilinear_lerp(parent._device_data(), xs, ys, plan[0], plan[1], plan[2], fill=np.nan,
                                                   mask=parent._mask_spec(), order=order, any_valid=flag)
            footprint = foot.get().astype(bool)
            if int(flag.get()[0]) == 0:
                raise ValueError(_ALL_NAN)
            out = self._new_cube_with(dev=dev, wcs=newwcs, mask=False, shape=dev.shape)
            out._mask = M.BooleanArrayMask(footprint[None], newwcs, shape=dev.shape)
            out._footprint = footprint
            return out
        mask = self._mask_spec() if filled else None
        if filled and not np.isnan(float(self._fill_value)) and self._mask is not None and M.contains(self._mask, M.NotNaNMask) \
                and not M.contains(self._mask, M.InvertedMask):
            # ~isnan(data) lowers to nothing (NaN samples never take part in a reduction), but HERE the excluded voxels are
            # replaced by a fill value that is a number (spectral_cube.py:2709-2712): the NaN samples must be named
            inc = ops.mask_include(self._device_data(), mask, nan_excluded=True)
            mask = ops.MaskSpec(_lib.MASK_ARRAY, 0.0, 0.0, inc)
        flag = DeviceArray((1,), np.uint32, self.device)
        if order >= 2:
            # scipy's recursive spline prefilter (which reproject_interp runs along all three axes of the NaN-filled cube)
            # turns ONE non-finite sample into NaN everywhere: the reference then raises "All values ... are nan"
            src = self._device_data()
            if mask is not None:
                src = ops.fill_masked(src, mask, float(self._fill_value))
            finite = ops.stats_global(src, mask=ops.MaskSpec(_lib.MASK_FINITE))["npts"]
            if int(finite) != int(np.prod(self._shape, dtype=np.int64)):
                raise ValueError("All values in reprojected cube are nan.  This can be caused"
                                 " by an error in which coordinates do not 'round-trip'.  Try "
                                 "setting ``roundtrip_coords=False``.  You might also check "
                                 "whether the WCS transformation produces valid pixel->world "
                                 "and world->pixel coordinates in each axis.")
            dev, foot = ops.resample_spline(src, xs, ys, order)
            flag = None
        else:
            zfold = False
            if zs is not None:
                # reproject_separable: channel positions inside [-0.5, nz - 0.5] are clipped to the cube,
                # each output channel is the linear blend of the two resampled planes around it
                inside = (zs >= -0.5) & (zs <= nz - 0.5)
                zc = np.clip(np.where(inside, zs, 0.0), 0.0, nz - 1.0)
                z0 = np.minimum(np.floor(zc).astype(np.int64), nz - 2)
                zlo = np.where(inside, z0, -1).astype(np.int32)
                zfold = ops.lerp_plan_folds(zlo) and os.environ.get("SPC_REPROJECT_FOLD", "1") != "0"
            if zfold:
                # ascending target channels: the blend rides in the resampling kernel (one pass, no resampled copy of the cube)
                dev, foot = ops.resample_bilinear_lerp(self._device_data(), xs, ys, zlo, zc - z0, np.ones(len(zs)),
                                                       fill=float(self._fill_value), mask=mask, order=order, any_valid=flag)
            else:
                dev, foot = ops.resample_bilinear(self._device_data(), xs, ys, fill=float(self._fill_value),
                                                  mask=mask, order=order, any_valid=flag)
        footprint = foot.get().astype(bool)
        valid3d = footprint[None]
        if zs is not None:
            if not inside.all():
                valid3d = footprint[None] & inside[:, None, None]
            if zfold:
                nothing = (not inside.any()) or int(flag.get()[0]) == 0
            else:
                dev = ops.spectral_lerp(dev, zlo, zc - z0, np.ones(len(zs)), np.nan)
                nothing = (not inside.any()) or ops.stats_global(dev)["npts"] == 0
        elif flag is None:
            nothing = not footprint.any()
        else:
            nothing = int(flag.get()[0]) == 0
        if nothing:
            raise ValueError("All values in reprojected cube are nan.  This can be caused"
                             " by an error in which coordinates do not 'round-trip'.  Try "
                             "setting ``roundtrip_coords=False``.  You might also check "
                             "whether the WCS transformation produces valid pixel->world "
                             "and world->pixel coordinates in each axis.")
        out = self._new_cube_with(dev=dev, wcs=newwcs, mask=False, shape=dev.shape)
        out._mask = M.BooleanArrayMask(valid3d, newwcs, shape=dev.shape)
        out._footprint = footprint
        return out

    def _pending_interpolation(self, order):
        """(parent, plan) when this cube is a spectral_interpolate result that has not been formed yet and a reprojection of
        order 0 / 1 may fold the interpolation in: the parent is resident (or fits), nothing replaced the NaN fill or the
        ~isnan mask in between, the plan's channels ascend"""
        lz = self._lazy
        if self._dev is not None or lz is None or getattr(lz, "lerp", None) is None or order not in (0, 1):
            return None
        if os.environ.get("SPC_REPROJECT_FOLD", "1") == "0":
            return None
        plan, fill = lz.lerp
        parent = lz.parent
        if not (np.isnan(fill) and np.isnan(float(self._fill_value))) or parent._stream_source() is not None:
            return None
        if not (isinstance(self._mask, M.NotNaNMask) and self._mask._data_ref._is_same_data(self)):
            return None
        if parent._shape[0] < 2 or not ops.lerp_plan_folds(plan[0]):
            return None
        return parent, plan

    def _reproject_streamed_with_spectral_axis(self, newwcs, zs, shape_yx, order, filled):
        """reproject of a cube above the HBM budget onto a CUBE header whose spectral axis differs from this cube's: the
        reference's one reproject_interp call (spectral_cube.py:2726-2732) resamples all three axes at once; for a separable
        WCS that is the celestial resampling of every source channel followed by the linear blend of the two resampled
        planes around every output channel (oracle_np.reproject_separable).  Out of core the two steps run one after the
        other: source slabs -> resampled planes in a float32 HOST array (nz x ny_out x nx_out: host memory, not HBM), then
        row strips of that array -> output channels, pending until write() / stream_into() / a reduction."""
        nz = self._shape[0]
        ny_out, nx_out = shape_yx
        celestial = join_celestial_spectral(newwcs, self._wcs, nz)          # the target's sky grid, this cube's channels
        spatial = self.reproject(celestial, order=order, filled=filled)      # (pending; raises the all-NaN ValueError itself)
        mid = np.empty((nz, ny_out, nx_out), dtype=np.float32)
        spatial.stream_into(mid)
        footprint = spatial._footprint
        midcube = SpectralCube(mid, wcs=celestial, device=self.device, allow_huge_operations=self.allow_huge_operations)
        inside = (zs >= -0.5) & (zs <= nz - 0.5)
        zc = np.clip(np.where(inside, zs, 0.0), 0.0, nz - 1.0)
        z0 = np.minimum(np.floor(zc).astype(np.int64), nz - 2)
        lo = np.where(inside, z0, -1).astype(np.int32)
        t, ones = zc - z0, np.ones(len(zs))
        if not inside.any():
            raise ValueError("All values in reprojected cube are nan.  This can be caused"
                             " by an error in which coordinates do not 'round-trip'.  Try "
                             "setting ``roundtrip_coords=False``.  You might also check "
                             "whether the WCS transformation produces valid pixel->world "
                             "and world->pixel coordinates in each axis.")
        thunk = _Thunk(lambda: ops.spectral_lerp(midcube._device_data(), lo, t, ones, np.nan))
        thunk.parent = midcube
        thunk.strip_fn = lambda dev, mspec, stream: ops.spectral_lerp(dev, lo, t, ones, np.nan, mask=mspec, stream=stream)
        shape = (len(zs), ny_out, nx_out)
        out = self._new_cube_with(lazy=thunk, wcs=newwcs, mask=False, shape=shape)
        out._mask = M.BooleanArrayMask(footprint[None] & inside[:, None, None], newwcs, shape=shape)
        out._footprint = footprint
        return out

    def __repr__(self):
        return "SpectralCube(hip) with shape=%s%s" % (self._shape, " and unit=%s" % self._unit if self._unit else "")


class BeamWarning(UserWarning):
    """spectral_cube.utils.BeamWarning"""


class NonFiniteBeamsWarning(UserWarning):
    """spectral_cube.utils.NonFiniteBeamsWarning"""


class VaryingResolutionSpectralCube(SpectralCube):
    """Cube with one beam per channel (spectral_cube.py:3776-3872, the Dask flavour at
    dask_spectral_cube.py:1467-1645).  What the accelerated path serves is ``convolve_to``: every
    channel is brought to a common beam with its own deconvolved kernel, runs of channels that share
    a beam going through the 2-D stencils in one launch.  Channels with non-finite beams are masked
    out, as the reference does; the beam-area checks around spectral reductions
    (base_class.py:673-790) are not mirrored."""

    def __init__(self, *args, beams=None, beam_table=None, goodbeams_mask=None, beam_threshold=0.01, **kw):
        if beams is None and beam_table is None:
            raise ValueError("Must give either a beam table or a list of beams to "
                             "initialize a VaryingResolutionSpectralCube")
        super().__init__(*args, **kw)
        if beam_table is not None:          # {"BMAJ", "BMIN", "BPA"} in degrees (io_fits.read_beams_table)
            beams = [Beam(a, b, p) for a, b, p in zip(beam_table["BMAJ"], beam_table["BMIN"], beam_table["BPA"])]
        beams = list(beams)
        if len(beams) != self._shape[0]:
            raise ValueError("Beam list must have same size as spectral dimension")
        self._beams = beams
        self.beam_threshold = beam_threshold
        good = np.array([b.isfinite for b in beams], dtype=bool)
        if goodbeams_mask is not None:
            good &= np.asarray(goodbeams_mask, dtype=bool)
        self._goodbeams_mask = good
        if not good.all():
            if goodbeams_mask is None:
                warnings.warn("There were {0} non-finite beams; layers with non-finite beams will be "
                              "masked out.".format(int(np.count_nonzero(~good))), NonFiniteBeamsWarning)
            bm = M.BooleanArrayMask(good[:, None, None], wcs=self._wcs, shape=self._shape)
            self._mask = bm if self._mask is None else (self._mask & bm)

    @property
    def beams(self):
        """the beams of the channels whose beam is good (base_class.py:497-501)"""
        return [b for b, g in zip(self._beams, self._goodbeams_mask) if g]

    @property
    def unmasked_beams(self):
        return self._beams

    @property
    def goodbeams_mask(self):
        return self._goodbeams_mask

    @property
    def beam(self):
        raise AttributeError("VaryingResolutionSpectralCubes have a `beams` list, not a single `beam`")

    def with_beams(self, beams, goodbeams_mask=None, raise_error_jybm=True):
        new = self._new_cube_with(data=self._data, dev=self._dev, lazy=self._lazy, shape=self._shape, same_data=True)
        beams = list(beams)
        if len(beams) != self._shape[0]:
            raise ValueError("Beam list must have same size as spectral dimension")
        new._beams = beams
        if goodbeams_mask is not None:
            new._goodbeams_mask = np.asarray(goodbeams_mask, dtype=bool)
        return new

    def _new_cube_with(self, **kw):
        new = SpectralCube._new_cube_with(self, **kw)
        if new._shape[0] == len(self._beams):
            new.__class__ = VaryingResolutionSpectralCube
            new._beams = self._beams
            new._goodbeams_mask = self._goodbeams_mask
            new.beam_threshold = self.beam_threshold
        return new

    def write(self, filename, overwrite=False, format=None):
        """the cube followed by its BEAMS table (hdulist, dask_spectral_cube.py:1493-1509)"""
        from . import io_fits
        SpectralCube.write(self, filename, overwrite=overwrite, format=format)
        io_fits.append_beams_table(os.fspath(filename), [b.major for b in self._beams],
                                   [b.minor for b in self._beams], [b.pa for b in self._beams])

    @warn_slow
    def convolve_to(self, beam, allow_smaller=False, convolve=None, **kwargs):
        """Convolve each channel to *beam* (dask_spectral_cube.py:1511-1630): channel k gets the kernel
        ``beam.deconvolve(beams[k]).as_kernel(pixscale)`` and, for Jy/beam data, the factor
        ``beam.sr / beams[k].sr``; channels whose beam is masked, equals the target, or (with
        ``allow_smaller``) cannot be deconvolved are passed through as filled data.  Returns a
        single-beam SpectralCube."""
        _check_convolve(convolve)
        if self._wcs is None:
            raise ValueError("convolve_to needs the celestial pixel scale of a WCS")
        psm = self._wcs.pixel_scale_matrix
        if psm[0, 1] != 0 or psm[1, 0] != 0:
            warnings.warn("The beams will produce convolution kernels that are not aware of any "
                          "misaligment between pixel and world coordinates, and there are off-diagonal "
                          "elements of the WCS spatial transformation matrix.  Unexpected results are "
                          "likely.", BeamWarning)
        pixscale = math.sqrt(abs(psm[0, 0] * psm[1, 1] - psm[0, 1] * psm[1, 0]))
        is_jybm = str(self._unit).replace(" ", "").upper() in ("JY/BEAM", "JYBEAM-1", "JY/BM")
        plans = []                                  # per channel: None (pass through) or (source beam, kernel, ratio)
        for bm, valid in zip(self._beams, self._goodbeams_mask):
            if not valid or beam == bm:
                plans.append(None)
                continue
            try:
                dk = beam.deconvolve(bm)
            except (BeamError, ValueError):
                if allow_smaller:
                    plans.append(None)
                    continue
                raise
            if plans and plans[-1] is not None and plans[-1][0] == bm:
                plans.append(plans[-1])             # same beam as the previous channel: share the launch
            else:
                plans.append((bm, dk.as_kernel(pixscale), beam.sr / bm.sr if is_jybm else 1.0))
        def runs(src, spec, zbase, stream=None):
            """channels zbase .. of the cube held in `src` (their mask in `spec`): runs of channels that share a plan, one launch each"""
            n = src.shape[0]
            out = DeviceArray(src.shape, np.float32, self.device)
            a = 0
            while a < n:
                b = a + 1
                while b < n and plans[zbase + b] is plans[zbase + a]:
                    b += 1
                o, m = out.planes(a, b), (spec.planes(a, b) if spec is not None else None)
                if plans[zbase + a] is None:
                    ops.fill_masked(src.planes(a, b), m, np.nan, out=o, stream=stream)
                else:
                    _, karr, ratio = plans[zbase + a]
                    ops.spatial_conv(src.planes(a, b), karr, mask=m, out=o, stream=stream)
                    if ratio != 1.0:
                        ops.scale_inplace(o, ratio, stream=stream)
                a = b
            return out

        if self._runs_wide():
            # a float64 cube: the same runs on the float64 samples (a pass-through channel = the 1 x 1 kernel: the filled sample)
            parent, dev_ = self, self.device

            def run64():
                src, spec = parent._device_data64(), parent._mask_spec64()
                n = src.shape[0]
                out = DeviceArray(src.shape, np.float64, dev_)
                a = 0
                while a < n:
                    b = a + 1
                    while b < n and plans[b] is plans[a]:
                        b += 1
                    o, m = out.planes(a, b), (spec.planes(a, b) if spec is not None else None)
                    if plans[a] is None:
                        ops.spatial_conv_f64(src.planes(a, b), np.ones((1, 1)), mask=m, out=o)
                    else:
                        _, karr, ratio = plans[a]
                        ops.spatial_conv_f64(src.planes(a, b), karr, mask=m, out=o)
                        if ratio != 1.0:
                            ops.scale_inplace_f64(o, ratio)
                    a = b
                return out
            return self._new_wide_cube(run64, plain=True).with_beam(beam, raise_error_jybm=False)
        if self._stream_source() is not None:
            # out of core: slabs of whole planes (every channel has its own kernel, the slab knows its first channel);
            # pending until write() / stream_into()
            parent = self
            thunk = _Thunk(lambda: parent._device_data())           # (never resident: raises HugeCubeError with the budget)
            thunk.parent, thunk.keeps_mask = parent, True
            thunk.slab_fn = lambda dev, mspec, stream: runs(dev, mspec, int(getattr(dev, "z0", 0)), stream)
            new = SpectralCube._new_cube_with(self, lazy=thunk, shape=self._shape)
            return new.with_beam(beam, raise_error_jybm=False)
        new = SpectralCube._new_cube_with(self, dev=runs(self._device_data(), self._mask_spec(), 0))
        return new.with_beam(beam, raise_error_jybm=False)

    def spectral_interpolate(self, *args, **kwargs):
        raise AttributeError("VaryingResolutionSpectralCubes can't be spectrally interpolated.  Convolve "
                             "to a common resolution with `convolve_to` before attempting spectral "
                             "interpolation.")

    def spectral_smooth(self, *args, **kwargs):
        raise AttributeError("VaryingResolutionSpectralCubes can't be spectrally smoothed.  Convolve to "
                             "a common resolution with `convolve_to` before attempting spectral "
                             "smoothed.")

    def __repr__(self):
        return "VaryingResolutionSpectralCube(hip) with shape=%s%s" % (
            self._shape, " and unit=%s" % self._unit if self._unit else "")


class _Thunk:
    op = "thunk"
    fusable = False

    def __init__(self, fn):
        self._fn = fn

    def __call__(self):
        return self._fn()
