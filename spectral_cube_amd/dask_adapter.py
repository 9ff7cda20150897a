"""Chunk functions for DaskSpectralCube's apply-per-chunk seams.

The reference exposes two operator-level seams:

* ``DaskSpectralCubeMixin.apply_function_parallel_spectral(function,
  accepts_chunks=True)`` (spectral_cube/dask_spectral_cube.py:555-638): the
  function receives a NaN-filled ``(nz, cy, cx)`` numpy chunk with the whole
  spectral axis and returns an array of the same shape (or a reduced one when
  the caller passes ``drop_axis=[0]``);
* ``apply_function_parallel_spatial(function, accepts_chunks=True)``
  (:502-552): ``(cz, ny, nx)`` chunks with whole image planes.

Each factory below returns such a function.  The chunk is staged to HBM, the
HIP kernel runs, the result comes back as numpy - so the functions are
re-entrant (no global state; one stream per call) and picklable (module-level
callables holding only numpy arrays), which the dask ``threads`` /
``processes`` schedulers need (dask_spectral_cube.py:278-312).  Masked voxels
arrive as NaN, so no mask is passed to the kernels.  Empty chunks are passed
through like the reference's wrappers do (:600-610).
"""
import numpy as np

from . import ops
from .device import DeviceArray


def _stage(chunk, device):
    return DeviceArray.from_numpy(np.ascontiguousarray(chunk, dtype=np.float32), device)


class SpectralSmoothChunk:
    """drop-in for the ``spectral_smooth`` chunk function
    (dask_spectral_cube.py:912-914)."""

    def __init__(self, kernel, device=0):
        self.kernel = np.asarray(getattr(kernel, "array", kernel), dtype=np.float64)
        self.device = device

    def __call__(self, chunk):
        if chunk.size == 0:
            return chunk
        out = ops.spectral_conv(_stage(chunk, self.device), self.kernel).get()
        return out.astype(chunk.dtype, copy=False)


class SpatialSmoothChunk:
    """drop-in for the ``spatial_smooth`` chunk function (:990-993, :540-547)."""

    def __init__(self, kernel, device=0):
        self.kernel = np.asarray(getattr(kernel, "array", kernel), dtype=np.float64)
        self.device = device

    def __call__(self, chunk, **kwargs):
        if chunk.size == 0:
            return chunk
        out = ops.spatial_conv(_stage(chunk, self.device), self.kernel).get()
        return out.astype(chunk.dtype, copy=False)


class SigmaClipChunk:
    """drop-in for the chunk function of ``sigma_clip_spectrally`` (dask_spectral_cube.py:851-878: astropy's
    ``sigma_clip(chunk, sigma=threshold, axis=0, masked=False, **kwargs)`` under
    ``apply_function_parallel_spectral(accepts_chunks=True)`` - the operation docs/dask.rst:176-275 times): the whole
    clip loop of a chunk is one kernel with the rays resident in registers."""

    def __init__(self, threshold, device=0, **kwargs):
        unknown = set(kwargs) - {"sigma_lower", "sigma_upper", "maxiters", "cenfunc", "stdfunc"}
        if unknown:
            raise NotImplementedError("sigma_clip options not on the device path: %s" % sorted(unknown))
        self.threshold, self.kwargs, self.device = float(threshold), kwargs, device

    def __call__(self, chunk, **ignored):
        if chunk.size == 0:
            return chunk
        out = ops.sigma_clip_axis0(_stage(chunk, self.device), sigma=self.threshold, **self.kwargs).get()
        return out.astype(chunk.dtype, copy=False)


class MomentChunk:
    """reduced chunk function (use with ``drop_axis=[0]``): moment map of a
    ``(nz, cy, cx)`` chunk.  ``pix_cen`` = offsets from channel 0, ``pix_size``
    and ``world0`` as in dask_spectral_cube.py:1083-1123."""

    def __init__(self, order, pix_cen, pix_size, world0=0.0, device=0):
        if order not in (0, 1, 2):
            raise ValueError("MomentChunk supports order 0, 1, 2")
        self.order = order
        self.pix_cen = np.asarray(pix_cen, dtype=np.float64)
        self.pix_size = float(pix_size)
        self.world0 = float(world0)
        self.device = device

    def __call__(self, chunk):
        if chunk.size == 0:
            return chunk.sum(axis=0)
        nz = chunk.shape[0]
        cref = self.pix_cen[nz // 2]
        key = ("m0", "m1", "m2")[self.order]
        r = ops.moments(_stage(chunk, self.device),
                        DeviceArray.from_numpy(self.pix_cen - cref, self.device),
                        dv=self.pix_size, m1_add=cref + self.world0, want=(key,))
        return r[key].get()


class SpectralInterpolateChunk:
    """drop-in for ``interp_wrapper`` (dask_spectral_cube.py:1342-1353)."""

    def __init__(self, inaxis, grid, fill_value=None, device=0):
        self.plan = ops.lerp_plan(inaxis, grid, fill_value)
        if self.plan[3] or self.plan[4]:
            raise ValueError("pass ascending axes; the caller flips the data like the reference does")
        self.device = device

    def __call__(self, chunk):
        if chunk.size <= 1:
            return chunk
        lo, t, inv, _, _, fill = self.plan
        return ops.spectral_lerp(_stage(chunk, self.device), lo, t, inv, fill).get()
