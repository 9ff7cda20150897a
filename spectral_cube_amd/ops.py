"""Functional layer over the C ABI: one Python function per entry point of
include/spcube_hip.h, operating on :class:`DeviceArray` cubes.

Each function names the reference code it stands in for (paths relative to the
reference tree).  There is no CPU fallback here.
"""
import ctypes as C
import os
import threading
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from .device import DeviceArray, _sh


@dataclass
class MaskSpec:
    """Device-evaluable mask (AND of the enabled terms); see SPC_MASK_* in
    include/spcube_hip.h and spectral_cube/masks.py:105-116, 425-435."""
    flags: int = 0
    thr_lo: float = 0.0
    thr_hi: float = 0.0
    array: Optional[DeviceArray] = None      # uint8, same shape as the cube

    def to_c(self):
        m = _lib.SpcMask()
        m.flags = self.flags
        m.thr_lo = self.thr_lo
        m.thr_hi = self.thr_hi
        m.row_stride = 0
        m.plane_stride = 0
        if self.flags & _lib.MASK_ARRAY:
            if self.array is None:
                raise ValueError("MASK_ARRAY set without an array")
            m.d_array = self.array.ptr
            m.row_stride = getattr(self.array, "row_stride", 0)
            m.plane_stride = getattr(self.array, "plane_stride", 0)
        return m

    def to_c64(self):
        """spc_mask_f64: the thresholds as they are (a float64 cube is compared in float64)"""
        m = _lib.SpcMask64()
        m.flags, m.thr_lo, m.thr_hi = self.flags, self.thr_lo, self.thr_hi
        m.row_stride = m.plane_stride = 0
        if self.flags & _lib.MASK_ARRAY:
            if self.array is None:
                raise ValueError("MASK_ARRAY set without an array")
            m.d_array = self.array.ptr
            m.row_stride = getattr(self.array, "row_stride", 0)
            m.plane_stride = getattr(self.array, "plane_stride", 0)
        return m

    def rows(self, y0, y1):
        """the same mask restricted to rows [y0, y1) (strided view of the array term)"""
        return MaskSpec(self.flags, self.thr_lo, self.thr_hi,
                        self.array.rows(y0, y1) if self.array is not None else None)

    def swap01(self):
        """the same mask for DeviceArray.swap01() views"""
        return MaskSpec(self.flags, self.thr_lo, self.thr_hi, self.array.swap01() if self.array is not None else None)

    def planes(self, z0, z1):
        """the same mask restricted to channels [z0, z1)"""
        return MaskSpec(self.flags, self.thr_lo, self.thr_hi,
                        self.array.planes(z0, z1) if self.array is not None else None)


def _cube_c(cube):
    if cube.dtype != np.float32 or len(cube.shape) != 3:
        raise TypeError("cube must be a float32 DeviceArray of shape (nz, ny, nx)")
    c = _lib.SpcCube()
    c.d_data = cube.ptr
    c.nz, c.ny, c.nx = cube.shape
    c.row_stride = getattr(cube, "row_stride", cube.shape[2])          # DeviceArray.rows() views
    c.plane_stride = getattr(cube, "plane_stride", cube.shape[1] * cube.shape[2])
    return c


def _mask_c(mask, cube):
    if mask is None:
        mask = MaskSpec()
    if mask.array is not None:
        if mask.array.shape != cube.shape or mask.array.dtype.itemsize != 1:
            raise ValueError("mask array must be 1-byte and match the cube shape")
    return mask.to_c()


def _kern(k):
    k = np.ascontiguousarray(k, dtype=np.float64).ravel()
    return k, k.ctypes.data_as(C.POINTER(C.c_double))


# ---- device scratch: one reusable buffer per (device, stream, host thread) ---------------------------
# The C ABI never allocates (include/spcube_hip.h): every call gets its scratch from the caller.  Calls
# queued on one stream run one after the other, so they can share ONE buffer that only ever grows; two
# streams, or two host threads (dask `threads` workers), get their own.  Growing replaces the buffer -
# the old one goes back through spc_free, which waits for the work still using it.
_ws_lock = threading.Lock()
_ws_cache = {}


def workspace(device, stream, kind, nz, ny, nx, p0=0, p1=0, explicit=None):
    """(c_void_p, nbytes) of a device scratch buffer large enough for entry point *kind* (a
    _lib.WS_* constant) - *explicit* (a DeviceArray) when the caller manages its own."""
    need = int(_lib.load().spc_workspace_bytes(int(kind), int(nz), int(ny), int(nx), int(p0), int(p1)))
    if explicit is not None:
        if explicit.nbytes < need:
            raise ValueError("workspace of %d bytes given, %d needed" % (explicit.nbytes, need))
        return C.c_void_p(explicit.ptr), explicit.nbytes
    h = _sh(stream)
    key = (device, getattr(h, "value", h) or 0, threading.get_ident())
    with _ws_lock:
        buf = _ws_cache.get(key)
        if buf is None or buf.nbytes < need:
            buf = DeviceArray((max(need, 1 << 16),), np.uint8, device)
            _ws_cache[key] = buf
    return C.c_void_p(buf.ptr), buf.nbytes


def release_workspaces():
    """drop every cached scratch buffer (they return to the pool of spc_malloc)"""
    with _ws_lock:
        _ws_cache.clear()


_WANT_ALL = ("m0", "m1", "m2")


def _moment_outputs(shape2d, device, want, out=None):
    types = dict(m0=np.float64, m1=np.float64, m2=np.float64, mu=np.float64, s0=np.float64,
                 argmax=np.int64, argmin=np.int64, vmax=np.float32, vmin=np.float32,
                 nvalid=np.int32)
    bufs = {}
    o = _lib.SpcMomentOutputs()
    for name in want:
        if name not in types:
            raise ValueError("unknown moment output %r" % name)
        if out is not None and name in out:
            bufs[name] = out[name]
            if bufs[name].shape != tuple(shape2d) or bufs[name].dtype != np.dtype(types[name]):
                raise ValueError("preallocated output %r has wrong shape/dtype" % name)
        else:
            bufs[name] = DeviceArray(shape2d, types[name], device)
        setattr(o, "d_" + name, bufs[name].ptr)
    o.out_row_stride = 0
    return o, bufs


def moments(cube, cen, dv=1.0, m1_add=0.0, mask=None, want=_WANT_ALL, stream=None, workspace=None,
            out=None):
    """Fused masked moment 0/1/2 (+argmax/argmin/max/min/count) along axis 0.

    Stands in for DaskSpectralCubeMixin.moment arithmetic
    (spectral_cube/dask_spectral_cube.py:1083-1104), moment_cubewise /
    slicewise / raywise (spectral_cube/_moments.py:30-193), allbadtonan
    (np_compat.py:3-27) and argmax/argmin (spectral_cube.py:793-819).

    cen: DeviceArray of nz float64 = pix_cen[z] - c_ref.  Returns a dict of
    (ny, nx) DeviceArrays for the names in *want*.
    """
    nz, ny, nx = cube.shape
    if cen.dtype != np.float64 or cen.shape != (nz,):
        raise TypeError("cen must be a float64 DeviceArray of length nz")
    o, bufs = _moment_outputs((ny, nx), cube.device, want, out)
    need = _lib.load().spc_moments_workspace_bytes(nz, ny, nx)
    ws = workspace
    if need and (ws is None or ws.nbytes < need):
        ws = DeviceArray((need,), np.uint8, cube.device)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    _lib.call("spc_moments_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), C.c_void_p(cen.ptr),
              float(dv), float(m1_add), C.byref(o), C.c_void_p(ws.ptr if ws is not None else 0),
              ws.nbytes if ws is not None else 0)
    bufs["_workspace"] = ws
    return bufs


_WANT_F64 = ("m0", "m1", "m2", "mu", "s0", "argmax", "argmin", "vmax", "vmin", "nvalid")


def _cube_c64(cube):
    if cube.dtype != np.float64 or len(cube.shape) != 3:
        raise TypeError("cube must be a float64 DeviceArray of shape (nz, ny, nx)")
    c = _lib.SpcCube()
    c.d_data = cube.ptr
    c.nz, c.ny, c.nx = cube.shape
    c.row_stride = getattr(cube, "row_stride", cube.shape[2])
    c.plane_stride = getattr(cube, "plane_stride", cube.shape[1] * cube.shape[2])
    return c


def _mask_c64(mask, cube):
    if mask is None:
        mask = MaskSpec()
    if mask.array is not None and (mask.array.shape != cube.shape or mask.array.dtype.itemsize != 1):
        raise ValueError("mask array must be 1-byte and match the cube shape")
    return mask.to_c64()


def moments_f64(cube, cen, dv=1.0, m1_add=0.0, mask=None, want=("m0", "m1", "m2"), stream=None):
    """Masked moment 0 / 1 / 2 (+ argmax / argmin / max / min / count) along axis 0 of a FLOAT64 cube, in the source's own
    precision: the reference keeps a float64 cube in float64 (masks.py:225) and so are its moment maps
    (_moments.py:30-193, dask_spectral_cube.py:1083-1104).  One pass gives S0, S1, the count and the extrema
    (spc_moments_f64); moment 2 is a second pass about the first moment (spc_moment_order_f64, the reference's own form,
    _moments.py:185-193).  The thresholds of *mask* are compared in float64; ``vmax`` / ``vmin`` are float64 maps."""
    nz, ny, nx = cube.shape
    if cen.dtype != np.float64 or cen.shape != (nz,):
        raise TypeError("cen must be a float64 DeviceArray of length nz")
    types = dict(m0=np.float64, m1=np.float64, mu=np.float64, s0=np.float64, argmax=np.int64, argmin=np.int64,
                 vmax=np.float64, vmin=np.float64, nvalid=np.int32)
    first = [n for n in want if n != "m2"]
    if "m2" in want:
        first += [n for n in ("mu", "s0") if n not in first]
    bufs, o = {}, _lib.SpcMomentOutputs64()
    for name in first:
        if name not in types:
            raise ValueError("unknown moment output %r" % name)
        bufs[name] = DeviceArray((ny, nx), types[name], cube.device)
        setattr(o, "d_" + name, bufs[name].ptr)
    o.out_row_stride = 0
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    _lib.call("spc_moments_f64", cube.device, _sh(stream), C.byref(c), C.byref(m), C.c_void_p(cen.ptr), float(dv), float(m1_add), C.byref(o))
    if "m2" in want:
        bufs["m2"] = moment_order_f64(cube, cen, 2, bufs["mu"], bufs["s0"], mask=mask, stream=stream)
    return {n: bufs[n] for n in want}


def moment_order_f64(cube, cen, order, mu, s0, mask=None, stream=None):
    """sum v (c - mu)^order / S0 of a float64 cube (dask_spectral_cube.py:1094-1099; _moments.py:185-193)"""
    nz, ny, nx = cube.shape
    out = DeviceArray((ny, nx), np.float64, cube.device)
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    _lib.call("spc_moment_order_f64", cube.device, _sh(stream), C.byref(c), C.byref(m), C.c_void_p(cen.ptr), int(order),
              C.c_void_p(mu.ptr), C.c_void_p(s0.ptr), C.c_void_p(out.ptr), 0)
    return out


# ---- the other operators of a float64 cube (spc_wide_ops.hip; the reference keeps float64, masks.py:225) -------------------
def stats_global_f64(cube, mask=None, stream=None):
    """stats_global of a float64 cube (spc_stats_global_f64): min / max are float64 samples, thresholds compared in float64"""
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    h = (C.c_double * 5)()
    ws, wsn = workspace(cube.device, stream, _lib.WS_STATS_GLOBAL_F64, *cube.shape)
    _lib.call("spc_stats_global_f64", cube.device, _sh(stream), C.byref(c), C.byref(m), h, ws, wsn)
    return {"npts": h[0], "min": h[1], "max": h[2], "sum": h[3], "sumsq": h[4]}


def stats_axis_f64(cube, axis, mask=None, want=("count", "min", "max", "sum", "sumsq"), stream=None):
    """stats_axis of a float64 cube (spc_stats_axis_f64): every map but the int32 count is float64"""
    if axis not in (0, 1, 2):
        raise ValueError("axis must be 0, 1 or 2")
    shp = tuple(n for i, n in enumerate(cube.shape) if i != axis)
    res = {k: DeviceArray(shp, np.int32 if k == "count" else np.float64, cube.device) for k in want}
    o = _lib.SpcStatsOutputs()
    for k in want:
        setattr(o, "d_" + k, res[k].ptr)
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    _lib.call("spc_stats_axis_f64", cube.device, _sh(stream), C.byref(c), C.byref(m), int(axis), C.byref(o))
    return res


def spectral_conv_f64(cube, kernel1d, mask=None, out=None, stream=None):
    """spectral_conv of a float64 cube: float64 in, float64 out (the Dask class keeps the chunk dtype,
    dask_spectral_cube.py:829, :880-917)"""
    if out is None:
        out = DeviceArray(cube.shape, np.float64, cube.device)
    k = np.ascontiguousarray(kernel1d, dtype=np.float64)
    if k.ndim != 1:
        raise ValueError("kernel must be 1-D")
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    ws, wsn = workspace(cube.device, stream, _lib.WS_SPECTRAL_CONV_F64, *cube.shape, len(k))
    _lib.call("spc_spectral_conv_f64", cube.device, _sh(stream), C.byref(c), C.byref(m), k.ctypes.data_as(C.POINTER(C.c_double)),
              len(k), C.c_void_p(out.ptr), 0, 0, ws, wsn)
    return out


def spatial_conv_f64(cube, kernel2d, mask=None, out=None, stream=None):
    """spatial_conv of a float64 cube (dask_spectral_cube.py:962-993): an outer-product kernel in two passes, any other
    kernel summed directly"""
    if out is None:
        out = DeviceArray(cube.shape, np.float64, cube.device)
    k2 = np.ascontiguousarray(kernel2d, dtype=np.float64)
    if k2.ndim != 2:
        raise ValueError("kernel must be 2-D")
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    sep = separable_factors(k2)
    ws, wsn = workspace(cube.device, stream, _lib.WS_SPATIAL_CONV_F64, *cube.shape, k2.shape[0], k2.shape[1])
    if sep is not None:
        ky, kx = (np.ascontiguousarray(f, dtype=np.float64) for f in sep)
        _lib.call("spc_spatial_conv_f64", cube.device, _sh(stream), C.byref(c), C.byref(m), ky.ctypes.data_as(C.POINTER(C.c_double)),
                  len(ky), kx.ctypes.data_as(C.POINTER(C.c_double)), len(kx), 1, C.c_void_p(out.ptr), 0, 0, ws, wsn)
    else:
        _lib.call("spc_spatial_conv_f64", cube.device, _sh(stream), C.byref(c), C.byref(m), k2.ctypes.data_as(C.POINTER(C.c_double)),
                  k2.shape[0], None, k2.shape[1], 0, C.c_void_p(out.ptr), 0, 0, ws, wsn)
    return out


def spectral_lerp_f64(cube, lo, t, inv_dx, fill=np.nan, mask=None, out=None, stream=None):
    """spectral_lerp of a float64 cube (scipy's interp1d on float64 samples, dask_spectral_cube.py:1342-1353)"""
    nz_out = len(lo)
    dev = cube.device
    d_lo = DeviceArray.from_numpy(np.asarray(lo, dtype=np.int32), dev)
    d_t = DeviceArray.from_numpy(np.asarray(t, dtype=np.float64), dev)
    d_inv = DeviceArray.from_numpy(np.asarray(inv_dx, dtype=np.float64), dev)
    if out is None:
        out = DeviceArray((nz_out,) + cube.shape[1:], np.float64, dev)
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    _lib.call("spc_spectral_lerp_f64", dev, _sh(stream), C.byref(c), C.byref(m), nz_out, C.c_void_p(d_lo.ptr), C.c_void_p(d_t.ptr),
              C.c_void_p(d_inv.ptr), float(fill), C.c_void_p(out.ptr), 0, 0)
    out._plan = (d_lo, d_t, d_inv)
    return out


def resample_bilinear_f64(cube, xs, ys, fill=np.nan, mask=None, stream=None, want_footprint=True, order=1, any_valid=None):
    """resample_bilinear of a float64 cube (spc_resample_bilinear_f64): float64 weights, float64 result"""
    dev = cube.device
    if isinstance(xs, DeviceArray) and isinstance(ys, DeviceArray):
        d_xs, d_ys = xs, ys
        ny_out, nx_out = xs.shape
    else:
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        ny_out, nx_out = xs.shape
        d_xs, d_ys = DeviceArray.from_numpy(xs, dev), DeviceArray.from_numpy(ys, dev)
    out = DeviceArray((cube.shape[0], ny_out, nx_out), np.float64, dev)
    foot = DeviceArray((ny_out, nx_out), np.uint8, dev) if want_footprint else None
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    _lib.call("spc_resample_bilinear_f64", dev, _sh(stream), C.byref(c), C.byref(m), float(fill), ny_out, nx_out,
              C.c_void_p(d_xs.ptr), C.c_void_p(d_ys.ptr), C.c_void_p(out.ptr), 0, 0,
              C.c_void_p(foot.ptr) if foot is not None else None, int(order),
              C.c_void_p(any_valid.ptr) if any_valid is not None else None)
    out._plan = (d_xs, d_ys)
    return out, foot


def scale_inplace_f64(arr, factor, stream=None):
    """arr *= factor for a float64 DeviceArray (spc_scale_f64)"""
    _lib.call("spc_scale_f64", arr.device, _sh(stream), C.c_void_p(arr.ptr), int(np.prod(arr.shape, dtype=np.int64)), float(factor))
    return arr


def percentile_axis0_f64(cube, q, mask=None, center=None, scale=1.0, stream=None):
    """percentile_axis0 of a float64 cube (spc_percentile_axis0_f64): float64 map; *center* a float64 (ny, nx) DeviceArray.
    Rays along y: the swap01() view (as for the float32 kernel).  HipUnsupported beyond 4096 samples per ray."""
    nz, ny, nx = cube.shape
    out = DeviceArray((ny, nx), np.float64, cube.device)
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    _lib.call("spc_percentile_axis0_f64", cube.device, _sh(stream), C.byref(c), C.byref(m), float(q),
              C.c_void_p(center.ptr) if center is not None else None, float(scale), C.c_void_p(out.ptr))
    return out


def sigma_clip_axis0_f64(cube, sigma=3.0, sigma_lower=None, sigma_upper=None, maxiters=5, cenfunc="median", stdfunc="std", mask=None,
                         stream=None):
    """sigma_clip_axis0 of a float64 cube (spc_sigma_clip_axis0_f64): float64 centre, spread and bounds"""
    if cenfunc not in ("median", "mean") or stdfunc not in ("std", "mad_std"):
        raise ValueError("cenfunc must be 'median' or 'mean', stdfunc 'std' or 'mad_std'")
    out = DeviceArray(cube.shape, np.float64, cube.device)
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    lo = float(sigma if sigma_lower is None else sigma_lower)
    hi = float(sigma if sigma_upper is None else sigma_upper)
    _lib.call("spc_sigma_clip_axis0_f64", cube.device, _sh(stream), C.byref(c), C.byref(m), lo, hi,
              -1 if maxiters is None else int(maxiters), 1 if cenfunc == "mean" else 0, 1 if stdfunc == "mad_std" else 0, C.c_void_p(out.ptr))
    return out


def narrow_f64(cube, stream=None):
    """float32 copy of a float64 DeviceArray (for the operators without a float64 form)"""
    out = DeviceArray(cube.shape, np.float32, cube.device)
    nz = cube.shape[0]
    for z0 in range(0, nz, 65535):
        z1 = min(nz, z0 + 65535)
        c = _cube_c64(cube.planes(z0, z1) if (z0, z1) != (0, nz) else cube)
        _lib.call("spc_narrow_f64_to_f32", cube.device, _sh(stream), C.byref(c),
                  C.c_void_p(out.ptr + z0 * cube.shape[1] * cube.shape[2] * 4), 0, 0)
    return out


def mask_include_f64(cube, mask=None, nan_excluded=False, stream=None):
    """mask_include on a float64 cube (spc_mask_include_f64)"""
    out = DeviceArray(cube.shape, np.uint8, cube.device)
    c, m = _cube_c64(cube), _mask_c64(mask, cube)
    _lib.call("spc_mask_include_f64", cube.device, _sh(stream), C.byref(c), C.byref(m), 1 if nan_excluded else 0, C.c_void_p(out.ptr))
    return out


def _given(out, name, shape, dtype, device):
    """out[name] when the caller preallocated it (checked), else a new DeviceArray"""
    a = out.get(name) if out else None
    if a is None:
        return DeviceArray(shape, dtype, device)
    if tuple(a.shape) != tuple(shape) or a.dtype != np.dtype(dtype):
        raise ValueError("preallocated output %r must be %s %s" % (name, tuple(shape), np.dtype(dtype)))
    return a


def argextrema_axis(cube, axis, mask=None, want=("argmax", "argmin"), stream=None, out=None):
    """argmax / argmin along a spatial axis (spectral_cube.py:793-819 with axis = 1 or 2): int64
    maps (nz, nx) / (nz, ny); first index on ties, 0 for rays without an included sample."""
    nz, ny, nx = cube.shape
    if axis not in (1, 2):
        raise ValueError("axis must be 1 or 2 (axis 0 comes out of ops.moments)")
    shape = (nz, nx) if axis == 1 else (nz, ny)
    bufs = {n: _given(out, n, shape, np.int64, cube.device) for n in want if n in ("argmax", "argmin")}
    c, m = _cube_c(cube), _mask_c(mask, cube)
    ptr = lambda n: C.c_void_p(bufs[n].ptr) if n in bufs else None  # noqa: E731
    _lib.call("spc_argextrema_axis_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), int(axis),
              ptr("argmin"), ptr("argmax"))
    return bufs


def moment_order(cube, cen, order, mu, s0, mask=None, stream=None):
    """sum v*(c-mu)^order / S0 - second pass for order > 2
    (dask_spectral_cube.py:1094-1099)."""
    nz, ny, nx = cube.shape
    out = DeviceArray((ny, nx), np.float64, cube.device)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    _lib.call("spc_moment_order_f32", cube.device, _sh(stream), C.byref(c), C.byref(m),
              C.c_void_p(cen.ptr), int(order), C.c_void_p(mu.ptr), C.c_void_p(s0.ptr),
              C.c_void_p(out.ptr), 0)
    return out


def moments_spatial(cube, cen2d, axis, pix_size, mask=None, want=_WANT_ALL, stream=None, out=None):
    """moment 0/1/2 along a spatial axis (golden tables
    spectral_cube/tests/test_moments.py:19-49)."""
    nz, ny, nx = cube.shape
    if axis not in (1, 2):
        raise ValueError("axis must be 1 or 2")
    shape = (nz, nx) if axis == 1 else (nz, ny)
    bufs = {n: _given(out, n, shape, np.float64, cube.device) for n in want}
    c, m = _cube_c(cube), _mask_c(mask, cube)
    ptr = lambda n: C.c_void_p(bufs[n].ptr) if n in bufs else None  # noqa: E731
    _lib.call("spc_moments_spatial_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), int(axis),
              C.c_void_p(cen2d.ptr), float(pix_size), ptr("m0"), ptr("m1"), ptr("m2"))
    return bufs


def moment_order_spatial(cube, cen2d, axis, order, mu, mask=None, stream=None, out=None):
    """sum v (c - mu)^order / sum v along a spatial axis: the second pass for order > 2
    (dask_spectral_cube.py:1094-1099 with axis != 0); *mu* = the m1 map of moments_spatial."""
    nz, ny, nx = cube.shape
    if axis not in (1, 2):
        raise ValueError("axis must be 1 or 2")
    out = _given({"o": out} if out is not None else None, "o", (nz, nx) if axis == 1 else (nz, ny), np.float64, cube.device)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    _lib.call("spc_moment_order_spatial_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), int(axis),
              C.c_void_p(cen2d.ptr), int(order), C.c_void_p(mu.ptr), C.c_void_p(out.ptr))
    return out


def spectral_conv(cube, kernel1d, mask=None, out=None, stream=None):
    """NaN-aware convolution along the spectral axis = chunk function of
    spectral_smooth (dask_spectral_cube.py:880-917)."""
    if out is None:
        out = DeviceArray(cube.shape, np.float32, cube.device)
    k, kp = _kern(kernel1d)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    ws, wsn = workspace(cube.device, stream, _lib.WS_SPECTRAL_CONV, *cube.shape, len(k))
    _lib.call("spc_spectral_conv_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), kp, len(k),
              C.c_void_p(out.ptr), 0, 0, ws, wsn)
    return out


def spectral_conv_moments(cube, kernel1d, cen, dv=1.0, m1_add=0.0, mask=None, want=_WANT_ALL,
                          stream=None, out=None, cen_host=None):
    """fused spectral_smooth -> moment (smoothed cube never written).
    cen_host: optional host copy of *cen* (lets the kernel use the linear-axis form)."""
    nz, ny, nx = cube.shape
    o, bufs = _moment_outputs((ny, nx), cube.device, want, out)
    k, kp = _kern(kernel1d)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    hc = None
    if cen_host is not None:
        cen_host = np.ascontiguousarray(cen_host, dtype=np.float64)
        hc = cen_host.ctypes.data_as(C.POINTER(C.c_double))
    ws, wsn = workspace(cube.device, stream, _lib.WS_SPECTRAL_CONV_MOMENTS, *cube.shape, len(k))
    _lib.call("spc_spectral_conv_moments_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), kp,
              len(k), C.c_void_p(cen.ptr), hc, float(dv), float(m1_add), C.byref(o), ws, wsn)
    return bufs


def separable_factors(kernel2d, rtol=1e-12):
    """Return (ky, kx) with outer(ky, kx) == kernel2d, or None."""
    k = np.asarray(kernel2d, dtype=np.float64)
    iy, ix = np.unravel_index(np.argmax(np.abs(k)), k.shape)
    piv = k[iy, ix]
    if piv == 0:
        return None
    ky = k[:, ix].copy()
    kx = k[iy, :] / piv
    if not np.allclose(np.outer(ky, kx), k, rtol=rtol, atol=rtol * abs(piv)):
        return None
    if piv > 0:
        # split the scale evenly: for a symmetric kernel (every Gaussian2DKernel with one stddev)
        # both factors then coincide, and handing the library IDENTICAL arrays lets it pick the
        # kernels that keep one set of weights in scalar registers for both passes
        s = np.sqrt(piv)
        ky, kx = ky / s, kx * s
        if ky.shape == kx.shape and np.allclose(ky, kx, rtol=rtol, atol=rtol * np.abs(ky).max()):
            kx = ky.copy()
    return ky, kx


MASKED_SPATIAL_ARITHMETIC = ("f16-split", "f32")
_arith_lock = threading.Lock()


class masked_spatial_arithmetic:
    """Scope in which the masked separable spatial stencil runs in the named arithmetic (the precision
    policy): "f16-split" (default) = every product on the fp16 matrix instruction with the samples, the taps and the x-pass
    result split hi + lo, float32 accumulation - 1e-6 of the data range against astropy's float64, inside the 1e-5 contract;
    "f32" = the ring kernels, float32 multiply-adds on the vector ALU - 2.5e-7 of the range, 1.2 - 1.5 x the time.
    (The library reads SPC_SPATIAL_RING per call; this sets it for the calls made inside the scope, one scope at a time.)"""

    def __init__(self, name):
        if name is not None and name not in MASKED_SPATIAL_ARITHMETIC:
            raise ValueError("arithmetic must be one of %r, got %r" % (MASKED_SPATIAL_ARITHMETIC, name))
        self.name = name

    def __enter__(self):
        if self.name is None:
            return self
        _arith_lock.acquire()
        self._old = os.environ.get("SPC_SPATIAL_RING")
        os.environ["SPC_SPATIAL_RING"] = "1" if self.name == "f32" else "0"
        return self

    def __exit__(self, *exc):
        if self.name is None:
            return False
        if self._old is None:
            os.environ.pop("SPC_SPATIAL_RING", None)
        else:
            os.environ["SPC_SPATIAL_RING"] = self._old
        _arith_lock.release()
        return False


def spatial_conv(cube, kernel2d, mask=None, out=None, stream=None, arithmetic=None):
    """NaN-aware per-channel 2-D convolution = chunk function of
    spatial_smooth (dask_spectral_cube.py:962-993, :540-547).  *arithmetic*: None (the library's default) or one of
    MASKED_SPATIAL_ARITHMETIC for the masked separable stencil (see masked_spatial_arithmetic)."""
    if arithmetic is not None:
        with masked_spatial_arithmetic(arithmetic):
            return spatial_conv(cube, kernel2d, mask=mask, out=out, stream=stream)
    if out is None:
        out = DeviceArray(cube.shape, np.float32, cube.device)
    k2 = np.ascontiguousarray(kernel2d, dtype=np.float64)
    if k2.ndim != 2:
        raise ValueError("kernel must be 2-D")
    c, m = _cube_c(cube), _mask_c(mask, cube)
    sep = separable_factors(k2)
    if sep is not None:
        (ky, kyp), (kx, kxp) = _kern(sep[0]), _kern(sep[1])
        ws, wsn = workspace(cube.device, stream, _lib.WS_SPATIAL_CONV_SEP, *cube.shape, len(ky), len(kx))
        _lib.call("spc_spatial_conv_sep_f32", cube.device, _sh(stream), C.byref(c), C.byref(m),
                  kyp, len(ky), kxp, len(kx), C.c_void_p(out.ptr), 0, 0, ws, wsn)
    else:
        k, kp = _kern(k2)
        ws, wsn = workspace(cube.device, stream, _lib.WS_SPATIAL_CONV2D, *cube.shape, k2.shape[0], k2.shape[1])
        _lib.call("spc_spatial_conv2d_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), kp,
                  k2.shape[0], k2.shape[1], C.c_void_p(out.ptr), 0, 0, ws, wsn)
    return out


def lerp_plan(inaxis, grid, fill_value=None):
    """Host-side plan for spectral_lerp restating scipy interp1d's index
    arithmetic (dask_spectral_cube.py:1291-1353).  Returns
    (lo int32, t float64, inv_dx float64, reverse_in, reverse_out, fill)."""
    inaxis = np.asarray(inaxis, dtype=np.float64)
    grid = np.asarray(grid, dtype=np.float64)
    reverse_in = np.mean(np.diff(inaxis)) < 0
    reverse_out = np.mean(np.diff(grid)) < 0
    if reverse_in:
        inaxis = inaxis[::-1]
    if reverse_out:
        grid = grid[::-1]
    if not (np.all(np.diff(grid) > 0) and np.all(np.diff(inaxis) > 0)):
        raise AssertionError("spectral axes must be strictly monotonic")
    np.testing.assert_allclose(np.diff(grid), np.mean(np.diff(grid)),
                               err_msg="Output grid must be linear")
    idx = np.clip(np.searchsorted(inaxis, grid), 1, len(inaxis) - 1)
    lo = (idx - 1).astype(np.int32)
    t = grid - inaxis[lo]
    inv_dx = 1.0 / (inaxis[lo + 1] - inaxis[lo])
    oob = (grid < inaxis[0]) | (grid > inaxis[-1])
    lo[oob] = -1
    fill = np.nan if fill_value is None else float(fill_value)
    return lo, t, inv_dx, bool(reverse_in), bool(reverse_out), fill


def spectral_lerp(cube, lo, t, inv_dx, fill=np.nan, mask=None, out=None, stream=None):
    """per-spaxel linear interpolation onto nz_out channels (chunk function
    of spectral_interpolate, dask_spectral_cube.py:1342-1353).  *cube* must
    already be in ascending spectral order."""
    nz_out = len(lo)
    dev = cube.device
    d_lo = DeviceArray.from_numpy(np.asarray(lo, dtype=np.int32), dev)
    d_t = DeviceArray.from_numpy(np.asarray(t, dtype=np.float64), dev)
    d_inv = DeviceArray.from_numpy(np.asarray(inv_dx, dtype=np.float64), dev)
    if out is None:
        out = DeviceArray((nz_out,) + cube.shape[1:], np.float32, dev)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    _lib.call("spc_spectral_lerp_f32", dev, _sh(stream), C.byref(c), C.byref(m), nz_out,
              C.c_void_p(d_lo.ptr), C.c_void_p(d_t.ptr), C.c_void_p(d_inv.ptr), float(fill),
              C.c_void_p(out.ptr), 0, 0)
    out._plan = (d_lo, d_t, d_inv)     # keep alive until the stream has consumed them
    return out


def _wcs_struct(w):
    code, crpix, lin, inv, ap, dp, php, pv1, plane0, (sip_order, sip_a, sip_b) = w.celestial_params()
    s = _lib.SpcCelestialWcs()
    s.proj = code
    s.pv1 = pv1
    s.plane0[0], s.plane0[1] = plane0
    s.sip_order = int(sip_order)
    for i in range(len(sip_a)):
        s.sip_a[i], s.sip_b[i] = sip_a[i], sip_b[i]
    s.crpix[0], s.crpix[1] = crpix
    for i in range(4):
        s.lin[i], s.lin_inv[i] = lin[i], inv[i]
    s.alpha_p, s.delta_p, s.phi_p = ap, dp, php
    return s


def wcs_pixel_map(wcs_in, wcs_out, shape_out, device=0, stream=None):
    """(xs, ys) float64 DeviceArrays: source-grid pixel coordinates of every pixel of the target grid,
    computed on the device (spc_wcs_pixel_map_f64; the host version is wcs.reproject_pixel_map).
    Pixels that cannot be projected hold -1e30."""
    ny, nx = (int(n) for n in shape_out)
    d_xs, d_ys = DeviceArray((ny, nx), np.float64, device), DeviceArray((ny, nx), np.float64, device)
    so, si = _wcs_struct(wcs_out), _wcs_struct(wcs_in)
    # target frame -> source frame (ICRS / FK5 / Galactic; NotImplementedError for pairs that are not built; None = same)
    # (ABI 4: 15 doubles - rotation, E-terms removed before it (FK4 target), E-terms added after it (FK4 source))
    from .wcs import frame_transform
    tr = frame_transform(getattr(wcs_out, "frame", None), getattr(wcs_in, "frame", None))
    rp = None
    if tr is not None:
        remove, rot, add = tr
        vals = list(np.ascontiguousarray(rot, dtype=np.float64).ravel())
        vals += list(remove) if remove is not None else [0.0, 0.0, 0.0]
        vals += list(add) if add is not None else [0.0, 0.0, 0.0]
        rp = (C.c_double * 15)(*vals)
    _lib.call("spc_wcs_pixel_map_f64", device, _sh(stream), C.byref(so), C.byref(si), rp, ny, nx,
              C.c_void_p(d_xs.ptr), C.c_void_p(d_ys.ptr))
    return d_xs, d_ys


def resample_bilinear(cube, xs, ys, fill=np.nan, mask=None, stream=None, want_footprint=True, out=None,
                      order=1, any_valid=None):
    """bilinear spatial resample of every channel at (xs, ys) source pixel
    coordinates (resampler of reproject_interp, spectral_cube.py:2726-2732).  xs, ys: host arrays or
    float64 DeviceArrays (wcs_pixel_map).  order: 1 bilinear, 0 nearest neighbour.  any_valid: optional
    1-element uint32 DeviceArray, set to 1 iff some output value is not NaN."""
    dev = cube.device
    if isinstance(xs, DeviceArray) and isinstance(ys, DeviceArray):
        if xs.dtype != np.float64 or ys.dtype != np.float64 or xs.shape != ys.shape or len(xs.shape) != 2:
            raise ValueError("xs, ys must be 2-D float64 maps of identical shape")
        d_xs, d_ys = xs, ys
        ny_out, nx_out = xs.shape
    else:
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        if xs.shape != ys.shape or xs.ndim != 2:
            raise ValueError("xs, ys must be 2-D maps of identical shape")
        ny_out, nx_out = xs.shape
        d_xs, d_ys = DeviceArray.from_numpy(xs, dev), DeviceArray.from_numpy(ys, dev)
    if out is None:
        out = DeviceArray((cube.shape[0], ny_out, nx_out), np.float32, dev)
    elif out.shape != (cube.shape[0], ny_out, nx_out) or out.dtype != np.float32 or getattr(out, "_is_view", False):
        raise ValueError("out must be a contiguous float32 (nz, ny_out, nx_out) DeviceArray")
    foot = DeviceArray((ny_out, nx_out), np.uint8, dev) if want_footprint else None
    c, m = _cube_c(cube), _mask_c(mask, cube)
    ws, wsn = workspace(dev, stream, _lib.WS_RESAMPLE_BILINEAR, *cube.shape, ny_out, nx_out)
    _lib.call("spc_resample_bilinear_f32", dev, _sh(stream), C.byref(c), C.byref(m), float(fill),
              ny_out, nx_out, C.c_void_p(d_xs.ptr), C.c_void_p(d_ys.ptr), C.c_void_p(out.ptr), 0, 0,
              C.c_void_p(foot.ptr) if foot is not None else None, int(order),
              C.c_void_p(any_valid.ptr) if any_valid is not None else None, ws, wsn)
    out._plan = (d_xs, d_ys)
    return out, foot


def lerp_plan_folds(lo):
    """a plan resample_bilinear_lerp accepts: monotone either way (a descending one is run reversed)"""
    lo = np.asarray(lo)
    return lerp_plan_is_foldable(lo) or lerp_plan_is_foldable(lo[::-1])


def lerp_plan_is_foldable(lo):
    """True when a spectral_lerp plan can ride in the resampling kernel (resample_bilinear_lerp): its non-negative
    entries ascend and are contiguous (an ascending grid on ascending channels)"""
    lo = np.asarray(lo)
    ok = np.nonzero(lo >= 0)[0]
    if len(ok) == 0:
        return False
    return bool(ok[-1] - ok[0] + 1 == len(ok) and np.all(np.diff(lo[ok]) >= 0))


def resample_bilinear_lerp(cube, xs, ys, lo, t, inv_dx, fill=np.nan, mask=None, stream=None, want_footprint=True, out=None,
                           order=1, any_valid=None):
    """resample_bilinear with the spectral interpolation folded in (spc_resample_bilinear_lerp_f32): output channel j is the
    linear blend (spectral_lerp's plan *lo*, *t*, *inv_dx*) of the RESAMPLED input planes lo[j], lo[j] + 1 - one read of the
    cube, one write of the result, for spectral_interpolate(...).reproject(...) (dask_spectral_cube.py:1342-1353 +
    spectral_cube.py:2700-2732) and for reproject onto a cube header with its own spectral axis.  Channels with lo < 0 are
    NaN planes.  Returns (cube of len(lo) channels, footprint)."""
    dev = cube.device
    lo = np.ascontiguousarray(lo, dtype=np.int32)
    t, inv_dx = np.asarray(t, dtype=np.float64), np.asarray(inv_dx, dtype=np.float64)
    flip = False
    if not lerp_plan_is_foldable(lo):
        # a descending plan (a reversed output grid, or a reversed input axis): the same kernel on the reversed plan, its
        # output planes written from the last one down (a negative plane stride)
        if not lerp_plan_is_foldable(lo[::-1]):
            raise _lib.HipUnsupported("resample_bilinear_lerp: the plan's channels must be monotone and contiguous (run the two passes)")
        lo, t, inv_dx, flip = lo[::-1].copy(), t[::-1].copy(), inv_dx[::-1].copy(), True
    if cube.shape[0] < 2:
        raise _lib.HipUnsupported("resample_bilinear_lerp: at least two input channels")
    if isinstance(xs, DeviceArray) and isinstance(ys, DeviceArray):
        if xs.dtype != np.float64 or ys.dtype != np.float64 or xs.shape != ys.shape or len(xs.shape) != 2:
            raise ValueError("xs, ys must be 2-D float64 maps of identical shape")
        d_xs, d_ys = xs, ys
        ny_out, nx_out = xs.shape
    else:
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        if xs.shape != ys.shape or xs.ndim != 2:
            raise ValueError("xs, ys must be 2-D maps of identical shape")
        ny_out, nx_out = xs.shape
        d_xs, d_ys = DeviceArray.from_numpy(xs, dev), DeviceArray.from_numpy(ys, dev)
    nz_out = len(lo)
    d_lo = DeviceArray.from_numpy(lo, dev)
    d_t = DeviceArray.from_numpy(np.asarray(t, dtype=np.float64), dev)
    d_inv = DeviceArray.from_numpy(np.asarray(inv_dx, dtype=np.float64), dev)
    if out is None:
        out = DeviceArray((nz_out, ny_out, nx_out), np.float32, dev)
    elif out.shape != (nz_out, ny_out, nx_out) or out.dtype != np.float32 or getattr(out, "_is_view", False):
        raise ValueError("out must be a contiguous float32 (nz_out, ny_out, nx_out) DeviceArray")
    foot = DeviceArray((ny_out, nx_out), np.uint8, dev) if want_footprint else None
    c, m = _cube_c(cube), _mask_c(mask, cube)
    ws, wsn = workspace(dev, stream, _lib.WS_RESAMPLE_BILINEAR_LERP, *cube.shape, ny_out, nx_out)
    plane = ny_out * nx_out
    _lib.call("spc_resample_bilinear_lerp_f32", dev, _sh(stream), C.byref(c), C.byref(m), float(fill),
              ny_out, nx_out, C.c_void_p(d_xs.ptr), C.c_void_p(d_ys.ptr), nz_out, C.c_void_p(d_lo.ptr), C.c_void_p(d_t.ptr),
              C.c_void_p(d_inv.ptr), C.c_void_p(out.ptr + ((nz_out - 1) * plane * 4 if flip else 0)), nx_out, -plane if flip else plane,
              C.c_void_p(foot.ptr) if foot is not None else None, int(order),
              C.c_void_p(any_valid.ptr) if any_valid is not None else None, ws, wsn)
    out._plan = (d_xs, d_ys, d_lo, d_t, d_inv)
    return out, foot


def spatial_conv_mfma(cube, kernel2d, mask=None, stream=None, out=None, want_cube=True, want_m0=False, dv=1.0, m0=None):
    """masked separable spatial_smooth with the denominator on the matrix cores (spc_spatial_conv_sep_mfma_f32), optionally
    fused with moment 0 of the smoothed cube under the ORIGINAL mask (the cube is then never written when want_cube is
    False).  Returns (smoothed cube or None, m0 map float64 or None).  HipUnsupported: kernels of more than 29 taps per axis,
    non-separable or negative kernels, mask terms other than the array / isfinite - the caller falls back to spatial_conv
    (+ moments)."""
    k = np.asarray(kernel2d, dtype=np.float64)
    fac = separable_factors(k)
    if fac is None:
        raise _lib.HipUnsupported("spatial_conv_mfma: the kernel is not an outer product")
    ky, kx = (np.ascontiguousarray(f, dtype=np.float64) for f in fac)
    dev = cube.device
    nz, ny, nx = cube.shape
    if want_cube and out is None:
        out = DeviceArray((nz, ny, nx), np.float32, dev)
    if want_m0 and m0 is None:
        m0 = DeviceArray((ny, nx), np.float64, dev)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    ws, wsn = workspace(dev, stream, _lib.WS_SPATIAL_CONV_MFMA, nz, ny, nx)
    _lib.call("spc_spatial_conv_sep_mfma_f32", dev, _sh(stream), C.byref(c), C.byref(m),
              ky.ctypes.data_as(C.POINTER(C.c_double)), len(ky), kx.ctypes.data_as(C.POINTER(C.c_double)), len(kx),
              C.c_void_p(out.ptr) if want_cube else None, 0, 0, float(dv), C.c_void_p(m0.ptr) if want_m0 else None, 0, ws, wsn)
    return (out if want_cube else None), (m0 if want_m0 else None)


def spatial_conv_mfma_moments(cube, kernel2d, d_cen, dv=1.0, m1_add=0.0, mask=None, want=("m0", "m1", "m2"), stream=None,
                              out=None, want_cube=False):
    """masked separable spatial_smooth fused with moments 0 / 1 / 2 of the smoothed cube under the ORIGINAL mask
    (spc_spatial_conv_sep_mfma_moments_f32): the smoothed cube is never written unless want_cube.  *d_cen*: DeviceArray of
    nz doubles (channel coordinates about the reference, as for moments()).  Returns (cube or None, {"m0": ..}) with
    float64 DeviceArray maps.  HipUnsupported as spatial_conv_mfma, and where the split form does not apply."""
    k = np.asarray(kernel2d, dtype=np.float64)
    fac = separable_factors(k)
    if fac is None:
        raise _lib.HipUnsupported("spatial_conv_mfma: the kernel is not an outer product")
    ky, kx = (np.ascontiguousarray(f, dtype=np.float64) for f in fac)
    dev = cube.device
    nz, ny, nx = cube.shape
    if want_cube and out is None:
        out = DeviceArray((nz, ny, nx), np.float32, dev)
    maps = {w: DeviceArray((ny, nx), np.float64, dev) for w in want}
    c, m = _cube_c(cube), _mask_c(mask, cube)
    higher = ("m1" in maps) or ("m2" in maps)
    ws, wsn = workspace(dev, stream, _lib.WS_SPATIAL_CONV_MFMA, nz, ny, nx, 3 if higher else 1)
    ptr = lambda w: C.c_void_p(maps[w].ptr) if w in maps else None
    _lib.call("spc_spatial_conv_sep_mfma_moments_f32", dev, _sh(stream), C.byref(c), C.byref(m),
              ky.ctypes.data_as(C.POINTER(C.c_double)), len(ky), kx.ctypes.data_as(C.POINTER(C.c_double)), len(kx),
              C.c_void_p(out.ptr) if want_cube else None, 0, 0, C.c_void_p(d_cen.ptr), float(dv), float(m1_add),
              ptr("m0"), ptr("m1"), ptr("m2"), 0, ws, wsn)
    return (out if want_cube else None), maps


def map_check(counts=None, expect=0, values=None, stream=None):
    """device-side checks of the algebraic smooth -> moment paths (spc_map_check): bit 0 = a count differs from *expect*, bit 1 = a
    value is not finite.  Returns the flags (an int; reads back four bytes)."""
    ref = counts if counts is not None else values
    n = int(np.prod(ref.shape, dtype=np.int64))
    flags = DeviceArray((1,), np.uint32, ref.device)
    _lib.call("spc_map_check", ref.device, _sh(stream), C.c_void_p(counts.ptr) if counts is not None else None, int(expect),
              C.c_void_p(values.ptr) if values is not None else None, n, C.c_void_p(flags.ptr))
    return int(flags.get(stream)[0])


def resample_spline(cube, xs, ys, order, stream=None, want_footprint=True, out=None, slab_bytes=2 << 30):
    """biquadratic (order 2) / bicubic (order 3) spatial resample of every channel at (xs, ys): scipy's
    map_coordinates(order, mode='constant', cval=nan) on the border-replicated planes, the resampler of
    reproject_interp(order='biquadratic' | 'bicubic') (spectral_cube.py:2667-2676, 2726-2732).  The cube must hold
    finite samples only (the caller checks: scipy's prefilter makes everything NaN otherwise).  The float64 spline
    coefficients of a slab of channels at a time live in a scratch buffer of at most *slab_bytes*."""
    dev = cube.device
    if order not in (2, 3):
        raise ValueError("spline order 2 or 3")
    if isinstance(xs, DeviceArray) and isinstance(ys, DeviceArray):
        d_xs, d_ys = xs, ys
        ny_out, nx_out = xs.shape
    else:
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        ys = np.ascontiguousarray(ys, dtype=np.float64)
        ny_out, nx_out = xs.shape
        d_xs, d_ys = DeviceArray.from_numpy(xs, dev), DeviceArray.from_numpy(ys, dev)
    nz, ny, nx = cube.shape
    if out is None:
        out = DeviceArray((nz, ny_out, nx_out), np.float32, dev)
    foot = DeviceArray((ny_out, nx_out), np.uint8, dev) if want_footprint else None
    per_plane = (ny + 2) * (nx + 2) * 8
    planes = int(max(1, min(nz, slab_bytes // per_plane, 65535)))
    coef = DeviceArray((planes * per_plane,), np.uint8, dev)
    for z0 in range(0, nz, planes):
        z1 = min(nz, z0 + planes)
        c = _cube_c(cube.planes(z0, z1))
        _lib.call("spc_resample_spline_f32", dev, _sh(stream), C.byref(c), int(order), ny_out, nx_out, C.c_void_p(d_xs.ptr),
                  C.c_void_p(d_ys.ptr), C.c_void_p(out.ptr + z0 * ny_out * nx_out * 4), 0, 0,
                  C.c_void_p(foot.ptr) if (foot is not None and z0 == 0) else None, C.c_void_p(coef.ptr), C.c_size_t(coef.nbytes))
    if stream is not None:
        _lib.call("spc_stream_sync", dev, _sh(stream))
    else:
        _lib.call("spc_stream_sync", dev, None)          # `coef` goes back to the pool: the kernels must be done with it
    out._plan = (d_xs, d_ys)
    return out, foot


# ---- statistics ----------------------------------------------
STAT_KEYS = ("count", "min", "max", "sum", "sumsq")
_STAT_DTYPES = {"count": np.int32, "min": np.float32, "max": np.float32, "sum": np.float64, "sumsq": np.float64}


def stats_global(cube, mask=None, stream=None):
    """{npts, min, max, sum, sumsq} of the included samples of the whole cube in ONE pass
    (per-chunk compute_stats + aggregation of statistics(), dask_spectral_cube.py:769-814).
    Returns python floats; synchronises."""
    c, m = _cube_c(cube), _mask_c(mask, cube)
    h = (C.c_double * 5)()
    ws, wsn = workspace(cube.device, stream, _lib.WS_STATS_GLOBAL, *cube.shape)
    _lib.call("spc_stats_global_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), h, ws, wsn)
    return {"npts": h[0], "min": h[1], "max": h[2], "sum": h[3], "sumsq": h[4]}


def stats_planes(cube, mask=None, stream=None):
    """{count, min, max, sum, sumsq} -> float64 arrays of length nz: the statistics of every channel's
    plane in ONE pass (nan-reductions with axis=(1, 2): spectra).  Synchronises."""
    nz = cube.shape[0]
    buf = (C.c_double * (5 * nz))()
    c, m = _cube_c(cube), _mask_c(mask, cube)
    ws, wsn = workspace(cube.device, stream, _lib.WS_STATS_PLANES, *cube.shape)
    _lib.call("spc_stats_planes_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), buf, ws, wsn)
    a = np.frombuffer(buf, dtype=np.float64).reshape(nz, 5)
    return {"count": a[:, 0].copy(), "min": a[:, 1].copy(), "max": a[:, 2].copy(), "sum": a[:, 3].copy(), "sumsq": a[:, 4].copy()}


def stats_axis(cube, axis, mask=None, want=STAT_KEYS, stream=None, out=None):
    """count / min / max / sum / sumsq maps along *axis* in ONE pass (the nan-reductions behind
    sum / mean / std / max / min, dask_spectral_cube.py:641-767).  Returns DeviceArrays
    (*out*: dict of preallocated ones, reused)."""
    if axis not in (0, 1, 2):
        raise ValueError("axis must be 0, 1 or 2")
    shp = tuple(n for i, n in enumerate(cube.shape) if i != axis)
    res = {}
    for k in want:
        a = out.get(k) if out else None
        if a is None:
            a = DeviceArray(shp, _STAT_DTYPES[k], cube.device)
        elif tuple(a.shape) != shp or a.dtype != _STAT_DTYPES[k]:
            raise ValueError("out[%r] must be %s %s" % (k, shp, _STAT_DTYPES[k]))
        res[k] = a
    o = _lib.SpcStatsOutputs()
    for k in want:
        setattr(o, "d_" + k, res[k].ptr)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    _lib.call("spc_stats_axis_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), int(axis), C.byref(o))
    return res


def map_conv2d(dmap, kernel2d, stream=None, out=None):
    """zero-filled, sum-normalised 2-D convolution of one float64 (ny, nx) DeviceArray (the map
    form of spatial_smooth used by the algebraic smooth -> moment path)."""
    if dmap.dtype != np.float64 or len(dmap.shape) != 2:
        raise TypeError("map must be a 2-D float64 DeviceArray")
    k = np.ascontiguousarray(kernel2d, dtype=np.float64)
    if k.ndim != 2:
        raise ValueError("kernel must be 2-D")
    if out is None:
        out = DeviceArray(dmap.shape, np.float64, dmap.device)
    ws, wsn = workspace(dmap.device, stream, _lib.WS_MAP_CONV2D, 1, dmap.shape[0], dmap.shape[1], k.shape[0], k.shape[1])
    _lib.call("spc_map_conv2d_f64", dmap.device, _sh(stream), C.c_void_p(dmap.ptr), dmap.shape[0], dmap.shape[1],
              k.ctypes.data_as(C.POINTER(C.c_double)), k.shape[0], k.shape[1], C.c_void_p(out.ptr), ws, wsn)
    return out


def map_arith(op, a, b, c=None, s=0.0, out=None, stream=None):
    """elementwise arithmetic on float64 maps (spc_map_arith_f64; op one of _lib.MAP_*): the algebra around
    map_conv2d without a trip to the host."""
    for m in (a, b, c):
        if m is not None and (m.dtype != np.float64 or tuple(m.shape) != tuple(a.shape)):
            raise TypeError("maps must be float64 DeviceArrays of one shape")
    if out is None:
        out = DeviceArray(a.shape, np.float64, a.device)
    _lib.call("spc_map_arith_f64", a.device, _sh(stream), int(op), C.c_void_p(a.ptr), C.c_void_p(b.ptr),
              C.c_void_p(c.ptr) if c is not None else None, float(s), C.c_void_p(out.ptr), int(np.prod(a.shape, dtype=np.int64)))
    return out


MAD_TO_STD = 1.482602218505602          # 1 / Phi^-1(3/4), astropy.stats.mad_std


def percentile_axis0(cube, q, mask=None, center=None, scale=1.0, stream=None, out=None):
    """q-th percentile along the spectral axis per spaxel (median: q = 50), numpy 'linear'
    interpolation, NaN / masked samples ignored (dask_spectral_cube.py:657-693); with *center*
    (a (ny, nx) float32 DeviceArray) of |x - center| times *scale* (mad_std, :711-731).
    Selection along y: pass ``cube.swap01()`` (and ``mask.swap01()``); the result is (nz, nx)."""
    if out is None:
        out = DeviceArray(cube.shape[1:], np.float32, cube.device)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    if center is not None and (center.dtype != np.float32 or tuple(center.shape) != tuple(cube.shape[1:])):
        raise TypeError("center must be a float32 (ny, nx) DeviceArray")
    _lib.call("spc_percentile_axis0_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), float(q),
              C.c_void_p(center.ptr) if center is not None else None, float(scale), C.c_void_p(out.ptr))
    return out


def fill_masked(cube, mask=None, fill=np.nan, stream=None, out=None):
    """device copy with excluded voxels replaced by *fill* (MaskBase._filled, masks.py:197-237)."""
    if out is None:
        out = DeviceArray(cube.shape, np.float32, cube.device)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    _lib.call("spc_fill_masked_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), float(fill), C.c_void_p(out.ptr), 0, 0)
    return out


def mask_include(cube, mask=None, nan_excluded=False, stream=None):
    """uint8 (nz, ny, nx) DeviceArray: 1 where *mask* (a MaskSpec evaluated on *cube*'s values) includes
    the voxel (spc_mask_include_u8) - a lazy mask bound to another device-resident cube is lowered
    like this instead of through a host copy of that cube."""
    out = DeviceArray(cube.shape, np.uint8, cube.device)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    _lib.call("spc_mask_include_u8", cube.device, _sh(stream), C.byref(c), C.byref(m), 1 if nan_excluded else 0,
              C.c_void_p(out.ptr))
    return out


def percentile_axis2(cube, q, mask=None, center=None, scale=1.0, stream=None, out=None):
    """q-th percentile along x per (z, y) row, no transposed copy (spc_percentile_axis2_f32); *center*: a (nz, ny)
    float32 DeviceArray.  Raises HipUnsupported for rows of more than 4096 samples."""
    nz, ny, nx = cube.shape
    if out is None:
        out = DeviceArray((nz, ny), np.float32, cube.device)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    if center is not None and (center.dtype != np.float32 or tuple(center.shape) != (nz, ny)):
        raise TypeError("center must be a float32 (nz, ny) DeviceArray")
    _lib.call("spc_percentile_axis2_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), float(q),
              C.c_void_p(center.ptr) if center is not None else None, float(scale), C.c_void_p(out.ptr))
    return out


def percentile_global(cube, q, mask=None, center=None, stream=None):
    """q-th percentile of all included samples of the cube (np.nanpercentile(..., axis=None));
    with *center* of |x - center|.  Returns a python float (NaN when nothing is included)."""
    out = C.c_double(0.0)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    ws, wsn = workspace(cube.device, stream, _lib.WS_PERCENTILE_GLOBAL, *cube.shape)
    _lib.call("spc_percentile_global_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), float(q),
              0 if center is None else 1, 0.0 if center is None else float(center), C.byref(out), ws, wsn)
    return out.value


def key_histogram(cube, prefix, pmask, shift, mask=None, center=None, stream=None):
    """One pass of the whole-cube selection on THIS rank's part of a cube (spc_key_histogram_f32): the 256 counters
    of the key byte at bit *shift* among the included samples whose key agrees with *prefix* on *pmask*.
    Returns a uint64 ndarray of 256."""
    h = np.zeros(256, np.uint64)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    ws, wsn = workspace(cube.device, stream, _lib.WS_PERCENTILE_GLOBAL, *cube.shape)
    _lib.call("spc_key_histogram_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), int(prefix), int(pmask), int(shift),
              0 if center is None else 1, 0.0 if center is None else float(center),
              h.ctypes.data_as(C.POINTER(C.c_uint64)), None, ws, wsn)
    return h


def key_next(cube, prefix, mask=None, center=None, stream=None):
    """smallest sample key above *prefix* on this rank's part (0xffffffff when there is none)."""
    nxt = C.c_uint32(0)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    ws, wsn = workspace(cube.device, stream, _lib.WS_PERCENTILE_GLOBAL, *cube.shape)
    _lib.call("spc_key_histogram_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), int(prefix), 0xffffffff, 0,
              0 if center is None else 1, 0.0 if center is None else float(center), None, C.byref(nxt), ws, wsn)
    return nxt.value


def key_to_float(key):
    """the float32 sample an order-preserving key stands for (spc_key_to_f32)."""
    return float(_lib.load().spc_key_to_f32(C.c_uint32(int(key))))


def fill_masked_transposed(cube, mask=None, fill=np.nan, stream=None):
    """(nz, nx, ny) device copy: excluded voxels replaced by *fill*, spatial axes exchanged
    (spc_fill_masked_transpose_f32) - rays along x become rays along y."""
    nz, ny, nx = cube.shape
    out = DeviceArray((nz, nx, ny), np.float32, cube.device)
    c, m = _cube_c(cube), _mask_c(mask, cube)
    _lib.call("spc_fill_masked_transpose_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), float(fill), C.c_void_p(out.ptr))
    return out


def sigma_clip_axis0(cube, sigma=3.0, sigma_lower=None, sigma_upper=None, maxiters=5, cenfunc="median",
                     stdfunc="std", mask=None, stream=None):
    """astropy.stats.sigma_clip(axis=0, masked=False, copy=True) on the device
    (DaskSpectralCubeMixin.sigma_clip_spectrally, dask_spectral_cube.py:851-878): returns a new
    float32 DeviceArray with masked and clipped samples set to NaN.  Every iteration is one
    selection (median), one statistics pass (std) and one clip pass; it stops when a pass clips
    nothing or after *maxiters* (None: until convergence)."""
    lo_s = sigma if sigma_lower is None else sigma_lower
    hi_s = sigma if sigma_upper is None else sigma_upper
    if cenfunc not in ("median", "mean") or stdfunc not in ("std", "mad_std"):
        raise NotImplementedError("cenfunc must be 'median' or 'mean', stdfunc 'std' or 'mad_std' on the device path")
    # the whole loop in one kernel, rays resident in registers (<= 4096 channels)
    if cube.shape[0] <= 4096 and os.environ.get("SPC_SIGMA_CLIP_FUSED", "1") != "0":
        out = DeviceArray(cube.shape, np.float32, cube.device)
        c, m = _cube_c(cube), _mask_c(mask, cube)
        ws, wsn = workspace(cube.device, stream, _lib.WS_SIGMA_CLIP, *cube.shape)
        try:
            _lib.call("spc_sigma_clip_axis0_f32", cube.device, _sh(stream), C.byref(c), C.byref(m), float(lo_s), float(hi_s),
                      -1 if maxiters is None else int(maxiters), 1 if cenfunc == "mean" else 0, 1 if stdfunc == "mad_std" else 0,
                      C.c_void_p(out.ptr), ws, wsn)
            return out
        except _lib.HipUnsupported:
            del out                    # planes beyond what the register-resident kernel addresses: the loop of kernels below
    work = fill_masked(cube, mask, np.nan, stream)
    nz, ny, nx = work.shape
    dev = work.device
    d_lo, d_hi = DeviceArray((ny, nx), np.float32, dev), DeviceArray((ny, nx), np.float32, dev)
    need_stats = cenfunc == "mean" or stdfunc == "std"
    st = None
    it = 0
    while maxiters is None or it < maxiters:
        it += 1
        if need_stats:
            st = stats_axis(work, 0, want=("count", "sum", "sumsq"), stream=stream, out=st)
        med = percentile_axis0(work, 50.0, stream=stream) if (cenfunc == "median" or stdfunc == "mad_std") else None
        spread = percentile_axis0(work, 50.0, center=med, scale=MAD_TO_STD, stream=stream) if stdfunc == "mad_std" else None
        center = med if cenfunc == "median" else None
        if center is None and spread is not None:          # mean centre with a mad_std spread: the mean map itself
            n = st["count"].get().astype(np.float64)
            with np.errstate(invalid="ignore", divide="ignore"):
                center = DeviceArray.from_numpy(np.where(n > 0, st["sum"].get() / n, np.nan).astype(np.float32), dev)
        ptr = lambda a: C.c_void_p(a.ptr) if a is not None else None  # noqa: E731
        use_stats = need_stats and (center is None or spread is None)
        _lib.call("spc_clip_bounds_f32", dev, _sh(stream), ny * nx,
                  ptr(st["count"]) if use_stats else None, ptr(st["sum"]) if use_stats else None,
                  ptr(st["sumsq"]) if use_stats else None, ptr(center), ptr(spread), float(lo_s), float(hi_s),
                  C.c_void_p(d_lo.ptr), C.c_void_p(d_hi.ptr))
        nch = C.c_uint64(0)
        ws, wsn = workspace(dev, stream, _lib.WS_CLIP_OUTSIDE, nz, ny, nx)
        _lib.call("spc_clip_outside_f32", dev, _sh(stream), C.c_void_p(work.ptr), nz, ny, nx,
                  C.c_void_p(d_lo.ptr), C.c_void_p(d_hi.ptr), C.byref(nch), ws, wsn)
        if nch.value == 0:
            break
    return work


def scale_inplace(arr, factor, stream=None):
    """arr *= factor on the device (contiguous float32 DeviceArray)."""
    if arr.dtype != np.float32 or getattr(arr, "_is_view", False):
        raise TypeError("scale_inplace needs a contiguous float32 DeviceArray")
    _lib.call("spc_scale_f32", arr.device, _sh(stream), C.c_void_p(arr.ptr), int(np.prod(arr.shape, dtype=np.int64)), float(factor))
    return arr
