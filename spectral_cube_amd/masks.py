"""Host-side mask algebra mirroring spectral_cube/masks.py, plus the lowering
of a mask tree to the device-evaluable :class:`~spectral_cube_amd.ops.MaskSpec`.

Semantics restated from the reference:
  * ``include`` / ``exclude`` (masks.py:105-138);
  * ``_filled``: ``where(include, data.astype(result_type(dtype, 0.0)), fill)``
    (masks.py:197-237);
  * ``&`` ``|`` ``^`` ``~`` composition (masks.py:239-250, 399-455);
  * BooleanArrayMask (masks.py:457-584), LazyMask (:586-668),
    LazyComparisonMask (:670-758), FunctionMask (:760-803).
WCS consistency checks between mask and cube are not reproduced (metadata
bookkeeping, out of scope).
"""
import operator

import threading

import numpy as np

from . import _lib

_substitute = threading.local()     # .host = (data identity, host array) while lower_mask() evaluates a host term


class MaskBase:
    def include(self, data=None, view=()):
        """True where the voxel takes part in computations."""
        return self._include(data=data, view=view)

    def exclude(self, data=None, view=()):
        return np.logical_not(self._include(data=data, view=view))

    def _include(self, data=None, view=()):
        raise NotImplementedError

    def any(self):
        return np.any(self.include())

    def _filled(self, data, fill=np.nan, view=()):
        dt = np.result_type(data.dtype, 0.0)
        sliced = np.array(data[view], dtype=dt, copy=True)
        sliced[self.exclude(data=data, view=view)] = fill
        return sliced

    def _flattened(self, data, view=()):
        return data[view][self.include(data=data, view=view)]

    def __and__(self, other):
        return CompositeMask(self, other, operation="and")

    def __or__(self, other):
        return CompositeMask(self, other, operation="or")

    def __xor__(self, other):
        return CompositeMask(self, other, operation="xor")

    def __invert__(self):
        return InvertedMask(self)

    # ---- device lowering --------------------------------------------------
    def _device_terms(self, data):
        """Return (flags, thr_lo, thr_hi, host_bool_array_or_None) when this
        mask can be expressed as an AND of device terms evaluated on *data*
        (the array the kernel will read); otherwise None."""
        return None


class InvertedMask(MaskBase):
    def __init__(self, mask):
        self._mask = mask

    def _include(self, data=None, view=()):
        return np.logical_not(self._mask.include(data=data, view=view))

    def __invert__(self):
        return self._mask


class CompositeMask(MaskBase):
    def __init__(self, mask1, mask2, operation="and"):
        if operation not in ("and", "or", "xor"):
            raise ValueError("Operation '{0}' not supported".format(operation))
        self._mask1, self._mask2, self._operation = mask1, mask2, operation

    def _include(self, data=None, view=()):
        a = self._mask1._include(data=data, view=view)
        b = self._mask2._include(data=data, view=view)
        if self._operation == "and":
            return np.bitwise_and(a, b)
        if self._operation == "or":
            return np.bitwise_or(a, b)
        return np.bitwise_xor(a, b)

    def _device_terms(self, data):
        if self._operation != "and":
            return None
        a = self._mask1._device_terms(data)
        b = self._mask2._device_terms(data)
        if a is None or b is None:
            return None
        return _and_terms(a, b)


class BooleanArrayMask(MaskBase):
    def __init__(self, mask, wcs=None, shape=None, include=True):
        mask = np.asarray(mask)
        if mask.dtype != bool:
            mask = mask.astype(bool)
        self._mask = mask
        self._wcs = wcs
        self._include_flag = include
        if shape is not None:
            np.broadcast_shapes(mask.shape, tuple(shape))   # raises like the reference on mismatch
        self._shape = tuple(shape) if shape is not None else mask.shape

    @property
    def shape(self):
        return self._shape

    def _include(self, data=None, view=()):
        m = np.broadcast_to(self._mask, self._shape)[view]
        return m if self._include_flag else np.logical_not(m)

    def _device_terms(self, data):
        m = np.broadcast_to(self._mask, self._shape)
        if not self._include_flag:
            m = np.logical_not(m)
        return (0, -np.inf, np.inf, m)


class LazyMask(MaskBase):
    """mask = function(data), evaluated lazily on the data it was created for
    (masks.py:586-668).  ``np.isfinite`` lowers to SPC_MASK_FINITE."""

    def __init__(self, function, cube=None, data=None):
        self._function = function
        if cube is not None:
            self._data_ref = cube
        elif data is not None:
            self._data_ref = _Holder(data)
        else:
            raise ValueError("Either a cube or (data & wcs) is required.")

    def _own_data(self):
        # lower_mask() evaluating on behalf of a view of the SAME data (cube._WideView: the float64 samples of a wide
        # FITS image) hands its host array over here, so that the predicate is decided on the samples the kernel will
        # read and not on the cube's float32-narrowed copy
        sub = getattr(_substitute, "host", None)
        if sub is not None and getattr(self._data_ref, "_data_id", self._data_ref) is sub[0]:
            return sub[1]
        return self._data_ref._host_data()

    def _include(self, data=None, view=()):
        return self._function(self._own_data()[view])

    def _device_terms(self, data):
        if self._function is np.isfinite and self._data_ref._is_same_data(data):
            return (_lib.MASK_FINITE, -np.inf, np.inf, None)
        return None


class NotNaNMask(LazyMask):
    """``~np.isnan(data)`` of the cube's own data - the mask spectral_interpolate
    and reproject attach (dask_spectral_cube.py:1364; spectral_cube.py:2741-2746
    with NaN outside the footprint).  Redundant on the device: NaN samples never
    take part in a reduction."""

    def __init__(self, cube):
        super().__init__(lambda x: ~np.isnan(x), cube=cube)

    def _device_terms(self, data):
        if self._data_ref._is_same_data(data):
            return (0, -np.inf, np.inf, None)
        return None


_CMP = {operator.gt: _lib.MASK_GT, operator.ge: _lib.MASK_GE,
        operator.lt: _lib.MASK_LT, operator.le: _lib.MASK_LE}


class LazyComparisonMask(LazyMask):
    """``cube > value`` etc. (masks.py:670-758)."""

    def __init__(self, function, comparison_value, cube=None, data=None):
        self._cmp = function
        self._value = comparison_value
        super().__init__(lambda x: function(x, comparison_value), cube=cube, data=data)

    def _device_terms(self, data):
        if self._cmp in _CMP and np.isscalar(self._value) and self._data_ref._is_same_data(data):
            v = float(self._value)
            # The device compares float32 data with a float32 threshold.  numpy does
            # the same for "weak" python scalars (NEP 50); typed float64 scalars are
            # compared in float64, so only lower those when exactly representable.
            weak = type(self._value) in (float, int)
            if v != v:
                return None          # a NaN threshold includes nothing: left to the host form (all False)
            if getattr(data, "_wide", False):
                # the float64 kernels (cube._mask_spec64): float64 samples against a float64 threshold, as numpy compares them
                flag = _CMP[self._cmp]
                return (flag, v, np.inf, None) if flag in (_lib.MASK_GT, _lib.MASK_GE) else (flag, -np.inf, v, None)
            if weak or float(np.float32(v)) == v or not np.isfinite(v):
                v = float(np.float32(v))          # +-inf stay +-inf: `cube > -inf` is a real comparison
                flag = _CMP[self._cmp]
                if flag in (_lib.MASK_GT, _lib.MASK_GE):
                    return (flag, v, np.inf, None)
                return (flag, -np.inf, v, None)
        return None


class FunctionMask(MaskBase):
    """mask = function(data[view]) (masks.py:760-803)."""

    def __init__(self, function):
        self._function = function

    def _include(self, data=None, view=()):
        out = self._function(data[view])
        if out.shape != data[view].shape:
            raise ValueError("Function did not return mask with correct shape - expected "
                             "{0}, got {1}".format(data[view].shape, out.shape))
        return out


class _Holder:
    def __init__(self, data):
        self._d = data

    def _host_data(self):
        return self._d

    def _is_same_data(self, data):
        return data is self._d


def _and_terms(a, b):
    fa, la, ha, ma = a
    fb, lb, hb, mb = b
    flags = (fa | fb) & _lib.MASK_FINITE
    # lower bound: the stricter of the two (GT beats GE at equal thresholds)
    lo, lof = -np.inf, 0
    for f, l in ((fa & (_lib.MASK_GT | _lib.MASK_GE), la), (fb & (_lib.MASK_GT | _lib.MASK_GE), lb)):
        if f and (not lof or l > lo or (l == lo and f == _lib.MASK_GT)):
            lo, lof = l, f
    hi, hif = np.inf, 0
    for f, h in ((fa & (_lib.MASK_LT | _lib.MASK_LE), ha), (fb & (_lib.MASK_LT | _lib.MASK_LE), hb)):
        if f and (not hif or h < hi or (h == hi and f == _lib.MASK_LT)):
            hi, hif = h, f
    if ma is None:
        m = mb
    elif mb is None:
        m = ma
    else:
        m = np.logical_and(ma, mb)
    return (flags | lof | hif, lo, hi, m)


def lower_mask(mask, data, shape):
    """Lower *mask* for kernels that read *data* (host ndarray identity used to
    decide whether lazy predicates may run on the device).

    Returns (flags, thr_lo, thr_hi, host_uint8_array_or_None).  Anything that
    cannot be expressed as an AND of device terms is materialised on the host
    into the array term - results are identical, only the traffic differs.
    """
    if mask is None:
        return (0, 0.0, 0.0, None)
    terms = mask._device_terms(data)
    if terms is None:
        # host form: FunctionMask and friends index the voxel ARRAY (masks.py:760-803), not the cube object
        host = data._host_data() if hasattr(data, "_host_data") else data
        ident = getattr(data, "_data_id", None)
        _substitute.host = (ident, host) if ident is not None else None
        try:
            inc = np.asarray(mask.include(data=host))
        finally:
            _substitute.host = None
        terms = (0, -np.inf, np.inf, np.broadcast_to(inc, shape))
    flags, lo, hi, m = terms
    arr = None
    if m is not None:
        flags |= _lib.MASK_ARRAY
        arr = np.ascontiguousarray(np.broadcast_to(m, shape)).view(np.uint8)
    # a bound that is not in force is passed as 0; one that is keeps its value, infinities included
    lo = float(lo) if flags & (_lib.MASK_GT | _lib.MASK_GE) else 0.0
    hi = float(hi) if flags & (_lib.MASK_LT | _lib.MASK_LE) else 0.0
    return (flags, lo, hi, arr)


def contains(mask, cls):
    """True when a term of the mask tree is an instance of *cls*"""
    if isinstance(mask, cls):
        return True
    if isinstance(mask, CompositeMask):
        return contains(mask._mask1, cls) or contains(mask._mask2, cls)
    if isinstance(mask, InvertedMask):
        return contains(mask._mask, cls)
    return False


def foreign_owner(mask):
    """The ONE other cube whose data every lazy term of *mask* is bound to (e.g. the parent of a
    smoothed cube, which keeps the parent's mask object: dask_spectral_cube.py:836-840), or None
    when there is no such single cube.  cube._mask_spec() then evaluates the mask on that cube's
    device data instead of copying it to the host."""
    owners = []

    def walk(m):
        if isinstance(m, CompositeMask):
            walk(m._mask1); walk(m._mask2)
        elif isinstance(m, InvertedMask):
            walk(m._mask)
        elif isinstance(m, LazyMask):
            owners.append(m._data_ref)
        elif isinstance(m, FunctionMask):
            owners.append(None)
    walk(mask)
    first = owners[0] if owners else None
    if first is None or isinstance(first, _Holder) or any(o is not first for o in owners):
        return None
    return first
