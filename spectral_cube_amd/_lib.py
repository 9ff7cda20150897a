"""ctypes binding of libspcube_hip.so (C ABI declared in include/spcube_hip.h).

The product path has NO CPU fallback: if the shared library cannot be loaded
every entry point raises :class:`HipLibraryError`.  Loading the library does
not need a GPU (symbols can be inspected on a CPU-only box); launching does.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libspcube_hip.so"
LIB_PATH = os.environ.get("SPC_HIP_LIBRARY", os.path.join(_HERE, LIB_NAME))

SPC_OK = 0
SPC_ERR_INVALID = -1
SPC_ERR_HIP = -2
SPC_ERR_UNSUPPORTED = -3
SPC_ERR_NOMEM = -4
SPC_ERR_COMM = -5

MASK_NONE, MASK_ARRAY, MASK_FINITE = 0, 1, 2
MASK_GT, MASK_GE, MASK_LT, MASK_LE = 4, 8, 16, 32
COMM_ID_BYTES = 128
ABI_VERSION = 8
MAP_MUL, MAP_SECOND_MOMENT_SUM, MAP_DIV_ADD, MAP_DIV_SUB_SQ = 0, 1, 2, 3
# spc_ws_kind
(WS_MOMENTS, WS_SPECTRAL_CONV, WS_SPECTRAL_CONV_MOMENTS, WS_SPATIAL_CONV_SEP, WS_SPATIAL_CONV2D, WS_RESAMPLE_BILINEAR,
 WS_STATS_GLOBAL, WS_STATS_PLANES, WS_MAP_CONV2D, WS_CLIP_OUTSIDE, WS_PERCENTILE_GLOBAL, WS_SPATIAL_CONV_MFMA, WS_SIGMA_CLIP,
 WS_RESAMPLE_BILINEAR_LERP, WS_STATS_GLOBAL_F64, WS_SPECTRAL_CONV_F64, WS_SPATIAL_CONV_F64) = range(17)


class HipLibraryError(RuntimeError):
    """libspcube_hip.so is missing/unloadable, or a HIP call failed."""


class HipInvalidArgument(ValueError):
    pass


class HipUnsupported(NotImplementedError):
    pass


class SpcMask(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("thr_lo", C.c_float), ("thr_hi", C.c_float),
                ("d_array", C.c_void_p), ("row_stride", C.c_int64),
                ("plane_stride", C.c_int64)]


class SpcCube(C.Structure):
    _fields_ = [("d_data", C.c_void_p), ("nz", C.c_int64), ("ny", C.c_int64),
                ("nx", C.c_int64), ("row_stride", C.c_int64),
                ("plane_stride", C.c_int64)]


class SpcMomentOutputs(C.Structure):
    _fields_ = [("d_m0", C.c_void_p), ("d_m1", C.c_void_p), ("d_m2", C.c_void_p),
                ("d_mu", C.c_void_p), ("d_s0", C.c_void_p),
                ("d_argmax", C.c_void_p), ("d_argmin", C.c_void_p),
                ("d_vmax", C.c_void_p), ("d_vmin", C.c_void_p),
                ("d_nvalid", C.c_void_p), ("out_row_stride", C.c_int64)]


class SpcMask64(C.Structure):
    """spc_mask_f64: the mask of a float64 cube (thresholds compared in float64)"""
    _fields_ = [("flags", C.c_uint32), ("thr_lo", C.c_double), ("thr_hi", C.c_double),
                ("d_array", C.c_void_p), ("row_stride", C.c_int64), ("plane_stride", C.c_int64)]


class SpcMomentOutputs64(C.Structure):
    """spc_moment_outputs_f64 (no d_m2: moment 2 of a float64 cube is the second pass, spc_moment_order_f64)"""
    _fields_ = [("d_m0", C.c_void_p), ("d_m1", C.c_void_p), ("d_mu", C.c_void_p), ("d_s0", C.c_void_p),
                ("d_argmax", C.c_void_p), ("d_argmin", C.c_void_p), ("d_vmax", C.c_void_p), ("d_vmin", C.c_void_p),
                ("d_nvalid", C.c_void_p), ("out_row_stride", C.c_int64)]


class SpcCelestialWcs(C.Structure):
    _fields_ = [("proj", C.c_int32), ("sip_order", C.c_int32), ("crpix", C.c_double * 2), ("lin", C.c_double * 4),
                ("lin_inv", C.c_double * 4), ("alpha_p", C.c_double), ("delta_p", C.c_double), ("phi_p", C.c_double),
                ("pv1", C.c_double), ("plane0", C.c_double * 2), ("sip_a", C.c_double * 55), ("sip_b", C.c_double * 55)]


class SpcStatsOutputs(C.Structure):
    _fields_ = [("d_count", C.c_void_p), ("d_min", C.c_void_p), ("d_max", C.c_void_p),
                ("d_sum", C.c_void_p), ("d_sumsq", C.c_void_p)]


class SpcDeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64),
                ("compute_units", C.c_int), ("wavefront_size", C.c_int),
                ("total_mem", C.c_int64), ("free_mem", C.c_int64),
                ("clock_khz", C.c_int)]


_vp, _i, _i64, _sz, _d, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_double, C.c_float
_P = C.POINTER

# name -> (restype, argtypes); mirrors include/spcube_hip.h one to one
SIGNATURES = {
    "spc_abi_version": (_i, []),
    "spc_last_error": (C.c_char_p, []),
    "spc_device_count": (_i, [_P(_i)]),
    "spc_get_device_info": (_i, [_i, _P(SpcDeviceInfo)]),
    "spc_malloc": (_i, [_i, _sz, _P(_vp)]),
    "spc_free": (_i, [_i, _vp]),
    "spc_argextrema_axis_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _i, _vp, _vp]),
    "spc_fill_masked_transpose_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _f, _vp]),
    "spc_percentile_global_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), C.c_double, _i, _f, _P(C.c_double), _vp, _sz]),
    "spc_key_histogram_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), C.c_uint32, C.c_uint32, _i, _i, _f,
                                    _P(C.c_uint64), _P(C.c_uint32), _vp, _sz]),
    "spc_key_to_f32": (_f, [C.c_uint32]),
    "spc_clip_bounds_f32": (_i, [_i, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _d, _d, _vp, _vp]),
    "spc_wcs_pixel_map_f64": (_i, [_i, _vp, _P(SpcCelestialWcs), _P(SpcCelestialWcs), _P(C.c_double), _i64, _i64, _vp, _vp]),
    "spc_spatial_conv_sep_mfma_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _P(C.c_double), _i, _P(C.c_double), _i, _vp, _i64, _i64,
                                           _d, _vp, _i64, _vp, _sz]),
    "spc_spatial_conv_sep_mfma_moments_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _P(C.c_double), _i, _P(C.c_double), _i, _vp, _i64,
                                                   _i64, _vp, _d, _d, _vp, _vp, _vp, _i64, _vp, _sz]),
    "spc_map_check": (_i, [_i, _vp, _vp, C.c_int32, _vp, _i64, _vp]),
    "spc_resample_spline_f32": (_i, [_i, _vp, _P(SpcCube), _i, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _sz]),
    "spc_stats_planes_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _P(C.c_double), _vp, _sz]),
    "spc_pool_trim": (_i, [_i]),
    "spc_pool_stats": (_i, [_i, _P(C.c_int64), _P(C.c_int64)]),
    "spc_host_alloc": (_i, [_sz, _P(_vp)]),
    "spc_host_free": (_i, [_vp]),
    "spc_memcpy_h2d": (_i, [_i, _vp, _vp, _sz, _vp]),
    "spc_memcpy_d2h": (_i, [_i, _vp, _vp, _sz, _vp]),
    "spc_memcpy_d2d": (_i, [_i, _vp, _vp, _sz, _vp]),
    "spc_memcpy3d_h2d": (_i, [_i, _vp, _sz, _sz, _vp, _sz, _sz, _sz, _sz, _sz, _vp]),
    "spc_memcpy3d_d2d": (_i, [_i, _vp, _sz, _sz, _vp, _sz, _sz, _sz, _sz, _sz, _vp]),
    "spc_memset": (_i, [_i, _vp, _i, _sz, _vp]),
    "spc_stream_create": (_i, [_i, _P(_vp)]),
    "spc_stream_destroy": (_i, [_i, _vp]),
    "spc_stream_sync": (_i, [_i, _vp]),
    "spc_device_sync": (_i, [_i]),
    "spc_event_create": (_i, [_i, _P(_vp)]),
    "spc_event_destroy": (_i, [_i, _vp]),
    "spc_event_record": (_i, [_i, _vp, _vp]),
    "spc_event_sync": (_i, [_i, _vp]),
    "spc_stream_wait_event": (_i, [_i, _vp, _vp]),
    "spc_event_elapsed_ms": (_i, [_i, _vp, _vp, _P(_f)]),
    "spc_moments_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "spc_workspace_bytes": (_sz, [_i, _i64, _i64, _i64, _i64, _i64]),
    "spc_moments_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _vp, _d, _d,
                             _P(SpcMomentOutputs), _vp, _sz]),
    "spc_moment_order_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _vp, _i, _vp, _vp, _vp, _i64]),
    # (spc_cube_f64 has the layout of spc_cube_f32: a pointer and five int64)
    "spc_moments_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _vp, _d, _d, _P(SpcMomentOutputs64)]),
    "spc_moment_order_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _vp, _i, _vp, _vp, _vp, _i64]),
    "spc_stats_global_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _P(_d), _vp, _sz]),
    "spc_stats_axis_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _i, _P(SpcStatsOutputs)]),
    "spc_spectral_conv_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _P(_d), _i, _vp, _i64, _i64, _vp, _sz]),
    "spc_spatial_conv_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _P(_d), _i, _P(_d), _i, _i, _vp, _i64, _i64, _vp, _sz]),
    "spc_spectral_lerp_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _i64, _vp, _vp, _vp, _d, _vp, _i64, _i64]),
    "spc_resample_bilinear_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _d, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _i, _vp]),
    "spc_scale_f64": (_i, [_i, _vp, _vp, _i64, _d]),
    "spc_percentile_axis0_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _d, _vp, _d, _vp]),
    "spc_sigma_clip_axis0_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _d, _d, _i, _i, _i, _vp]),
    "spc_narrow_f64_to_f32": (_i, [_i, _vp, _P(SpcCube), _vp, _i64, _i64]),
    "spc_mask_include_f64": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask64), _i, _vp]),
    "spc_moments_spatial_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _i, _vp, _d, _vp, _vp, _vp]),
    "spc_moment_order_spatial_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _i, _vp, _i, _vp, _vp]),
    "spc_spectral_conv_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _P(_d), _i, _vp, _i64, _i64, _vp, _sz]),
    "spc_spectral_conv_moments_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _P(_d), _i, _vp, _P(_d), _d, _d,
                                           _P(SpcMomentOutputs), _vp, _sz]),
    "spc_spatial_conv_sep_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _P(_d), _i, _P(_d), _i,
                                      _vp, _i64, _i64, _vp, _sz]),
    "spc_spatial_conv2d_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _P(_d), _i, _i, _vp, _i64, _i64, _vp, _sz]),
    "spc_spectral_lerp_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _i64, _vp, _vp, _vp, _f,
                                   _vp, _i64, _i64]),
    "spc_resample_bilinear_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _f, _i64, _i64, _vp, _vp,
                                       _vp, _i64, _i64, _vp, _i, _vp, _vp, _sz]),
    "spc_resample_bilinear_lerp_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _f, _i64, _i64, _vp, _vp, _i64, _vp, _vp, _vp,
                                            _vp, _i64, _i64, _vp, _i, _vp, _vp, _sz]),
    "spc_percentile_axis0_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _d, _vp, _f, _vp]),
    "spc_percentile_axis2_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _d, _vp, _f, _vp]),
    "spc_mask_include_u8": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _i, _vp]),
    "spc_fill_masked_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _f, _vp, _i64, _i64]),
    "spc_sigma_clip_axis0_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _d, _d, _i, _i, _i, _vp, _vp, _sz]),
    "spc_clip_outside_f32": (_i, [_i, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _P(C.c_uint64), _vp, _sz]),
    "spc_map_conv2d_f64": (_i, [_i, _vp, _vp, _i64, _i64, _P(_d), _i, _i, _vp, _vp, _sz]),
    "spc_map_arith_f64": (_i, [_i, _vp, _i, _vp, _vp, _vp, _d, _vp, _i64]),
    "spc_scale_f32": (_i, [_i, _vp, _vp, _i64, _d]),
    "spc_fits_to_f32": (_i, [_i, _vp, _vp, _i, _d, _d, _i, _i64, _i64, _vp]),
    "spc_fits_to_f64": (_i, [_i, _vp, _vp, _i, _d, _d, _i, _i64, _i64, _vp]),
    "spc_stats_global_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _P(_d), _vp, _sz]),
    "spc_stats_axis_f32": (_i, [_i, _vp, _P(SpcCube), _P(SpcMask), _i, _P(SpcStatsOutputs)]),
    "spc_comm_unique_id": (_i, [_P(C.c_uint8)]),
    "spc_comm_init": (_i, [_i, _P(C.c_uint8), _i, _i, _P(_vp)]),
    "spc_comm_destroy": (_i, [_vp]),
    "spc_allgather_rows": (_i, [_vp, _vp, _vp, _vp, _sz]),
    "spc_allgather_rows_batch": (_i, [_vp, _vp, _i, _P(_vp), _P(_vp), _P(_sz)]),
}

_lib = None
_lock = threading.Lock()
_load_error = None


def load():
    """Load the shared library once and declare every prototype."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            _load_error = ("%s not found - build it with `python -c 'import __graft_entry__ as g; "
                           "g.build()'` or `make -C spectral_cube_amd/csrc`" % LIB_PATH)
            raise HipLibraryError(_load_error)
        try:
            lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        except OSError as exc:  # missing ROCm runtime etc.
            _load_error = "cannot load %s: %s" % (LIB_PATH, exc)
            raise HipLibraryError(_load_error)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError = ABI mismatch, keep it loud
            fn.restype = res
            fn.argtypes = args
        if lib.spc_abi_version() != ABI_VERSION:
            raise HipLibraryError("ABI version mismatch: %d" % lib.spc_abi_version())
        _lib = lib
        return _lib


def last_error():
    msg = load().spc_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Translate a spc_status into the exception the reference would raise."""
    if rc == SPC_OK:
        return
    msg = last_error()
    if rc == SPC_ERR_INVALID:
        raise HipInvalidArgument(msg)
    if rc == SPC_ERR_UNSUPPORTED:
        raise HipUnsupported(msg)
    if rc == SPC_ERR_NOMEM:
        raise MemoryError(msg)
    raise HipLibraryError("status %d: %s" % (rc, msg))


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args))


def device_count():
    n = C.c_int(0)
    call("spc_device_count", C.byref(n))
    return n.value


def require_gpu():
    if device_count() < 1:
        raise HipLibraryError("no HIP device visible: the spectral_cube_amd product path has no CPU fallback")
