/*
 * spcube_hip.h - C ABI of libspcube_hip.so, the MI355X (gfx950) engine behind
 * spectral-cube's dense hot path.
 *
 * The reference (radio-astro-tools/spectral-cube) is pure Python and has no
 * FFI; the operator seams this library is bound at are listed per entry point
 * (paths relative to the reference tree).  INTEGRATION.md shows the ctypes
 * binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, a negative spc_status otherwise;
 *     spc_last_error() gives a thread-local message; nothing throws;
 *   - the caller owns every buffer, scratch included; "d_" pointers are device (HBM)
 *     addresses obtained from spc_malloc (or any hipMalloc), "h_" pointers are host.
 *     No compute entry point allocates, frees or drains the device: a call that needs
 *     scratch takes (d_workspace, workspace_bytes) - spc_workspace_bytes() gives the
 *     size - and queues everything on the caller's stream.  The host is only made to
 *     wait (for THAT stream) where a result comes back to it: the entry points with
 *     an "h_" output say so;
 *   - cubes are C-contiguous (nz, ny, nx) float32, spectral axis first, x
 *     fastest; row/plane strides are given in ELEMENTS so that a (y) row strip
 *     of a larger cube can be processed in place;
 *   - every launch takes a device index and a stream handle (hipStream_t as
 *     void*, NULL = default stream); there is no global mutable state, so dask
 *     `threads` workers and one-process-per-GPU drivers may call concurrently;
 *   - no torch / numpy types cross this boundary.
 */
#ifndef SPCUBE_HIP_H
#define SPCUBE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPC_ABI_VERSION 8

typedef enum {
    SPC_OK = 0,
    SPC_ERR_INVALID = -1,     /* bad argument (shape, pointer, flag) */
    SPC_ERR_HIP = -2,         /* HIP runtime error, see spc_last_error() */
    SPC_ERR_UNSUPPORTED = -3, /* valid request this build cannot serve */
    SPC_ERR_NOMEM = -4,
    SPC_ERR_COMM = -5         /* RCCL error */
} spc_status;

/* ---- mask specification ------------------------------------------------
 * Restates MaskBase.include / _filled (spectral_cube/masks.py:105-116,
 * 197-237) for the mask kinds that can be evaluated on the fly:
 *   BooleanArrayMask (masks.py:457-584)  -> SPC_MASK_ARRAY (uint8, !=0 = include)
 *   LazyMask(np.isfinite) (io/fits.py:214) -> SPC_MASK_FINITE
 *   LazyComparisonMask cube > x etc. (masks.py:670-758) -> SPC_MASK_GT/GE/LT/LE
 * Flags are AND-ed (CompositeMask 'and', masks.py:425-435).  Other
 * compositions are materialised to a uint8 array by the host.
 * A voxel is "valid" when it is included AND its value is not NaN (NaN-filled
 * data fed to nansum, dask_spectral_cube.py:1083). */
#define SPC_MASK_NONE   0u
#define SPC_MASK_ARRAY  1u
#define SPC_MASK_FINITE 2u
#define SPC_MASK_GT     4u   /* data >  thr_lo */
#define SPC_MASK_GE     8u   /* data >= thr_lo */
#define SPC_MASK_LT     16u  /* data <  thr_hi */
#define SPC_MASK_LE     32u  /* data <= thr_hi */

typedef struct {
    uint32_t flags;
    float thr_lo;
    float thr_hi;
    const uint8_t* d_array;      /* device ptr, may be NULL unless SPC_MASK_ARRAY */
    int64_t row_stride;          /* elements; 0 = same as the cube's */
    int64_t plane_stride;        /* elements; 0 = same as the cube's */
} spc_mask;

/* cube view: d_data points at voxel (0,0,0) of the (sub)cube */
typedef struct {
    const float* d_data;
    int64_t nz, ny, nx;
    int64_t row_stride;          /* elements between (z,y,x) and (z,y+1,x) */
    int64_t plane_stride;        /* elements between (z,y,x) and (z+1,y,x) */
} spc_cube_f32;

/* ---- library / device management -------------------------------------- */
int spc_abi_version(void);
const char* spc_last_error(void);
int spc_device_count(int* count);
typedef struct {
    char name[128];
    char arch[64];               /* e.g. "gfx950:sramecc+:xnack-" */
    int compute_units;
    int wavefront_size;
    int64_t total_mem;
    int64_t free_mem;
    int clock_khz;
} spc_device_info;
int spc_get_device_info(int device, spc_device_info* info);

/* Device buffers.  spc_free keeps the block in a per-device pool (after draining the device, as
 * hipFree does) and spc_malloc hands idle blocks of nearly the requested size out again: cube-sized
 * hipMalloc calls intermittently take seconds on this stack.  The pool is bounded
 * (SPC_POOL_MAX_BYTES, default half of the device memory; SPC_POOL=0 disables it), is emptied when
 * a real allocation runs out of memory, and by spc_pool_trim.  spc_free also accepts pointers that
 * came from a plain hipMalloc.  Buffers are NOT zeroed (hipMalloc does not promise it either);
 * SPC_POOL_POISON=1 fills every returned block with 0xFF bytes to catch code that assumes so. */
int spc_malloc(int device, size_t bytes, void** d_ptr);
int spc_free(int device, void* d_ptr);
int spc_pool_trim(int device);
int spc_pool_stats(int device, int64_t* live_bytes, int64_t* idle_bytes);
int spc_host_alloc(size_t bytes, void** h_ptr);       /* pinned */
int spc_host_free(void* h_ptr);
int spc_memcpy_h2d(int device, void* d_dst, const void* h_src, size_t bytes, void* stream);
int spc_memcpy_d2h(int device, void* h_dst, const void* d_src, size_t bytes, void* stream);
int spc_memcpy_d2d(int device, void* d_dst, const void* d_src, size_t bytes, void* stream);
/* strided 3-D copies (host chunk (nz,cy,cx) <-> device strip); pitches in bytes */
int spc_memcpy3d_h2d(int device, void* d_dst, size_t d_row_pitch, size_t d_plane_pitch,
                     const void* h_src, size_t h_row_pitch, size_t h_plane_pitch,
                     size_t row_bytes, size_t ny, size_t nz, void* stream);
/* the same between two device buffers (row strips of a cube, tiling a cube from a smaller one) */
int spc_memcpy3d_d2d(int device, void* d_dst, size_t dst_row_pitch, size_t dst_plane_pitch,
                     const void* d_src, size_t src_row_pitch, size_t src_plane_pitch,
                     size_t row_bytes, size_t ny, size_t nz, void* stream);
int spc_memset(int device, void* d_ptr, int value, size_t bytes, void* stream);
int spc_stream_create(int device, void** stream);
int spc_stream_destroy(int device, void* stream);
int spc_stream_sync(int device, void* stream);
int spc_device_sync(int device);
int spc_event_create(int device, void** event);
int spc_event_destroy(int device, void* event);
int spc_event_record(int device, void* event, void* stream);
int spc_event_sync(int device, void* event);
int spc_stream_wait_event(int device, void* stream, void* event);   /* hipStreamWaitEvent */
int spc_event_elapsed_ms(int device, void* start, void* stop, float* ms);

/* ---- device scratch -----------------------------------------------------
 * Bytes of d_workspace an entry point needs for a (nz,ny,nx) cube; p0 / p1 are the
 * kernel extents or the output shape it is called with.  An upper bound that only
 * depends on these numbers (not on the data, the mask kind or environment switches),
 * so one buffer sized once serves a whole pipeline of equal-sized calls.  Contents
 * need not be preserved between calls; two calls in flight at the same time (two
 * streams) need two workspaces. */
typedef enum {
    SPC_WS_MOMENTS = 0,               /* spc_moments_f32 (== spc_moments_workspace_bytes) */
    SPC_WS_SPECTRAL_CONV = 1,         /* spc_spectral_conv_f32, p0 = ntaps */
    SPC_WS_SPECTRAL_CONV_MOMENTS = 2, /* spc_spectral_conv_moments_f32, p0 = ntaps */
    SPC_WS_SPATIAL_CONV_SEP = 3,      /* spc_spatial_conv_sep_f32, p0 = nky, p1 = nkx */
    SPC_WS_SPATIAL_CONV2D = 4,        /* spc_spatial_conv2d_f32, p0 = nky, p1 = nkx */
    SPC_WS_RESAMPLE_BILINEAR = 5,     /* spc_resample_bilinear_f32, p0 = ny_out, p1 = nx_out */
    SPC_WS_STATS_GLOBAL = 6,          /* spc_stats_global_f32 */
    SPC_WS_STATS_PLANES = 7,          /* spc_stats_planes_f32 */
    SPC_WS_MAP_CONV2D = 8,            /* spc_map_conv2d_f64, (ny,nx) = map, p0 = nky, p1 = nkx */
    SPC_WS_CLIP_OUTSIDE = 9,          /* spc_clip_outside_f32 */
    SPC_WS_PERCENTILE_GLOBAL = 10,    /* spc_percentile_global_f32 */
    SPC_WS_SPATIAL_CONV_MFMA = 11,    /* spc_spatial_conv_sep_mfma_f32 */
    SPC_WS_SIGMA_CLIP = 12,           /* spc_sigma_clip_axis0_f32 (ABI 7) */
    SPC_WS_RESAMPLE_BILINEAR_LERP = 13, /* spc_resample_bilinear_lerp_f32, nz = INPUT channels, p0 = ny_out, p1 = nx_out (ABI 8) */
    SPC_WS_STATS_GLOBAL_F64 = 14,     /* spc_stats_global_f64 (ABI 8) */
    SPC_WS_SPECTRAL_CONV_F64 = 15,    /* spc_spectral_conv_f64, p0 = ntaps (ABI 8) */
    SPC_WS_SPATIAL_CONV_F64 = 16      /* spc_spatial_conv_f64, p0 = nky, p1 = nkx: the taps + the (num, den) planes of a slab of at
                                       * most 256 MiB (a smaller workspace is accepted as long as one plane fits) (ABI 8) */
} spc_ws_kind;
size_t spc_workspace_bytes(int kind, int64_t nz, int64_t ny, int64_t nx, int64_t p0, int64_t p1);

/* ---- moments ------------------------------------------------------------
 * Replaces DaskSpectralCubeMixin.moment (spectral_cube/dask_spectral_cube.py
 * :1031-1132, arithmetic :1083-1104 and nansum_allbadtonan :54-59),
 * moment_cubewise/_slicewise/_raywise (spectral_cube/_moments.py:30-193),
 * allbadtonan (np_compat.py:3-27) and argmax/argmin
 * (spectral_cube/spectral_cube.py:793-819) for axis 0, in ONE pass:
 *   S0 = sum v, S1 = sum v*c[z], S2 = sum v*c[z]^2 over valid voxels (fp64)
 *   m0 = dv*S0 (NaN when no valid voxel), m1 = S1/S0 + m1_add,
 *   m2 = S2/S0 - (S1/S0)^2
 * d_cen: nz doubles in device memory = pix_cen[z] - c_ref (spectral_cube.py
 * :1473-1475); the host folds c_ref and world(chan 0) into m1_add
 * (dask_spectral_cube.py:1122-1123).  Any output pointer may be NULL.
 * argmax/argmin: first index on ties, 0 for rays without valid voxel.
 * d_workspace: spc_moments_workspace_bytes() bytes (may be NULL if 0). */
typedef struct {
    double* d_m0; double* d_m1; double* d_m2;   /* (ny,nx) float64 */
    double* d_mu;      /* S1/S0 without m1_add (input of spc_moment_order_f32) */
    double* d_s0;      /* raw S0 */
    int64_t* d_argmax; int64_t* d_argmin;       /* (ny,nx) int64 */
    float* d_vmax; float* d_vmin;               /* max/min over valid voxels (NaN if none) */
    int32_t* d_nvalid;
    int64_t out_row_stride;                     /* elements; 0 = nx */
} spc_moment_outputs;

size_t spc_moments_workspace_bytes(int64_t nz, int64_t ny, int64_t nx);
int spc_moments_f32(int device, void* stream, const spc_cube_f32* cube,
                    const spc_mask* mask, const double* d_cen, double dv,
                    double m1_add, const spc_moment_outputs* out,
                    void* d_workspace, size_t workspace_bytes);

/* general order N >= 2 second pass:  out = sum v*(c - mu)^N / S0
 * (dask_spectral_cube.py:1094-1099; _moments.py:185-193).  d_mu, d_s0 come
 * from spc_moments_f32. */
int spc_moment_order_f32(int device, void* stream, const spc_cube_f32* cube,
                         const spc_mask* mask, const double* d_cen, int order,
                         const double* d_mu, const double* d_s0, double* d_out,
                         int64_t out_row_stride);

/* ---- float64 cubes: moments along the spectral axis in the SOURCE's precision ------------------
 * The reference keeps a float64 source in float64 (np.result_type(dtype, 0.0), spectral_cube/masks.py:225; a
 * BITPIX = -64 / 32 / 64 FITS image arrives as float64 through astropy, io/fits.py:63-172) and so are its moment maps
 * (spectral_cube/_moments.py:30-193, dask_spectral_cube.py:1083-1104).  Same arguments and outputs as spc_moments_f32 /
 * spc_moment_order_f32 with 8-byte samples, float64 thresholds and float64 extrema; no workspace.  There is no d_m2:
 * moment 2 is spc_moment_order_f64(order = 2) about the first pass's d_mu - the reference's own two-pass form
 * (_moments.py:185-193), which keeps float64 precision where S2 / S0 - mu^2 would not.  nz < 2^21. */
typedef struct {
    const double* d_data;
    int64_t nz, ny, nx;
    int64_t row_stride;          /* elements */
    int64_t plane_stride;        /* elements */
} spc_cube_f64;

typedef struct {
    uint32_t flags;              /* SPC_MASK_* */
    double thr_lo;
    double thr_hi;
    const uint8_t* d_array;
    int64_t row_stride;          /* elements; 0 = same as the cube's */
    int64_t plane_stride;        /* elements; 0 = same as the cube's */
} spc_mask_f64;

typedef struct {
    double* d_m0; double* d_m1;                 /* (ny,nx) float64 */
    double* d_mu;      /* S1/S0 without m1_add (input of spc_moment_order_f64) */
    double* d_s0;      /* raw S0 */
    int64_t* d_argmax; int64_t* d_argmin;       /* (ny,nx) int64 */
    double* d_vmax; double* d_vmin;             /* max/min over valid voxels (NaN if none) */
    int32_t* d_nvalid;
    int64_t out_row_stride;                     /* elements; 0 = nx */
} spc_moment_outputs_f64;

int spc_moments_f64(int device, void* stream, const spc_cube_f64* cube,
                    const spc_mask_f64* mask, const double* d_cen, double dv,
                    double m1_add, const spc_moment_outputs_f64* out);

int spc_moment_order_f64(int device, void* stream, const spc_cube_f64* cube,
                         const spc_mask_f64* mask, const double* d_cen, int order,
                         const double* d_mu, const double* d_s0, double* d_out,
                         int64_t out_row_stride);

/* ---- the other operators for float64 cubes (ABI 8) ---------------------------
 * The reference keeps a float64 source in float64 through every operator (np.result_type(dtype, 0.0),
 * spectral_cube/masks.py:225; the Dask class keeps the chunk dtype, dask_spectral_cube.py:829).  Same semantics and
 * arithmetic as the float32 entry points of the same name - astropy's convolve (float64 top / bot, one division),
 * scipy's interp1d slope form, nansum_allbadtonan - without the rounding of samples or results to float32; mask
 * thresholds are compared in float64.  Plain HBM streams, not tuned like the float32 kernels.
 *   spc_stats_global_f64: h_stats (HOST, 5 doubles) = {npts, min, max, sum, sumsq} (dask_spectral_cube.py:769-814);
 *                         synchronises the stream.  Workspace SPC_WS_STATS_GLOBAL_F64.
 *   spc_stats_axis_f64:   maps with the reduced axis removed, (ny,nx) / (nz,nx) / (nz,ny); NULL outputs are skipped; a
 *                         ray without an included sample gives count 0 and NaN (:641-767).
 *   spc_spectral_conv_f64 / spc_spatial_conv_f64: spectral_smooth / spatial_smooth (:880-917, :962-993); host taps.
 *                         spatial: separable != 0 takes the two factors h_ky (nky) and h_kx (nkx) of an outer-product
 *                         kernel; separable == 0 takes h_ky = the (nky, nkx) table, row-major, and ignores h_kx.
 *   spc_spectral_lerp_f64: spectral_interpolate (:1342-1353), the plan of spc_spectral_lerp_f32.
 *   spc_narrow_f64_to_f32: the float32 copy of a float64 cube (what an operator without a float64 form is given).
 *   spc_mask_include_f64: spc_mask_include_u8 on a float64 cube. */
typedef struct spc_stats_outputs_f64 {
    int32_t* d_count;
    double* d_min;
    double* d_max;
    double* d_sum;
    double* d_sumsq;
} spc_stats_outputs_f64;
int spc_stats_global_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask,
                         double* h_stats, void* d_workspace, size_t workspace_bytes);
int spc_stats_axis_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask,
                       int axis, const spc_stats_outputs_f64* out);
int spc_spectral_conv_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask,
                          const double* h_kernel, int ntaps, double* d_out, int64_t out_row_stride,
                          int64_t out_plane_stride, void* d_workspace, size_t workspace_bytes);
int spc_spatial_conv_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask,
                         const double* h_ky, int nky, const double* h_kx, int nkx, int separable,
                         double* d_out, int64_t out_row_stride, int64_t out_plane_stride,
                         void* d_workspace, size_t workspace_bytes);
int spc_spectral_lerp_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask,
                          int64_t nz_out, const int32_t* d_lo, const double* d_t, const double* d_inv_dx,
                          double fill, double* d_out, int64_t out_row_stride, int64_t out_plane_stride);
/* spc_resample_bilinear_f32 on a float64 cube: float64 weights and results (reproject_interp computes in float64), gather
 * form; spc_scale_f64: d_data[i] *= factor (the Jy/beam area ratio of convolve_to, dask_spectral_cube.py:1445-1464). */
int spc_resample_bilinear_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask,
                              double fill, int64_t ny_out, int64_t nx_out, const double* d_xs, const double* d_ys,
                              double* d_out, int64_t out_row_stride, int64_t out_plane_stride,
                              uint8_t* d_footprint, int order, uint32_t* d_any_valid);
int spc_scale_f64(int device, void* stream, double* d_data, int64_t n, double factor);
/* Order statistics of float64 rays along the first axis of the view (spc_percentile_axis0_f32 / spc_sigma_clip_axis0_f32 on the
 * float64 samples; dask_spectral_cube.py:657-731, :851-878): q-th percentile with numpy's linear rule (q = 50: the median, the
 * mean of two middle samples), of |x - d_center| when d_center (a (ny,nx) float64 map) is given, times scale (mad_std); and the
 * whole sigma-clip loop for centre = median | mean and spread = std | mad_std (float64 centre and bounds; d_out (nz,ny,nx) C-contiguous
 * float64, masked and clipped samples NaN).  Rays of up to 4096 samples (sorted in LDS), else SPC_ERR_UNSUPPORTED. */
int spc_percentile_axis0_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask,
                             double q, const double* d_center, double scale, double* d_out);
int spc_sigma_clip_axis0_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask,
                             double sigma_lower, double sigma_upper, int maxiters, int center_is_mean,
                             int spread_is_mad, double* d_out);
int spc_narrow_f64_to_f32(int device, void* stream, const spc_cube_f64* cube, float* d_out,
                          int64_t out_row_stride, int64_t out_plane_stride);
int spc_mask_include_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask,
                         int nan_excluded, uint8_t* d_out);

/* ---- FITS payload -> float32 -------------------
 * Converts n raw big-endian FITS image samples (already in HBM) to native
 * float32: what astropy.io.fits does on the host behind
 * spectral_cube/io/fits.py:63-172 (read_data_fits) / :171-260 (load_fits_cube).
 * bitpix in {8, 16, 32, 64, -32, -64}.  Scaling follows astropy: BITPIX 8/16 in
 * float32 (raw * BSCALE + BZERO), 32/64 in float64 then rounded, floating types
 * in their own precision; has_blank/blank: integer BLANK value -> NaN. */
int spc_fits_to_f32(int device, void* stream, const void* d_raw, int bitpix,
                    double bscale, double bzero, int has_blank, int64_t blank,
                    int64_t n, float* d_out);

/* The wide sample types in their own precision: bitpix in {-64, 32, 64} -> native float64 (what astropy hands the
 * reference for such an image, spectral_cube/io/fits.py:63-172); feeds spc_moments_f64. */
int spc_fits_to_f64(int device, void* stream, const void* d_raw, int bitpix,
                    double bscale, double bzero, int has_blank, int64_t blank,
                    int64_t n, double* d_out);

/* d_data[i] *= factor over n contiguous floats: the Jy/beam rescaling by the ratio of beam
 * areas in convolve_to (spectral_cube/dask_spectral_cube.py:1450-1457). */
int spc_scale_f32(int device, void* stream, float* d_data, int64_t n, double factor);

/* ---- statistics ------------------------------
 * One read of the cube gives count / min / max / sum / sum of squares of the
 * included, non-NaN samples, accumulated in float64.
 *
 * spc_stats_global_f32 replaces the per-chunk compute_stats + aggregation of
 * DaskSpectralCubeMixin.statistics (spectral_cube/dask_spectral_cube.py:769-814)
 * and the axis=None forms of sum / mean / std / max / min (:641-767).
 * h_stats (HOST, 5 doubles) = {npts, min, max, sum, sumsq}; min / max are NaN
 * when nothing is included.  Synchronises the stream. */
int spc_stats_global_f32(int device, void* stream, const spc_cube_f32* cube,
                         const spc_mask* mask, double* h_stats, void* d_workspace, size_t workspace_bytes);

/* Reductions along one axis (0, 1 or 2) behind sum / mean / std / max / min with
 * axis given (dask_spectral_cube.py:641-767; spectral_cube.py:578-791).  Output
 * maps are C-contiguous with the reduced axis removed: (ny,nx), (nz,nx), (nz,ny).
 * NULL outputs are skipped.  A ray without included samples gives count 0 and
 * NaN in the four floating maps (nansum_allbadtonan, nanmin / nanmax of all-NaN). */
/* The same five numbers for every channel: h_stats (HOST) = nz records {npts, min, max, sum, sumsq}
 * of the (ny,nx) planes - the nan-reductions with axis=(1, 2), e.g. the mean spectrum cube.mean(axis=(1, 2)). */
int spc_stats_planes_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                         double* h_stats, void* d_workspace, size_t workspace_bytes);

typedef struct spc_stats_outputs {
    int32_t* d_count;
    float* d_min;
    float* d_max;
    double* d_sum;
    double* d_sumsq;
} spc_stats_outputs;
int spc_stats_axis_f32(int device, void* stream, const spc_cube_f32* cube,
                       const spc_mask* mask, int axis, const spc_stats_outputs* out);

/* argmax / argmin along a SPATIAL axis (BaseSpectralCube.argmax / argmin with axis = 1 or 2,
 * spectral_cube/spectral_cube.py:793-819; axis 0 is an output of spc_moments_f32): nanargmax of
 * the data filled with -inf (argmin: +inf) - excluded and NaN samples take the fill, the first
 * index wins ties, rays without an included sample give 0.  int64 maps (nz,nx) / (nz,ny),
 * C-contiguous; a NULL output is skipped. */
int spc_argextrema_axis_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                            int axis, int64_t* d_argmin, int64_t* d_argmax);

/* ---- order statistics along the spectral axis ---
 * q-th percentile (numpy 'linear' interpolation; q = 50: the median) of the
 * included, non-NaN samples of every ray: DaskSpectralCubeMixin.median /
 * percentile (spectral_cube/dask_spectral_cube.py:657-693).  With d_center (a
 * (ny,nx) float32 map) the statistic is taken of |x - center|, and the result is
 * multiplied by scale: median absolute deviation -> mad_std (:711-731, astropy
 * stats.mad_std: scale = 1.482602218505602).  Rays without a valid sample give
 * NaN.  d_out: (ny,nx) float32, C-contiguous.
 * The rays run along the FIRST axis of the view that is passed in: for a selection along y
 * of a (nz, ny, nx) cube hand over the same buffer as {nz' = ny, ny' = nz, row_stride' =
 * plane_stride, plane_stride' = row_stride} (mask strides likewise); the result is (nz, nx). */
int spc_percentile_axis0_f32(int device, void* stream, const spc_cube_f32* cube,
                             const spc_mask* mask, double q, const float* d_center,
                             float scale, float* d_out);
/* The same along x (axis 2) without a transposed copy: rays = the rows of the cube, which are contiguous; d_center and
 * d_out are (nz, ny).  Rows of more than 4096 samples: SPC_ERR_UNSUPPORTED (transpose with
 * spc_fill_masked_transpose_f32 and use the exchanged-stride form above). */
int spc_percentile_axis2_f32(int device, void* stream, const spc_cube_f32* cube,
                             const spc_mask* mask, double q, const float* d_center,
                             float scale, float* d_out);

/* The same statistic over the WHOLE cube (median / percentile / mad_std with axis=None): four
 * histogram passes over the key bytes.  With has_center the statistic is taken of
 * |x - center| (float32 arithmetic, as numpy does for a float32 cube).  *h_out (HOST) gets the
 * value in double; NaN when nothing is included. */
int spc_percentile_global_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                              double q, int has_center, float center, double* h_out,
                              void* d_workspace, size_t workspace_bytes);

/* ONE pass of that selection, for a cube that is sharded over ranks (each rank calls this on its strip, the ranks add
 * their counters - or take the minimum of their next keys - and walk them together; the reference's counterpart is
 * the dask reduction tree under dask_spectral_cube.py:657-693).  Samples are compared through their order-preserving
 * 32-bit keys (of |x - center| with has_center); spc_key_to_f32 maps a key back to its float.
 *   h_hist != NULL (HOST, 256 counters): h_hist[d] = included samples whose key agrees with `prefix` on the bits of
 *       `pmask` and carries byte d at bit `shift` (24, 16, 8 or 0);
 *   h_next != NULL (HOST): the smallest key above `prefix`, 0xffffffff when there is none.
 * Exactly one of the two must be given.  Workspace: SPC_WS_PERCENTILE_GLOBAL. */
int spc_key_histogram_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                          uint32_t prefix, uint32_t pmask, int shift, int has_center, float center,
                          uint64_t* h_hist, uint32_t* h_next, void* d_workspace, size_t workspace_bytes);
float spc_key_to_f32(uint32_t key);

/* out[z][x][y] = included ? data[z][y][x] : fill, d_out a C-contiguous (nz, nx, ny) buffer: the
 * filled copy with the spatial axes exchanged, which turns an order statistic along x
 * (median(axis=2)) into one along y for spc_percentile_axis0_f32's exchanged-stride form. */
int spc_fill_masked_transpose_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                  float fill, float* d_out);

/* ---- sigma clipping along the spectral axis (dask_spectral_cube.py:851-878) ----
 * astropy.stats.sigma_clip(axis=0, masked=False) iterates: per-ray centre and std
 * (spc_percentile_axis0_f32 / spc_stats_axis_f32), then everything outside
 * [centre - sigma_lower*std, centre + sigma_upper*std] becomes NaN, until nothing
 * changes.  These are the two elementwise ends of that loop:
 * spc_fill_masked_f32: out = included ? data : fill   (the NaN-filled working copy,
 *                      MaskBase._filled, masks.py:197-237);
 * spc_clip_outside_f32: in place, contiguous (nz,ny,nx): v < lo[y,x] || v > hi[y,x] -> NaN;
 *                      *h_nchanged (HOST) = number of samples clipped by this call. */
int spc_fill_masked_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                        float fill, float* d_out, int64_t out_row_stride, int64_t out_plane_stride);
/* The whole loop in ONE kernel for centre = median (center_is_mean = 0) or mean and spread = std (spread_is_mad = 0) or
 * mad_std (1.4826 x the median of |x - median|, a second selection per iteration): the rays stay in
 * registers across the iterations, the cube is read once and the clipped copy written once (d_out: (nz,ny,nx)
 * C-contiguous float32; masked and clipped samples NaN).  maxiters < 0: until nothing changes.  Same arithmetic as
 * the pieces above (float64 sums, float32 centre and bounds).  Rays of more than 4096 channels: SPC_ERR_UNSUPPORTED.
 * (ABI 7) d_workspace / workspace_bytes: spc_workspace_bytes(SPC_WS_SIGMA_CLIP, nz, ny, nx, 0, 0) bytes of caller-owned
 * scratch, or NULL / 0.  With it the kernel's block shape follows the mask (a probe counts the valid samples of 64 rays: rays
 * of at most 128 valid samples - signal masks - are packed and clipped by one wave each, in blocks whose runs per plane are
 * twice as wide); without it the shape is the one that suits dense rays. */
int spc_sigma_clip_axis0_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                             double sigma_lower, double sigma_upper, int maxiters, int center_is_mean,
                             int spread_is_mad, float* d_out, void* d_workspace, size_t workspace_bytes);
/* The include map of a mask evaluated on the cube it is bound to, as a uint8 (nz,ny,nx) array in
 * HBM: d_out = included ? 1 : 0 (MaskBase.include, masks.py:105-116).  For masks that belong to
 * ANOTHER cube's data - a smoothed cube keeps its parent's mask object (dask_spectral_cube.py:836-840):
 * the predicate terms run on the parent's device data instead of a host copy of it.  nan_excluded != 0
 * additionally drops NaN samples (~isnan style masks). */
int spc_mask_include_u8(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                        int nan_excluded, uint8_t* d_out);
/* per-ray clipping bounds of one iteration, all on the device: lo = centre - sigma_lower * std,
 * hi = centre + sigma_upper * std over n rays.  centre = d_center (e.g. the median map) or, when
 * NULL, the mean from d_count / d_sum / d_sumsq; std = d_spread (e.g. the mad_std map) or, when
 * NULL, sqrt(max(sumsq / n - mean^2, 0)) - float64 arithmetic, float32 bounds like astropy's. */
int spc_clip_bounds_f32(int device, void* stream, int64_t n, const int32_t* d_count, const double* d_sum,
                        const double* d_sumsq, const float* d_center, const float* d_spread,
                        double sigma_lower, double sigma_upper, float* d_lo, float* d_hi);
int spc_clip_outside_f32(int device, void* stream, float* d_cube, int64_t nz, int64_t ny, int64_t nx,
                         const float* d_lo, const float* d_hi, uint64_t* h_nchanged,
                         void* d_workspace, size_t workspace_bytes);

/* 2-D convolution of one (ny, nx) float64 map with an odd-sized kernel (zero fill outside,
 * kernel normalised by its sum, true convolution, NaN propagates).  Serves the algebraic
 * spatial_smooth -> moment path: when every voxel is valid, astropy's convolution
 * (dask_spectral_cube.py:962-993) commutes with the sums along the spectral axis
 * (:1083-1104), so the moment sums of the smoothed cube are the smoothed moment sums. */
int spc_map_conv2d_f64(int device, void* stream, const double* d_in, int64_t ny, int64_t nx,
                       const double* h_kernel, int nky, int nkx, double* d_out,
                       void* d_workspace, size_t workspace_bytes);

/* Elementwise arithmetic on float64 maps of n elements, so that the algebra around spc_map_conv2d_f64 (moment
 * sums from moments, moments from smoothed sums) stays on the device:
 *   SPC_MAP_MUL                out = a * b
 *   SPC_MAP_SECOND_MOMENT_SUM  out = (a + b*b) * c        S2 = (m2 + mu^2) * S0
 *   SPC_MAP_DIV_ADD            out = a / b + s            mu' = S1' / S0' (+ axis offset)
 *   SPC_MAP_DIV_SUB_SQ         out = a / b - c*c          m2' = S2' / S0' - mu'^2
 * d_out may alias an input. */
typedef enum { SPC_MAP_MUL = 0, SPC_MAP_SECOND_MOMENT_SUM = 1, SPC_MAP_DIV_ADD = 2, SPC_MAP_DIV_SUB_SQ = 3 } spc_map_op;
int spc_map_arith_f64(int device, void* stream, int op, const double* d_a, const double* d_b, const double* d_c,
                      double s, double* d_out, int64_t n);

/* The checks the algebraic smooth -> moment paths make before they trust their shortcut, on the device: *d_flags (one
 * uint32 in HBM, cleared by this call on the same stream) gets bit 0 if any of the n int32 counts differs from
 * `expect` (a spaxel with an invalid voxel), bit 1 if any of the n float64 values is not finite.  Either array may be
 * NULL.  Asynchronous: the caller reads back four bytes instead of two maps. (ABI 4) */
int spc_map_check(int device, void* stream, const int32_t* d_counts, int32_t expect, const double* d_values,
                  int64_t n, uint32_t* d_flags);

/* moments along a spatial axis (axis = 1 or 2), reference golden tables
 * spectral_cube/tests/test_moments.py:19-49.  d_cen is a (ny,nx) float64 map
 * of offsets along that axis (spectral_cube.py:1476-1503), pix_size the pixel
 * scale (spectral_cube.py:1530-1533).  Outputs have shape (nz,nx) for axis 1
 * and (nz,ny) for axis 2, C-contiguous. */
int spc_moments_spatial_f32(int device, void* stream, const spc_cube_f32* cube,
                            const spc_mask* mask, int axis, const double* d_cen,
                            double pix_size, double* d_m0, double* d_m1,
                            double* d_m2);

/* order N >= 2 along a spatial axis, second pass: out = sum v (c - mu)^N / sum v with d_mu = the
 * moment-1 map of spc_moments_spatial_f32 (offsets, no world coordinate added), same output shapes
 * (dask_spectral_cube.py:1094-1099; _moments.py:185-193 with axis != 0). */
int spc_moment_order_spatial_f32(int device, void* stream, const spc_cube_f32* cube,
                                 const spc_mask* mask, int axis, const double* d_cen, int order,
                                 const double* d_mu, double* d_out);

/* ---- NaN-aware convolution (astropy.convolution.convolve semantics:
 * boundary='fill', fill_value=0, nan_treatment='interpolate',
 * normalize_kernel=True) ---------------------------------------------------
 * spectral: replaces the chunk function of DaskSpectralCubeMixin.
 * spectral_smooth (dask_spectral_cube.py:880-917; NumPy twin
 * spectral_cube.py:3186-3222).  h_kernel: ntaps (odd) doubles on the HOST.
 * Input voxels failing the mask are treated as NaN (the reference convolves
 * the NaN-filled chunk).  Output float32, same shape. */
int spc_spectral_conv_f32(int device, void* stream, const spc_cube_f32* cube,
                          const spc_mask* mask, const double* h_kernel, int ntaps,
                          float* d_out, int64_t out_row_stride,
                          int64_t out_plane_stride, void* d_workspace, size_t workspace_bytes);

/* fused spectral_smooth -> moments (legal because the Dask smooth is lazy and
 * keeps the ORIGINAL mask, dask_spectral_cube.py:836-840): the smoothed cube
 * is never written.  Semantics = spc_spectral_conv_f32 followed by
 * spc_moments_f32 with the same mask re-applied to the smoothed values.
 * h_cen: optional HOST copy of d_cen (nz doubles, may be NULL); when it shows
 * the axis is linear the kernel derives the offsets from the channel index
 * instead of loading them (every FITS spectral axis is linear). */
int spc_spectral_conv_moments_f32(int device, void* stream, const spc_cube_f32* cube,
                                  const spc_mask* mask, const double* h_kernel,
                                  int ntaps, const double* d_cen, const double* h_cen,
                                  double dv, double m1_add, const spc_moment_outputs* out,
                                  void* d_workspace, size_t workspace_bytes);

/* spatial: replaces the per-channel 2-D convolution of spatial_smooth
 * (dask_spectral_cube.py:962-993 + :540-547; NumPy twin spectral_cube.py
 * :2808-2842).  Separable form: kernel2d = outer(h_ky, h_kx) (exact for
 * Gaussian2DKernel).  Non-separable kernels: spc_spatial_conv2d_f32. */
int spc_spatial_conv_sep_f32(int device, void* stream, const spc_cube_f32* cube,
                             const spc_mask* mask, const double* h_ky, int nky,
                             const double* h_kx, int nkx, float* d_out,
                             int64_t out_row_stride, int64_t out_plane_stride,
                             void* d_workspace, size_t workspace_bytes);
int spc_spatial_conv2d_f32(int device, void* stream, const spc_cube_f32* cube,
                           const spc_mask* mask, const double* h_kernel, int nky,
                           int nkx, float* d_out, int64_t out_row_stride,
                           int64_t out_plane_stride, void* d_workspace, size_t workspace_bytes);

/* masked separable spatial_smooth with the DENOMINATOR on the matrix cores, optionally fused with moment 0
 * (dask_spectral_cube.py:962-993 then :1083-1104: the Dask graph never materialises the smoothed cube either).
 * out = sum k d [valid] / sum k [valid] like spc_spatial_conv_sep_f32; the numerator is float32 vector arithmetic
 * (two columns per packed FMA), the denominator - a convolution of a 0 / 1 array - is two banded-Toeplitz matrix
 * products per 16 x 16 tile on v_mfma_f32_16x16x32_f16: mask bits are exact in fp16, the (power-of-two scaled) taps and
 * the intermediate go in as fp16 hi + lo pairs with float32 accumulation (<= 1e-6 relative on the denominator).
 * One block = a band of 16 rows x a strip of 480 columns, walking a chunk of channels.
 *   d_out  (may be NULL): the smoothed cube (nz, ny, nx), float32.
 *   d_m0   (may be NULL): moment 0 of the smoothed cube under the ORIGINAL mask, dv * nansum over channels, NaN where no
 *          channel contributes (float64 (ny, nx), row stride m0_row_stride or nx) - the cube is then never written.
 * SPC_ERR_UNSUPPORTED (the caller falls back to spc_spatial_conv_sep_f32 + spc_moments_f32): more than 29 (third form: 33) taps per
 * axis, a negative tap or a zero centre tap, mask terms other than SPC_MASK_ARRAY / SPC_MASK_FINITE, odd nx or strides.
 * (ABI 4) */
int spc_spatial_conv_sep_mfma_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                  const double* h_ky, int nky, const double* h_kx, int nkx,
                                  float* d_out, int64_t out_row_stride, int64_t out_plane_stride,
                                  double dv, double* d_m0, int64_t m0_row_stride,
                                  void* d_workspace, size_t workspace_bytes);

/* (ABI 6) The same operator with moments 1 / 2 of the smoothed cube as well, and - for both entry points, wherever nx and
 * the strides are multiples of 4 and the bases 16-byte aligned - a third form in which EVERY product runs on
 * v_mfma_f32_16x16x32_f16: the masked samples go in as fp16 hi + lo under one power-of-two scale per step of 16 rows
 * (the running maximum of the wave's channel), the taps as fp16 hi + lo, the x-pass result is split again for the y pass
 * (spc_spatial_split.hip; <= 1e-6 of the data range against float64).  Moments follow spc_moments_f32's conventions:
 *   d_cen  nz doubles in device memory = pix_cen[z] - c_ref;  m1 = S1 / S0 + m1_add,  m2 = S2 / S0 - (S1 / S0)^2,
 *   S_n = nansum over channels of c^n * smoothed value under the ORIGINAL mask (dask_spectral_cube.py:1083-1104 on the
 *   lazily smoothed cube of :962-993); rays without a contribution: m0 NaN, m1 / m2 NaN (0 / 0).
 * Any of d_out, d_m0, d_m1, d_m2 may be NULL (not all).  Workspace: spc_workspace_bytes(SPC_WS_SPATIAL_CONV_MFMA, nz, ny,
 * nx, 3, 0) when d_m1 or d_m2 is asked for.  SPC_ERR_UNSUPPORTED as above, and for moments 1 / 2 where the third form
 * does not apply. */
int spc_spatial_conv_sep_mfma_moments_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                          const double* h_ky, int nky, const double* h_kx, int nkx,
                                          float* d_out, int64_t out_row_stride, int64_t out_plane_stride,
                                          const double* d_cen, double dv, double m1_add,
                                          double* d_m0, double* d_m1, double* d_m2, int64_t map_row_stride,
                                          void* d_workspace, size_t workspace_bytes);

/* ---- resampling -----------------------------------------------------------
 * spectral lerp: replaces interp_wrapper / scipy interp1d(kind='linear') of
 * DaskSpectralCubeMixin.spectral_interpolate (dask_spectral_cube.py:1342-1353).
 * For output channel j: lo = d_lo[j] (int32 input channel), and
 *   out = (in[lo+1]-in[lo]) * d_inv_dx[j] * d_t[j] + in[lo]      (float64 maths)
 * d_lo[j] < 0 marks out-of-range channels which receive `fill`.
 * NaN in either bracketing sample gives NaN (scipy semantics). */
int spc_spectral_lerp_f32(int device, void* stream, const spc_cube_f32* cube,
                          const spc_mask* mask, int64_t nz_out, const int32_t* d_lo,
                          const double* d_t, const double* d_inv_dx, float fill,
                          float* d_out, int64_t out_row_stride,
                          int64_t out_plane_stride);

/* Pixel map of a reprojection, on the device: for every pixel of the target grid (wcs_out) the
 * 0-based pixel coordinates in the source grid (wcs_in) - what reproject_interp obtains from
 * astropy.wcs (pixel_to_world on the target, world_to_pixel on the source; spectral_cube.py:2700-2732).
 * FITS paper II arithmetic in float64 for the zenithal projections TAN / SIN / ARC / STG / ZEA and the
 * (pseudo-)cylindrical CAR / SFL / CEA / MER / AIT, with astropy's SIP stage on either side (all_pix2world /
 * all_world2pix); pixels that cannot be projected (or lie beyond the edge of the sky of an all-sky projection, or
 * where the SIP inverse does not converge) get -1e30 (outside every footprint).  The host fills the
 * struct from the header (spectral_cube_amd/wcs.py). */
#define SPC_SIP_MAX_ORDER 9
#define SPC_SIP_TERMS 55           /* coefficients of u^p v^q, p + q <= 9: row p starts at p * 10 - p * (p - 1) / 2 */
typedef struct spc_celestial_wcs {
    int32_t proj;              /* 0 TAN, 1 SIN, 2 ARC, 3 STG, 4 ZEA, 5 CAR, 6 SFL, 7 CEA, 8 MER, 9 AIT */
    int32_t sip_order;         /* 0 = no SIP distortion; else max(A_ORDER, B_ORDER) <= 9 (ABI 4; was `reserved`) */
    double crpix[2];           /* FITS 1-based reference pixel (x, y) */
    double lin[4];             /* CDELT_i * PC_ij, 2 x 2 row-major: degrees per pixel */
    double lin_inv[4];         /* its inverse */
    double alpha_p, delta_p;   /* celestial coordinates of the native pole (radians) */
    double phi_p;              /* LONPOLE (radians) */
    double pv1;                /* CEA: PV2_1 (lambda), else unused (ABI 3) */
    double plane0[2];          /* (x0, y0) degrees: the projection of the user's fiducial point (PV1_0 != 0 with PV1_1 /
                                * PV1_2; wcslib prjoff), added before deprojection, subtracted after projection (ABI 4) */
    double sip_a[SPC_SIP_TERMS];   /* SIP forward polynomials (Shupe et al. 2005): u' = u + A(u, v), v' = v + B(u, v) with */
    double sip_b[SPC_SIP_TERMS];   /* (u, v) = pixel - CRPIX; as the TARGET they are evaluated, as the SOURCE they are
                                    * inverted by Newton's method - the limit of astropy's all_world2pix iteration (ABI 4) */
} spc_celestial_wcs;
/* frame_rot (HOST pointer, 15 doubles, may be NULL = same frame): [0..8] row-major rotation of the unit sphere that
 * takes the TARGET's celestial frame to the SOURCE's (ICRS / FK5(equinox) / FK4-NO-E(equinox) / Galactic:
 * spectral_cube_amd/wcs.py::frame_transform) - reproject_interp transforms the target's sky coordinates to the source's
 * frame before it asks the source WCS for pixels (the reference's own test goes RA/DEC -> GLON/GLAT,
 * tests/test_regrid.py:99-135); [9..11] the E-terms of aberration removed from the unit vector BEFORE the rotation (an
 * FK4 target; zeros = none), [12..14] the E-terms added AFTER it (an FK4 source; zeros = none).
 * (ABI 3: the argument; ABI 4: 15 doubles instead of 9.) */
int spc_wcs_pixel_map_f64(int device, void* stream, const spc_celestial_wcs* wcs_out,
                          const spc_celestial_wcs* wcs_in, const double* frame_rot,
                          int64_t ny_out, int64_t nx_out, double* d_xs, double* d_ys);

/* spatial resample: replaces the inner resampler of
 * reproject.reproject_interp(order='bilinear' | 'nearest-neighbor') called from
 * BaseSpectralCube.reproject (spectral_cube.py:2700-2732).  d_xs, d_ys:
 * (ny_out,nx_out) float64 source pixel coordinates (0-based).  Output pixels
 * whose source falls outside [-0.5, n-0.5] are NaN and get footprint 0.
 * order: 1 = bilinear (a NaN neighbour propagates even with weight 0, as in
 * scipy.ndimage.map_coordinates), 0 = nearest neighbour (floor(x + 0.5)).
 * d_footprint (uint8, (ny_out,nx_out)) may be NULL.  d_any_valid (one uint32 in
 * HBM, may be NULL) is set to 1 iff some output value is not NaN: the reference's
 * "All values in reprojected cube are nan" check (spectral_cube.py:2733-2739)
 * without another pass over the output. */
int spc_resample_bilinear_f32(int device, void* stream, const spc_cube_f32* cube,
                              const spc_mask* mask, float fill, int64_t ny_out,
                              int64_t nx_out, const double* d_xs, const double* d_ys,
                              float* d_out, int64_t out_row_stride,
                              int64_t out_plane_stride, uint8_t* d_footprint,
                              int order, uint32_t* d_any_valid,
                              void* d_workspace, size_t workspace_bytes);

/* spatial resample with the spectral interpolation folded in (ABI 8): ONE pass for what the reference does as
 * spectral_interpolate (dask_spectral_cube.py:1342-1353) followed by reproject (spectral_cube.py:2700-2732), and for
 * reproject onto a cube header whose spectral axis differs from the cube's (reproject_interp's single trilinear call for
 * a separable WCS, spectral_cube.py:2726-2732).  Output channel j (nz_out of them) is
 *   out[j] = (R[lo+1] - R[lo]) * d_inv_dx[j] * d_t[j] + R[lo],   lo = d_lo[j],
 * where R[k] is input channel k resampled exactly as spc_resample_bilinear_f32 does (mask, fill, order, footprint as there)
 * and the blend is spc_spectral_lerp_f32's arithmetic - both operators are linear interpolations with NaN propagation,
 * so they commute: a value is NaN iff one of its eight source samples is excluded / NaN or the pixel lies outside the
 * footprint, as in the two-pass form; finite values agree with it to float32 rounding (each input plane is read and
 * resampled ONCE, nz instead of nz_out gathers, and the intermediate cube is never written).  d_lo[j] < 0 marks channels
 * outside the input range (NaN planes).  Precondition: the non-negative entries of d_lo ascend and are contiguous in j
 * (an ascending output grid on ascending input channels; a descending plan is run reversed, with d_out at its LAST plane and
 * a negative out_plane_stride: the strides are signed); cube->nz >= 2.
 * Workspace: spc_workspace_bytes(SPC_WS_RESAMPLE_BILINEAR_LERP, nz, ny, nx, ny_out, nx_out). */
int spc_resample_bilinear_lerp_f32(int device, void* stream, const spc_cube_f32* cube,
                                   const spc_mask* mask, float fill, int64_t ny_out,
                                   int64_t nx_out, const double* d_xs, const double* d_ys,
                                   int64_t nz_out, const int32_t* d_lo, const double* d_t,
                                   const double* d_inv_dx, float* d_out, int64_t out_row_stride,
                                   int64_t out_plane_stride, uint8_t* d_footprint,
                                   int order, uint32_t* d_any_valid,
                                   void* d_workspace, size_t workspace_bytes);

/* spline resample: reproject_interp(order='biquadratic' | 'bicubic') for channels that map onto themselves
 * (spectral_cube.py:2667-2676 documents the orders; :2726-2732 is the call).  order 2 / 3 =
 * scipy.ndimage.map_coordinates(order, mode='constant', cval=nan) on the planes replicated by one border pixel: B-spline
 * prefilter with mirror boundaries along y and x, 3 x 3 / 4 x 4 gather, float64 throughout, NaN outside
 * [-0.5, n - 0.5].  The cube must hold finite samples only (scipy's recursive prefilter turns ONE non-finite sample into
 * an all-NaN result; the caller checks and raises the reference's "All values in reprojected cube are nan").
 * d_workspace: nz x (ny + 2) x (nx + 2) float64 coefficients (the caller resamples a slab of channels at a time). (ABI 4) */
int spc_resample_spline_f32(int device, void* stream, const spc_cube_f32* cube, int order, int64_t ny_out,
                            int64_t nx_out, const double* d_xs, const double* d_ys, float* d_out,
                            int64_t out_row_stride, int64_t out_plane_stride, uint8_t* d_footprint,
                            void* d_workspace, size_t workspace_bytes);

/* ---- multi-GPU stitch (RCCL over xGMI) -----------------------------------
 * One process per GPU; each rank owns a row strip of the 2-D map.  The id is
 * created on rank 0 (spc_comm_unique_id) and distributed by the host
 * launcher (torch.distributed store / file), then every rank calls
 * spc_comm_init.  spc_allgather_rows gathers equal-sized strips. */
#define SPC_COMM_ID_BYTES 128
int spc_comm_unique_id(uint8_t id[SPC_COMM_ID_BYTES]);
int spc_comm_init(int device, const uint8_t id[SPC_COMM_ID_BYTES], int nranks,
                  int rank, void** comm);
int spc_comm_destroy(void* comm);
int spc_allgather_rows(void* comm, void* stream, const void* d_send, void* d_recv,
                       size_t bytes_per_rank);
/* n all-gathers in ONE grouped launch (ncclGroupStart / ncclGroupEnd): e.g. the three moment maps of a row chunk,
 * each gathered into its own destination map. */
int spc_allgather_rows_batch(void* comm, void* stream, int n, const void* const* d_send, void* const* d_recv,
                             const size_t* bytes_per_rank);

#ifdef __cplusplus
}
#endif
#endif /* SPCUBE_HIP_H */
