// FITS image payload -> native float32 on the device (the step
// before the hot path).  The reference reads cubes through astropy.io.fits
// (spectral_cube/io/fits.py:63-172, load_fits_cube :171-260), which byte-swaps and scales on
// the host; here the raw big-endian bytes are uploaded as they are (pinned staging, see
// spectral_cube_amd/io_fits.py) and this kernel does astropy's conversion at HBM speed:
//   BITPIX -32 / -64 : byte swap (float64 rounded to float32), optional BSCALE/BZERO in the
//                      file's own precision;
//   BITPIX 8 / 16    : float32(raw) * float32(BSCALE) + float32(BZERO)   (astropy scales these
//   BITPIX 32 / 64   : float64(raw) * BSCALE + BZERO, rounded to float32  in float32 / float64)
//   BLANK (integer types) -> NaN.
#include "spc_common.h"
#include <algorithm>

namespace {

struct FitsArgs {
    const uint8_t* raw;
    float* out;
    int64_t n;
    int bitpix;
    int scaled;          // BSCALE != 1 or BZERO != 0
    double bscale, bzero;
    int has_blank;
    int64_t blank;
};

__device__ __forceinline__ float fits_one(const FitsArgs& A, int64_t i) {
    switch (A.bitpix) {
        case -32: {
            const uint32_t w = __builtin_bswap32(reinterpret_cast<const uint32_t*>(A.raw)[i]);
            float v = __uint_as_float(w);
            if (A.scaled) { v *= (float)A.bscale; v += (float)A.bzero; }
            return v;
        }
        case -64: {
            const uint64_t w = __builtin_bswap64(reinterpret_cast<const uint64_t*>(A.raw)[i]);
            double v = __longlong_as_double((long long)w);
            if (A.scaled) { v *= A.bscale; v += A.bzero; }
            return (float)v;
        }
        case 8: {
            const uint8_t r = A.raw[i];
            if (A.has_blank && (int64_t)r == A.blank) return NAN;
            float v = (float)r;
            if (A.scaled) { v *= (float)A.bscale; v += (float)A.bzero; }
            return v;
        }
        case 16: {
            const uint16_t w = reinterpret_cast<const uint16_t*>(A.raw)[i];
            const int16_t r = (int16_t)((w >> 8) | (w << 8));
            if (A.has_blank && (int64_t)r == A.blank) return NAN;
            float v = (float)r;
            if (A.scaled) { v *= (float)A.bscale; v += (float)A.bzero; }
            return v;
        }
        case 32: {
            const int32_t r = (int32_t)__builtin_bswap32(reinterpret_cast<const uint32_t*>(A.raw)[i]);
            if (A.has_blank && (int64_t)r == A.blank) return NAN;
            double v = (double)r;
            if (A.scaled) { v *= A.bscale; v += A.bzero; }
            return (float)v;
        }
        default: {   // 64
            const int64_t r = (int64_t)__builtin_bswap64(reinterpret_cast<const uint64_t*>(A.raw)[i]);
            if (A.has_blank && r == A.blank) return NAN;
            double v = (double)r;
            if (A.scaled) { v *= A.bscale; v += A.bzero; }
            return (float)v;
        }
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fits_to_f32_kernel(const FitsArgs A) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool fast32 = (A.bitpix == -32) && !A.scaled && ((((uintptr_t)A.raw) & 15) == 0) && ((((uintptr_t)A.out) & 15) == 0);
    if (fast32) {                             // the common case: 16 bytes in, 16 bytes out per lane
        const int64_t n4 = A.n / 4;
        for (int64_t i = t; i < n4; i += stride) {
            const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(A.raw) + i);
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = __uint_as_float(__builtin_bswap32(w[c]));
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(A.out) + i);
        }
        for (int64_t i = n4 * 4 + t; i < A.n; i += stride) A.out[i] = fits_one(A, i);
        return;
    }
    for (int64_t i = t; i < A.n; i += stride) A.out[i] = fits_one(A, i);
}

}  // namespace

extern "C" int spc_fits_to_f32(int device, void* stream, const void* d_raw, int bitpix, double bscale,
                               double bzero, int has_blank, int64_t blank, int64_t n, float* d_out) {
    SPC_REQUIRE(d_raw && d_out, "NULL pointer argument");
    SPC_REQUIRE(n >= 0, "negative sample count");
    SPC_REQUIRE(bitpix == 8 || bitpix == 16 || bitpix == 32 || bitpix == 64 || bitpix == -32 || bitpix == -64,
                "BITPIX must be one of 8, 16, 32, 64, -32, -64 (got %d)", bitpix);
    if (n == 0) return SPC_OK;
    SPC_DEVICE(device);
    FitsArgs A{};
    A.raw = (const uint8_t*)d_raw; A.out = d_out; A.n = n; A.bitpix = bitpix;
    A.bscale = bscale; A.bzero = bzero;
    A.scaled = (bscale != 1.0 || bzero != 0.0) ? 1 : 0;
    A.has_blank = (has_blank && bitpix > 0) ? 1 : 0;      // BLANK is only defined for integer images
    A.blank = blank;
    const int64_t per_block = 256 * 4 * 4;
    const unsigned nblocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, (n + per_block - 1) / per_block));
    hipLaunchKernelGGL(fits_to_f32_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

// ---- the wide sample types in their own precision: BITPIX = -64 / 32 / 64 -> float64 (what astropy hands the reference:
// float64 for -64, and raw * BSCALE + BZERO in float64 - or the integers, which np.result_type(dtype, 0.0) makes float64 in
// masks.py:225 - for 32 / 64).  Feeds spc_moments_f64.
namespace {
__global__ __launch_bounds__(256) void fits_to_f64_kernel(const FitsArgs A, double* out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < A.n; i += stride) {
        double v;
        if (A.bitpix == -64) {
            v = __longlong_as_double((long long)__builtin_bswap64(__builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(A.raw) + i)));
        } else {
            const int64_t r = A.bitpix == 32 ? (int64_t)(int32_t)__builtin_bswap32(reinterpret_cast<const uint32_t*>(A.raw)[i])
                                             : (int64_t)__builtin_bswap64(reinterpret_cast<const uint64_t*>(A.raw)[i]);
            if (A.has_blank && r == A.blank) { out[i] = nan; continue; }
            v = (double)r;
        }
        if (A.scaled) { v *= A.bscale; v += A.bzero; }
        out[i] = v;
    }
}
}  // namespace

extern "C" int spc_fits_to_f64(int device, void* stream, const void* d_raw, int bitpix, double bscale,
                               double bzero, int has_blank, int64_t blank, int64_t n, double* d_out) {
    SPC_REQUIRE(d_raw && d_out, "NULL pointer argument");
    SPC_REQUIRE(n >= 0, "negative sample count");
    SPC_REQUIRE(bitpix == 32 || bitpix == 64 || bitpix == -64, "BITPIX must be one of 32, 64, -64 (got %d)", bitpix);
    if (n == 0) return SPC_OK;
    SPC_DEVICE(device);
    FitsArgs A{};
    A.raw = (const uint8_t*)d_raw; A.out = nullptr; A.n = n; A.bitpix = bitpix;
    A.bscale = bscale; A.bzero = bzero;
    A.scaled = (bscale != 1.0 || bzero != 0.0) ? 1 : 0;
    A.has_blank = (has_blank && bitpix > 0) ? 1 : 0;
    A.blank = blank;
    const unsigned nblocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(8192, (n + 1023) / 1024));
    hipLaunchKernelGGL(fits_to_f64_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, A, d_out);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}


// out[i] *= factor over n contiguous floats (the Jy/beam rescaling of convolve_to,
// spectral_cube/dask_spectral_cube.py:1450-1457); NaNs stay NaNs.
namespace {
__global__ __launch_bounds__(256) void scale_f32_kernel(float* p, int64_t n, float f) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] *= f;
}
}  // namespace

extern "C" int spc_scale_f32(int device, void* stream, float* d_data, int64_t n, double factor) {
    SPC_REQUIRE(d_data != nullptr && n >= 0, "bad arguments");
    if (n == 0) return SPC_OK;
    SPC_DEVICE(device);
    const unsigned nblocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(16384, (n + 1023) / 1024));
    hipLaunchKernelGGL(scale_f32_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, d_data, n, (float)factor);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}
