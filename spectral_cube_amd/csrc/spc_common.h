// Internal helpers shared by the HIP translation units of libspcube_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include "../../include/spcube_hip.h"

void spc_set_error(const char* fmt, ...);

#define SPC_HIP(call)                                                         \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            spc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                          __FILE__, __LINE__);                                \
            return SPC_ERR_HIP;                                               \
        }                                                                     \
    } while (0)

#define SPC_REQUIRE(cond, ...)                                                \
    do {                                                                      \
        if (!(cond)) {                                                        \
            spc_set_error(__VA_ARGS__);                                       \
            return SPC_ERR_INVALID;                                           \
        }                                                                     \
    } while (0)

#define SPC_LAUNCH_CHECK()                                                    \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) {                                               \
            spc_set_error("kernel launch failed: %s (%s:%d)",                 \
                          hipGetErrorString(e_), __FILE__, __LINE__);         \
            return SPC_ERR_HIP;                                               \
        }                                                                     \
    } while (0)

// RAII "current device" switch so concurrent host threads on different GPUs
// do not disturb each other (hipSetDevice is per-thread).
struct SpcDeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit SpcDeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~SpcDeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

#define SPC_DEVICE(dev)                                                       \
    SpcDeviceGuard guard_(dev);                                               \
    if (!guard_.ok) {                                                         \
        spc_set_error("hipSetDevice(%d) failed", dev);                        \
        return SPC_ERR_HIP;                                                   \
    }

// ---- scratch allocations ------------------------------------------------------------------
// Synchronous hipMalloc / hipFree on purpose: with the stream-ordered allocator (hipMallocAsync /
// hipFreeAsync) of this ROCm build, calls that alternate between kernel families intermittently
// computed on wrong data; hipFree also drains the device, which is
// what lets a scratch buffer be released right after the last kernel using it was queued.
static inline hipError_t spc_scratch_alloc(void** p, size_t bytes, hipStream_t) { return hipMalloc(p, bytes ? bytes : 1); }
static inline hipError_t spc_scratch_free(void* p, hipStream_t) { return p ? hipFree(p) : hipSuccess; }

// ---- tile flags handed from a speculative kernel to the kernel that redoes flagged tiles -----
// Producer and consumer are queued back to back on one stream; the kernel boundary orders them.
__device__ __forceinline__ void spc_flag_set(unsigned char* p) { *reinterpret_cast<volatile unsigned char*>(p) = 1; }
__device__ __forceinline__ unsigned char spc_flag_get(const unsigned char* p) {
    return *reinterpret_cast<const volatile unsigned char*>(p);
}
static inline hipError_t spc_flags_clear(unsigned char* d_flags, size_t n, hipStream_t st) {
    return n ? hipMemsetAsync(d_flags, 0, n, st) : hipSuccess;
}

// ---- device-side mask evaluation -----------------------------------------
struct MaskDev {
    uint32_t flags;
    float thr_lo, thr_hi;
    const uint8_t* arr;
    int64_t row_stride, plane_stride;
};

static inline int spc_mask_to_dev(const spc_mask* m, const spc_cube_f32* c, MaskDev* out) {
    out->flags = 0; out->thr_lo = 0.f; out->thr_hi = 0.f; out->arr = nullptr;
    out->row_stride = c->row_stride; out->plane_stride = c->plane_stride;
    if (!m) return SPC_OK;
    out->flags = m->flags; out->thr_lo = m->thr_lo; out->thr_hi = m->thr_hi;
    if (m->flags & SPC_MASK_ARRAY) {
        SPC_REQUIRE(m->d_array != nullptr, "SPC_MASK_ARRAY set but d_array is NULL");
        out->arr = m->d_array;
        if (m->row_stride) out->row_stride = m->row_stride;
        if (m->plane_stride) out->plane_stride = m->plane_stride;
    }
    SPC_REQUIRE((m->flags & ~63u) == 0, "unknown mask flags 0x%x", m->flags);
    return SPC_OK;
}

static inline int spc_check_cube(const spc_cube_f32* c) {
    SPC_REQUIRE(c != nullptr && c->d_data != nullptr, "cube pointer is NULL");
    SPC_REQUIRE(c->nz > 0 && c->ny > 0 && c->nx > 0, "cube shape must be positive (got %lld,%lld,%lld)",
                (long long)c->nz, (long long)c->ny, (long long)c->nx);
    SPC_REQUIRE(c->row_stride >= c->nx, "row_stride %lld < nx %lld", (long long)c->row_stride, (long long)c->nx);
    SPC_REQUIRE(c->plane_stride >= c->row_stride * (c->ny - 1) + c->nx, "plane_stride too small");
    return SPC_OK;
}

// for kernels that only ever form z * plane_stride + y * row_stride + x: also accepts a view whose
// first two axes are exchanged (row_stride > plane_stride), e.g. selection along y instead of z
static inline int spc_check_cube_any_order(const spc_cube_f32* c) {
    SPC_REQUIRE(c != nullptr && c->d_data != nullptr, "cube pointer is NULL");
    SPC_REQUIRE(c->nz > 0 && c->ny > 0 && c->nx > 0, "cube shape must be positive (got %lld,%lld,%lld)",
                (long long)c->nz, (long long)c->ny, (long long)c->nx);
    SPC_REQUIRE(c->row_stride >= c->nx && c->plane_stride >= c->nx, "row / plane stride smaller than nx");
    SPC_REQUIRE(c->plane_stride >= c->row_stride * (c->ny - 1) + c->nx ||
                c->row_stride >= c->plane_stride * (c->nz - 1) + c->nx, "overlapping rows and planes");
    return SPC_OK;
}

// predicate part of the mask (array part is handled by the caller's loads)
__device__ __forceinline__ bool spc_pred(uint32_t flags, float thr_lo, float thr_hi, float v) {
    bool inc = true;
    if (flags & SPC_MASK_FINITE) inc = inc && (fabsf(v) <= 3.402823466e+38f);  // false for NaN/inf
    if (flags & SPC_MASK_GT) inc = inc && (v > thr_lo);
    if (flags & SPC_MASK_GE) inc = inc && (v >= thr_lo);
    if (flags & SPC_MASK_LT) inc = inc && (v < thr_hi);
    if (flags & SPC_MASK_LE) inc = inc && (v <= thr_hi);
    return inc;
}
