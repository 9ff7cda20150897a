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

// ---- caller-owned device scratch -------------------------------------------------------------
// No entry point allocates, frees or drains: whatever scratch a call needs (tile flags, partial
// records, weight tables, intermediate (num, den) planes) is carved out of the d_workspace the
// caller passes in - spc_workspace_bytes() says how much - and everything is queued on the
// caller's stream, so calls on different streams overlap and a stream of calls never blocks the
// host except where a result has to come back to it (documented per entry point).
struct SpcWorkspace {
    char* base;
    size_t size, used;
    SpcWorkspace(void* p, size_t n) : base((char*)p), size(p ? n : 0), used(0) {}
    // 256-byte aligned slices; nullptr when the workspace is too small
    void* take(size_t bytes) {
        const size_t at = (used + 255) & ~(size_t)255;
        if (!base || at + bytes > size) { used = at + bytes; return nullptr; }
        used = at + bytes;
        return base + at;
    }
};
static inline size_t spc_ws_round(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

#define SPC_WS_TAKE(var, ws, type, count)                                                          \
    type* var = (type*)(ws).take(sizeof(type) * (size_t)(count));                                   \
    if (!var) {                                                                                    \
        spc_set_error("d_workspace too small: this call needs at least %zu bytes (spc_workspace_bytes)", \
                      (ws).used);                                                                  \
        return SPC_ERR_INVALID;                                                                    \
    }

// A small host table (kernel taps, per-channel weights) -> device memory by kernel launches only:
// the bytes travel as kernel arguments, so the host buffer may die when the call returns and nothing
// synchronises (a hipMemcpyAsync from pageable memory would need the buffer kept alive or a wait).
hipError_t spc_table_upload(void* d_dst, const void* h_src, size_t bytes, hipStream_t st);

// per-entry-point scratch sizes (defined next to the entry points, dispatched by spc_workspace_bytes)
size_t spc_ws_spectral_conv(int64_t nz, int64_t ny, int64_t nx, int64_t ntaps, bool fused);
size_t spc_ws_spatial_conv_sep(int64_t nz, int64_t ny, int64_t nx, int64_t nky, int64_t nkx);
size_t spc_ws_spatial_conv2d(int64_t nz, int64_t ny, int64_t nx, int64_t nky, int64_t nkx);
size_t spc_ws_resample_bilinear(int64_t ny_out, int64_t nx_out);
size_t spc_ws_resample_bilinear_lerp(int64_t nz, int64_t ny_out, int64_t nx_out);
size_t spc_ws_stats(int kind, int64_t nz, int64_t ny, int64_t nx, int64_t p0, int64_t p1);
size_t spc_ws_percentile_global(void);
size_t spc_ws_sigma_clip(void);
size_t spc_ws_wide(int kind, int64_t nz, int64_t ny, int64_t nx, int64_t p0, int64_t p1);
size_t spc_ws_spatial_conv_mfma(int64_t nz, int64_t ny, int64_t nx, int64_t nsum);
int spc_spatial_conv_split_store(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask, const double* h_ky, int nky,
                                 const double* h_kx, int nkx, float* d_out, int64_t out_row_stride, int64_t out_plane_stride);

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
// mask predicate -> |v| <= lim && !(v <= lo) && !(v >= hi): three compares whatever the flags.  An absent bound is
// NaN (the negated compare is then true for every v), >= / <= become strict compares against the neighbouring float,
// a NaN threshold rejects everything (numpy: x > nan is False); lim = FLT_MAX under isfinite, +inf otherwise (NaN
// samples fail |v| <= lim either way: they are never valid).
static inline void spc_canonical_pred(uint32_t f, float thr_lo, float thr_hi, float* lim, float* lo, float* hi) {
    *lim = (f & SPC_MASK_FINITE) ? 3.402823466e+38f : INFINITY;
    *lo = NAN;
    *hi = NAN;
    if (f & (SPC_MASK_GT | SPC_MASK_GE)) {
        if (thr_lo != thr_lo) *lim = -1.f;
        else if (f & SPC_MASK_GT) *lo = thr_lo;
        else *lo = (thr_lo == -INFINITY) ? NAN : nextafterf(thr_lo, -INFINITY);
    }
    if (f & (SPC_MASK_LT | SPC_MASK_LE)) {
        if (thr_hi != thr_hi) *lim = -1.f;
        else if (f & SPC_MASK_LT) *hi = thr_hi;
        else *hi = (thr_hi == INFINITY) ? NAN : nextafterf(thr_hi, INFINITY);
    }
}

struct MaskDev {
    uint32_t flags;
    float thr_lo, thr_hi;
    const uint8_t* arr;
    int64_t row_stride, plane_stride;
    float lim, lo, hi;           // the predicate terms in canonical form (spc_canonical_pred), set by spc_mask_to_dev
};

static inline int spc_mask_to_dev(const spc_mask* m, const spc_cube_f32* c, MaskDev* out) {
    out->flags = 0; out->thr_lo = 0.f; out->thr_hi = 0.f; out->arr = nullptr;
    out->row_stride = c->row_stride; out->plane_stride = c->plane_stride;
    spc_canonical_pred(0u, 0.f, 0.f, &out->lim, &out->lo, &out->hi);
    if (!m) return SPC_OK;
    out->flags = m->flags; out->thr_lo = m->thr_lo; out->thr_hi = m->thr_hi;
    spc_canonical_pred(m->flags, m->thr_lo, m->thr_hi, &out->lim, &out->lo, &out->hi);
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

// Buffer descriptor over (up to 4 GiB from) a wave-uniform base.  The base must be PROVABLY uniform, otherwise hipcc wraps
// every buffer op in a waterfall loop: both halves go through readfirstlane.
__device__ __forceinline__ auto spc_plane_srd(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)0xffffffffu, 0x00020000);
}

// predicate part of the mask AND "not NaN", in the canonical form: three compares and no flag tests (spc_pred below costs five
// flag-guarded compares and ~11 scalar instructions per sample whatever the mask: round 4 found the reductions bound by that)
__device__ __forceinline__ bool spc_pred_valid(const MaskDev& m, float v) {
    return (fabsf(v) <= m.lim) & !(v <= m.lo) & !(v >= m.hi);
}

// predicate part of the mask (array part is handled by the caller's loads); a NaN sample passes unless isfinite is asked for
__device__ __forceinline__ bool spc_pred(uint32_t flags, float thr_lo, float thr_hi, float v) {
    bool inc = true;
    if (flags & SPC_MASK_FINITE) inc = inc && (fabsf(v) <= 3.402823466e+38f);  // false for NaN/inf
    if (flags & SPC_MASK_GT) inc = inc && (v > thr_lo);
    if (flags & SPC_MASK_GE) inc = inc && (v >= thr_lo);
    if (flags & SPC_MASK_LT) inc = inc && (v < thr_hi);
    if (flags & SPC_MASK_LE) inc = inc && (v <= thr_hi);
    return inc;
}
