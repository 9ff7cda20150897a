// spc_wide.h - what the float64 translation units share: the mask of a float64 cube on the device (thresholds kept in
// float64: numpy compares a float64 cube in float64, spectral_cube/masks.py:225) and the argument checks of spc_cube_f64.
#pragma once
#include "spc_common.h"

namespace {

struct MaskDev64 {
    uint32_t flags;
    double lo, hi;               // the thresholds as given
    const uint8_t* arr;
    int64_t row_stride, plane_stride;
    double clim, clo, chi;       // canonical form: |v| <= clim && !(v <= clo) && !(v >= chi)  (canonical64)
};

// the float64 twin of spc_canonical_pred (spc_common.h): three compares whatever the flags, NaN samples never pass; an absent bound
// is NaN, >= / <= become strict compares against the neighbouring double, a NaN threshold rejects everything
static inline void canonical64(uint32_t f, double thr_lo, double thr_hi, double* lim, double* lo, double* hi) {
    *lim = (f & SPC_MASK_FINITE) ? 1.7976931348623157e308 : INFINITY;
    *lo = NAN;
    *hi = NAN;
    if (f & (SPC_MASK_GT | SPC_MASK_GE)) {
        if (thr_lo != thr_lo) *lim = -1.0;
        else if (f & SPC_MASK_GT) *lo = thr_lo;
        else *lo = (thr_lo == -INFINITY) ? NAN : nextafter(thr_lo, -INFINITY);
    }
    if (f & (SPC_MASK_LT | SPC_MASK_LE)) {
        if (thr_hi != thr_hi) *lim = -1.0;
        else if (f & SPC_MASK_LT) *hi = thr_hi;
        else *hi = (thr_hi == INFINITY) ? NAN : nextafter(thr_hi, INFINITY);
    }
}

__device__ __forceinline__ bool pred64(const MaskDev64& m, double v) {      // mask predicate AND "not NaN"
    return (fabs(v) <= m.clim) & !(v <= m.clo) & !(v >= m.chi);
}

inline int check_cube64(const spc_cube_f64* c) {
    SPC_REQUIRE(c != nullptr && c->d_data != nullptr, "cube pointer is NULL");
    SPC_REQUIRE(c->nz > 0 && c->ny > 0 && c->nx > 0, "cube shape must be positive (got %lld,%lld,%lld)",
                (long long)c->nz, (long long)c->ny, (long long)c->nx);
    SPC_REQUIRE(c->row_stride >= c->nx, "row_stride %lld < nx %lld", (long long)c->row_stride, (long long)c->nx);
    SPC_REQUIRE(c->plane_stride >= c->row_stride * (c->ny - 1) + c->nx, "plane_stride too small");
    SPC_REQUIRE(c->nz < (1LL << 21), "nz too large for the float64 moment kernel (%lld)", (long long)c->nz);
    return SPC_OK;
}

// for kernels that only ever form z * plane_stride + y * row_stride + x: also a view whose first two axes are exchanged
inline int check_cube64_any_order(const spc_cube_f64* c) {
    SPC_REQUIRE(c != nullptr && c->d_data != nullptr, "cube pointer is NULL");
    SPC_REQUIRE(c->nz > 0 && c->ny > 0 && c->nx > 0, "cube shape must be positive (got %lld,%lld,%lld)",
                (long long)c->nz, (long long)c->ny, (long long)c->nx);
    SPC_REQUIRE(c->row_stride >= c->nx && c->plane_stride >= c->nx, "row / plane stride smaller than nx");
    SPC_REQUIRE(c->plane_stride >= c->row_stride * (c->ny - 1) + c->nx ||
                c->row_stride >= c->plane_stride * (c->nz - 1) + c->nx, "overlapping rows and planes");
    return SPC_OK;
}

inline int mask64_to_dev(const spc_mask_f64* m, const spc_cube_f64* c, MaskDev64* out) {
    out->flags = 0; out->lo = 0.0; out->hi = 0.0; out->arr = nullptr;
    out->row_stride = c->row_stride; out->plane_stride = c->plane_stride;
    canonical64(0u, 0.0, 0.0, &out->clim, &out->clo, &out->chi);
    if (!m) return SPC_OK;
    SPC_REQUIRE((m->flags & ~63u) == 0, "unknown mask flags 0x%x", m->flags);
    out->flags = m->flags; out->lo = m->thr_lo; out->hi = m->thr_hi;
    canonical64(m->flags, m->thr_lo, m->thr_hi, &out->clim, &out->clo, &out->chi);
    if (m->flags & SPC_MASK_ARRAY) {
        SPC_REQUIRE(m->d_array != nullptr, "SPC_MASK_ARRAY set but d_array is NULL");
        out->arr = m->d_array;
        if (m->row_stride) out->row_stride = m->row_stride;
        if (m->plane_stride) out->plane_stride = m->plane_stride;
    }
    return SPC_OK;
}

}  // namespace
