// Single-pass statistics: count / min / max / sum / sum of squares of the included samples.
//
// spc_stats_global_f32 replaces the per-chunk compute_stats + aggregation of
// DaskSpectralCubeMixin.statistics (spectral_cube/dask_spectral_cube.py:769-814);
// spc_stats_axis_f32 replaces the per-axis nan-reductions behind sum / mean / std / max /
// min (dask_spectral_cube.py:641-767; NumPy class spectral_cube.py:578-791).  The reference
// makes one pass per statistic (and nanstd two); here every statistic of a call comes out of
// ONE read of the cube, accumulated in float64.
//
//   global : the cube is a linear stream - 16-byte loads, grid-stride, per-thread
//            accumulators, wave + LDS reduction, one partial record per block; the (few
//            thousand) partials are finished on the host.
//   axis 0 / 1 : "march" kernel - a lane owns 4 adjacent x, the 4 waves of a block split the
//            marched axis (z or y), LDS combine.  Same access pattern as the moment kernel.
//   axis 2 : one wave per (z, y) row, wave reduction.
#include "spc_common.h"
#include <algorithm>
#include <vector>

namespace {

struct Acc {                      // plain aggregate (lives in LDS too)
    double sum, ssq;
    float mn, mx;
    int cnt;
};
__device__ __forceinline__ Acc acc_zero() { return Acc{0.0, 0.0, INFINITY, -INFINITY, 0}; }

__device__ __forceinline__ void acc_add(Acc& a, float v, bool ok) {
    float w = ok ? v : 0.f;                  // select in float32 (one v_cndmask), then widen
    asm volatile("" : "+v"(w));
    const double d = (double)w;
    a.sum += d;
    a.ssq = fma(d, d, a.ssq);
    const float e = ok ? v : __builtin_nanf("");             // fminf / fmaxf return the other operand
    a.mn = fminf(a.mn, e);
    a.mx = fmaxf(a.mx, e);
    a.cnt += ok ? 1 : 0;
}

__device__ __forceinline__ void acc_merge(Acc& a, const Acc& b) {
    a.sum += b.sum; a.ssq += b.ssq;
    a.mn = fminf(a.mn, b.mn); a.mx = fmaxf(a.mx, b.mx);
    a.cnt += b.cnt;
}

__device__ __forceinline__ Acc acc_wave_reduce(Acc a) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        Acc b;
        b.sum = __shfl_xor(a.sum, d, 64); b.ssq = __shfl_xor(a.ssq, d, 64);
        b.mn = __shfl_xor(a.mn, d, 64); b.mx = __shfl_xor(a.mx, d, 64);
        b.cnt = __shfl_xor(a.cnt, d, 64);
        acc_merge(a, b);
    }
    return a;
}

__device__ __forceinline__ bool included(const MaskDev& m, float v, unsigned mk) {
    return spc_pred_valid(m, v) & (mk != 0);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct StatArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    // global
    int64_t nrows, rowlen;            // the cube as nrows rows of rowlen samples (1 row when contiguous)
    int64_t row_a, row_b;             // row r -> offset (r / ny) * row_a + (r % ny) * row_b  (r itself when 1 row)
    double* partial;                  // [gridDim.x][5]
    // axis
    int64_t n_outer, outer_stride, n_march, march_stride;
    int64_t m_outer_stride, m_march_stride;
    int32_t* o_cnt; float* o_min; float* o_max; double* o_sum; double* o_ssq;
};

// ---- whole cube -------------------------------------------------------------------------
// Round 4: a global reduction is free to read the cube LINEARLY (7.1 TB/s, tools/micro/copy_patterns.hip), so what it costs
// per sample decides the rate.  One accumulator set per vector component (four independent float64 chains instead of one
// serial chain of 16 dependent additions per iteration), the excluded sample becomes a NaN that v_min / v_max ignore (one
// select instead of two), the valid count is a scalar population count of the compare mask (per wave, not per lane), the
// predicate is the canonical three-compare form: ~13 instructions per sample against ~28.
struct Acc4 {
    double sum[4], ssq[4];
    float mn[4], mx[4];
};

template <bool ARR, bool THR>
__device__ __forceinline__ void acc4_add(Acc4& a, int& cnt, const MaskDev& mk, const f32x4& v, uint32_t m) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bool ok = THR ? spc_pred_valid(mk, v[c]) : (fabsf(v[c]) <= mk.lim);    // no threshold terms: one compare
        if (ARR) ok = ok & (((m >> (8 * c)) & 0xffu) != 0);
        cnt += __builtin_popcountll(__ballot(ok));          // wave-wide: every lane carries the same count
        float w = ok ? v[c] : 0.f;
        asm volatile("" : "+v"(w));                          // select before widening (one v_cndmask)
        const double d = (double)w;
        a.sum[c] += d;
        a.ssq[c] = fma(d, d, a.ssq[c]);
        const float e = ok ? v[c] : __builtin_nanf("");      // fminf / fmaxf return the other operand
        a.mn[c] = fminf(a.mn[c], e);
        a.mx[c] = fmaxf(a.mx[c], e);
    }
}

template <bool ARR, bool THR>
__global__ __launch_bounds__(256) void stats_global_kernel(const StatArgs A) {
    __shared__ double s_red[4][5];
    Acc a = acc_zero();
    Acc4 q;
#pragma unroll
    for (int c = 0; c < 4; ++c) { q.sum[c] = 0.0; q.ssq[c] = 0.0; q.mn[c] = INFINITY; q.mx[c] = -INFINITY; }
    int wave_cnt = 0;                                        // valid samples met by this WAVE in the 16-byte body
    const int t = threadIdx.x;
    constexpr int U = 8;
    // one contiguous run: every block strides through it.  Rows of a strided view (a strip of a cube): the blocks share the ROWS
    // out - round 3a walked them one after the other with all blocks striding inside each, which left a row of 1024 samples to
    // one block and the others idle (64 MB in 19 ms)
    const bool rows = A.nrows > 1;
    for (int64_t r = rows ? blockIdx.x : 0; r < A.nrows; r += rows ? gridDim.x : 1) {
        const int64_t off = !rows ? 0 : (r / A.ny) * A.row_a + (r % A.ny) * A.row_b;
        const int64_t moff = !rows ? 0 : (r / A.ny) * A.mask.plane_stride + (r % A.ny) * A.mask.row_stride;
        const float* p = A.cube + off;
        const uint8_t* pm = ARR ? A.mask.arr + moff : nullptr;
        const bool al = ((((uintptr_t)p) & 15) == 0) && (!ARR || ((((uintptr_t)pm) & 3) == 0));
        const int64_t n4 = al ? A.rowlen / 4 : 0;
        // 16-byte body: a block takes U consecutive chunks of 256 x 16 bytes (U x 4 KiB contiguous), then strides on
        const int64_t stride = rows ? 256 * U : (int64_t)gridDim.x * 256 * U;
        const int64_t first = rows ? 0 : (int64_t)blockIdx.x * 256 * U;
        int64_t i = first + t;
        for (; (i - t) + U * 256 <= n4; i += stride) {       // whole groups: block-uniform
            f32x4 v[U];
            uint32_t m[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i + u * 256);
                m[u] = ARR ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pm) + i + u * 256) : 0x01010101u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc4_add<ARR, THR>(q, wave_cnt, A.mask, v[u], m[u]);
        }
        // the last, partial group of chunks of this block
        for (int u = 0; u < U; ++u) {
            const int64_t k = i + u * 256;
            if (k - t + 255 < n4) {                          // whole chunk inside: wave-uniform, all lanes load
                const f32x4 v = reinterpret_cast<const f32x4*>(p)[k];
                const uint32_t m = ARR ? reinterpret_cast<const uint32_t*>(pm)[k] : 0x01010101u;
                acc4_add<ARR, THR>(q, wave_cnt, A.mask, v, m);
            } else if (k < n4) {                             // a ragged chunk: per-lane accumulators
                const f32x4 v = reinterpret_cast<const f32x4*>(p)[k];
                const uint32_t m = ARR ? reinterpret_cast<const uint32_t*>(pm)[k] : 0x01010101u;
#pragma unroll
                for (int c = 0; c < 4; ++c) acc_add(a, v[c], included(A.mask, v[c], (m >> (8 * c)) & 0xffu));
            }
        }
        // samples beyond the last whole 16 bytes (or the whole run when it is not aligned)
        for (int64_t j = n4 * 4 + (rows ? 0 : (int64_t)blockIdx.x * 256) + t; j < A.rowlen; j += rows ? 256 : (int64_t)gridDim.x * 256) {
            const float v = p[j];
            acc_add(a, v, included(A.mask, v, ARR ? pm[j] : 1u));
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        a.sum += q.sum[c]; a.ssq += q.ssq[c];
        a.mn = fminf(a.mn, q.mn[c]); a.mx = fmaxf(a.mx, q.mx[c]);
    }
    if ((t & 63) == 0) a.cnt += wave_cnt;                    // once per wave
    a = acc_wave_reduce(a);
    const int w = t >> 6;
    if ((t & 63) == 0) {
        s_red[w][0] = (double)a.cnt; s_red[w][1] = (double)a.mn; s_red[w][2] = (double)a.mx;
        s_red[w][3] = a.sum; s_red[w][4] = a.ssq;
    }
    __syncthreads();
    if (t == 0) {
        double* o = A.partial + (int64_t)blockIdx.x * 5;
        o[0] = s_red[0][0] + s_red[1][0] + s_red[2][0] + s_red[3][0];
        o[1] = fmin(fmin(s_red[0][1], s_red[1][1]), fmin(s_red[2][1], s_red[3][1]));
        o[2] = fmax(fmax(s_red[0][2], s_red[1][2]), fmax(s_red[2][2], s_red[3][2]));
        o[3] = (s_red[0][3] + s_red[1][3]) + (s_red[2][3] + s_red[3][3]);
        o[4] = (s_red[0][4] + s_red[1][4]) + (s_red[2][4] + s_red[3][4]);
    }
}

// ---- per channel: statistics of every (ny, nx) plane (sum / mean / ... with axis=(1, 2): spectra) -------
// grid = (segments, nz): a block streams its share of one plane, one partial record per block; the host adds
// the few segments of a plane.  Contiguous planes are read as one run of 16-byte loads, strided ones row by row.
template <bool ARR>
__global__ __launch_bounds__(256) void stats_planes_kernel(const StatArgs A) {
    __shared__ double s_red[4][5];
    Acc a = acc_zero();
    const int t = threadIdx.x;
    const int64_t z = blockIdx.y;
    const bool flat = (A.row_stride == A.nx) && (!ARR || A.mask.row_stride == A.nx);
    const int64_t nrows = flat ? 1 : A.ny, rowlen = flat ? A.ny * A.nx : A.nx;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t r = 0; r < nrows; ++r) {
        const float* p = A.cube + z * A.plane_stride + r * A.row_stride;
        const uint8_t* pm = ARR ? A.mask.arr + z * A.mask.plane_stride + r * A.mask.row_stride : nullptr;
        const bool al = ((((uintptr_t)p) & 15) == 0) && (!ARR || ((((uintptr_t)pm) & 3) == 0));
        const int64_t n4 = al ? rowlen / 4 : 0;
        constexpr int U = 4;
        int64_t i = (int64_t)blockIdx.x * 256 + t;
        for (; i + (U - 1) * stride < n4; i += U * stride) {
            f32x4 v[U];
            uint32_t m[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i + u * stride);
                m[u] = ARR ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pm) + i + u * stride) : 0x01010101u;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc_add(a, v[u][c], included(A.mask, v[u][c], (m[u] >> (8 * c)) & 0xffu));
        }
        for (; i < n4; i += stride) {
            const f32x4 v = reinterpret_cast<const f32x4*>(p)[i];
            const uint32_t m = ARR ? reinterpret_cast<const uint32_t*>(pm)[i] : 0x01010101u;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc_add(a, v[c], included(A.mask, v[c], (m >> (8 * c)) & 0xffu));
        }
        for (int64_t j = n4 * 4 + (int64_t)blockIdx.x * 256 + t; j < rowlen; j += stride) {
            const float v = p[j];
            acc_add(a, v, included(A.mask, v, ARR ? pm[j] : 1u));
        }
    }
    a = acc_wave_reduce(a);
    const int w = t >> 6;
    if ((t & 63) == 0) {
        s_red[w][0] = (double)a.cnt; s_red[w][1] = (double)a.mn; s_red[w][2] = (double)a.mx;
        s_red[w][3] = a.sum; s_red[w][4] = a.ssq;
    }
    __syncthreads();
    if (t == 0) {
        double* o = A.partial + (z * gridDim.x + blockIdx.x) * 5;
        o[0] = s_red[0][0] + s_red[1][0] + s_red[2][0] + s_red[3][0];
        o[1] = fmin(fmin(s_red[0][1], s_red[1][1]), fmin(s_red[2][1], s_red[3][1]));
        o[2] = fmax(fmax(s_red[0][2], s_red[1][2]), fmax(s_red[2][2], s_red[3][2]));
        o[3] = (s_red[0][3] + s_red[1][3]) + (s_red[2][3] + s_red[3][3]);
        o[4] = (s_red[0][4] + s_red[1][4]) + (s_red[2][4] + s_red[3][4]);
    }
}

__device__ __forceinline__ void write_outputs(const StatArgs& A, int64_t o, const Acc& a) {
    if (A.o_cnt) A.o_cnt[o] = a.cnt;
    if (A.o_min) A.o_min[o] = a.cnt ? a.mn : NAN;        // nanmin / nanmax of an all-NaN ray is NaN
    if (A.o_max) A.o_max[o] = a.cnt ? a.mx : NAN;
    if (A.o_sum) A.o_sum[o] = a.cnt ? a.sum : NAN;       // nansum_allbadtonan (dask_spectral_cube.py:54-59)
    if (A.o_ssq) A.o_ssq[o] = a.cnt ? a.ssq : NAN;
}

// ---- axis 0 (march z) and axis 1 (march y): lanes along x ---------------------------------
template <int VEC, bool ARR>
__global__ __launch_bounds__(256) void stats_march_kernel(const StatArgs A) {
    constexpr int ZW = 4, U = VEC == 4 ? 8 : 4;
    __shared__ Acc s_acc[ZW - 1][64][VEC];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t x = ((int64_t)blockIdx.x * 64 + lane) * VEC;
    const int64_t o = blockIdx.y;
    const bool live = x < A.nx;
    const int64_t xc = live ? x : 0;
    const float* p = A.cube + o * A.outer_stride + xc;
    const uint8_t* pm = ARR ? A.mask.arr + o * A.m_outer_stride + xc : nullptr;
    Acc a[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) a[c] = acc_zero();
    auto fetch = [&](int64_t k, float (&v)[VEC], unsigned (&mk)[VEC]) {
        if (VEC == 4) {
            const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + k * A.march_stride));
            const uint32_t m = ARR ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pm + k * A.m_march_stride)) : 0x01010101u;
#pragma unroll
            for (int c = 0; c < VEC; ++c) { v[c] = q[c]; mk[c] = (m >> (8 * c)) & 0xffu; }
        } else {
            v[0] = p[k * A.march_stride];
            mk[0] = ARR ? pm[k * A.m_march_stride] : 1u;
        }
    };
    int64_t k0 = w;
    for (; k0 + (int64_t)(U - 1) * ZW < A.n_march; k0 += ZW * U) {      // U planes (stride ZW) in flight per lane
        float v[U][VEC];
        unsigned mk[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(k0 + (int64_t)u * ZW, v[u], mk[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < VEC; ++c) acc_add(a[c], v[u][c], included(A.mask, v[u][c], mk[u][c]));
    }
    for (; k0 < A.n_march; k0 += ZW) {
        float v[VEC];
        unsigned mk[VEC];
        fetch(k0, v, mk);
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc_add(a[c], v[c], included(A.mask, v[c], mk[c]));
    }
    if (w > 0) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) s_acc[w - 1][lane][c] = a[c];
    }
    __syncthreads();
    if (w == 0 && live) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
#pragma unroll
            for (int k = 0; k < ZW - 1; ++k) acc_merge(a[c], s_acc[k][lane][c]);
            if (x + c < A.nx) write_outputs(A, o * A.nx + x + c, a[c]);
        }
    }
}

// ---- axis 2: one wave per (z, y) row ------------------------------------------------------
template <bool ARR>
__global__ __launch_bounds__(256) void stats_rows_kernel(const StatArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= A.nz * A.ny) return;
    const int64_t z = r / A.ny, y = r - z * A.ny;
    const float* p = A.cube + z * A.plane_stride + y * A.row_stride;
    const uint8_t* pm = ARR ? A.mask.arr + z * A.mask.plane_stride + y * A.mask.row_stride : nullptr;
    Acc a = acc_zero();
    const bool al = ((((uintptr_t)p) & 15) == 0) && (!ARR || ((((uintptr_t)pm) & 3) == 0));
    const int64_t n4 = al ? A.nx / 4 : 0;
    for (int64_t i = lane; i < n4; i += 64) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i);
        const uint32_t m = ARR ? reinterpret_cast<const uint32_t*>(pm)[i] : 0x01010101u;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc_add(a, v[c], included(A.mask, v[c], (m >> (8 * c)) & 0xffu));
    }
    for (int64_t j = n4 * 4 + lane; j < A.nx; j += 64) {
        const float v = p[j];
        acc_add(a, v, included(A.mask, v, ARR ? pm[j] : 1u));
    }
    a = acc_wave_reduce(a);
    if (lane == 0) write_outputs(A, r, a);
}

// ---- argmax / argmin along a spatial axis (a8: spectral_cube.py:793-819) -------------------
// nanargmax of the data filled with -inf (argmin: +inf): excluded and NaN samples take the fill,
// the FIRST index wins ties, a ray without included samples gives 0.  Axis 0 comes out of the
// fused moment kernel (spc_moments_f32); these serve axis 1 (march over y) and axis 2 (along the
// row), with the same loops as the statistics kernels above.
struct ArgAcc { float mn, mx; int imn, imx; };
__device__ __forceinline__ ArgAcc arg_zero() { return ArgAcc{INFINITY, -INFINITY, 0, 0}; }
__device__ __forceinline__ void arg_add(ArgAcc& a, float v, bool ok, int k) {   // k ascending per caller
    const bool up = ok && (v > a.mx), dn = ok && (v < a.mn);
    a.mx = up ? v : a.mx; a.imx = up ? k : a.imx;
    a.mn = dn ? v : a.mn; a.imn = dn ? k : a.imn;
}
__device__ __forceinline__ void arg_merge(ArgAcc& a, const ArgAcc& b) {
    const bool up = (b.mx > a.mx) || (b.mx == a.mx && b.imx < a.imx);
    const bool dn = (b.mn < a.mn) || (b.mn == a.mn && b.imn < a.imn);
    a.mx = up ? b.mx : a.mx; a.imx = up ? b.imx : a.imx;
    a.mn = dn ? b.mn : a.mn; a.imn = dn ? b.imn : a.imn;
}
struct ArgArgs {
    StatArgs s;
    int64_t* o_argmin;
    int64_t* o_argmax;
};

template <int VEC, bool ARR>
__global__ __launch_bounds__(256) void arg_march_kernel(const ArgArgs B) {
    const StatArgs& A = B.s;
    constexpr int ZW = 4, U = VEC == 4 ? 8 : 4;
    __shared__ ArgAcc s_acc[ZW - 1][64][VEC];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t x = ((int64_t)blockIdx.x * 64 + lane) * VEC;
    const int64_t o = blockIdx.y;
    const bool live = x < A.nx;
    const int64_t xc = live ? x : 0;
    const float* p = A.cube + o * A.outer_stride + xc;
    const uint8_t* pm = ARR ? A.mask.arr + o * A.m_outer_stride + xc : nullptr;
    ArgAcc a[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) a[c] = arg_zero();
    auto fetch = [&](int64_t k, float (&v)[VEC], unsigned (&mk)[VEC]) {
        if (VEC == 4) {
            const f32x4 q = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + k * A.march_stride));
            const uint32_t m = ARR ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pm + k * A.m_march_stride)) : 0x01010101u;
#pragma unroll
            for (int c = 0; c < VEC; ++c) { v[c] = q[c]; mk[c] = (m >> (8 * c)) & 0xffu; }
        } else {
            v[0] = p[k * A.march_stride];
            mk[0] = ARR ? pm[k * A.m_march_stride] : 1u;
        }
    };
    int64_t k0 = w;
    for (; k0 + (int64_t)(U - 1) * ZW < A.n_march; k0 += ZW * U) {      // U planes (stride ZW) in flight per lane
        float v[U][VEC];
        unsigned mk[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(k0 + (int64_t)u * ZW, v[u], mk[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < VEC; ++c) arg_add(a[c], v[u][c], included(A.mask, v[u][c], mk[u][c]), (int)(k0 + (int64_t)u * ZW));
    }
    for (; k0 < A.n_march; k0 += ZW) {
        float v[VEC];
        unsigned mk[VEC];
        fetch(k0, v, mk);
#pragma unroll
        for (int c = 0; c < VEC; ++c) arg_add(a[c], v[c], included(A.mask, v[c], mk[c]), (int)k0);
    }
    if (w > 0) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) s_acc[w - 1][lane][c] = a[c];
    }
    __syncthreads();
    if (w == 0 && live) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
#pragma unroll
            for (int k = 0; k < ZW - 1; ++k) arg_merge(a[c], s_acc[k][lane][c]);
            if (x + c < A.nx) {
                if (B.o_argmin) B.o_argmin[o * A.nx + x + c] = a[c].imn;
                if (B.o_argmax) B.o_argmax[o * A.nx + x + c] = a[c].imx;
            }
        }
    }
}

template <bool ARR>
__global__ __launch_bounds__(256) void arg_rows_kernel(const ArgArgs B) {
    const StatArgs& A = B.s;
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= A.nz * A.ny) return;
    const int64_t z = r / A.ny, y = r - z * A.ny;
    const float* p = A.cube + z * A.plane_stride + y * A.row_stride;
    const uint8_t* pm = ARR ? A.mask.arr + z * A.mask.plane_stride + y * A.mask.row_stride : nullptr;
    ArgAcc a = arg_zero();
    const bool al = ((((uintptr_t)p) & 15) == 0) && (!ARR || ((((uintptr_t)pm) & 3) == 0));
    const int64_t n4 = al ? A.nx / 4 : 0;
    // four 16-byte loads per lane in flight (a row is short: one wave per row lives on its load latency)
    constexpr int U = 4;
    for (int64_t i0 = lane; i0 < n4; i0 += 64 * U) {
        f32x4 v[U];
        uint32_t m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = min(i0 + 64 * u, n4 - 1);
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i);
            m[u] = ARR ? reinterpret_cast<const uint32_t*>(pm)[i] : 0x01010101u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + 64 * u;
            if (i < n4) {
#pragma unroll
                for (int c = 0; c < 4; ++c) arg_add(a, v[u][c], included(A.mask, v[u][c], (m[u] >> (8 * c)) & 0xffu), (int)(4 * i + c));
            }
        }
    }
    for (int64_t j = n4 * 4 + lane; j < A.nx; j += 64) {
        const float v = p[j];
        arg_add(a, v, included(A.mask, v, ARR ? pm[j] : 1u), (int)j);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ArgAcc b;
        b.mn = __shfl_xor(a.mn, d, 64); b.mx = __shfl_xor(a.mx, d, 64);
        b.imn = __shfl_xor(a.imn, d, 64); b.imx = __shfl_xor(a.imx, d, 64);
        arg_merge(a, b);
    }
    if (lane == 0) {
        if (B.o_argmin) B.o_argmin[r] = a.imn;
        if (B.o_argmax) B.o_argmax[r] = a.imx;
    }
}

// ---- 2-D convolution of ONE float64 map --------------------------------------------------
// For the algebraic spatial_smooth -> moment path (spectral_cube_amd/cube.py): with every voxel
// valid, spatial smoothing (astropy's NaN-free branch: zero fill outside, division by the kernel
// sum) commutes with the sums along z, so S_n' = conv2d(S_n) - three small maps instead of nz
// planes.  Direct, unoptimised on purpose: ny * nx * nky * nkx fp64 FMAs (2048^2 x 29^2 = 3.5
// GFMA, ~1 ms), inputs come from L2.  NaN inputs propagate (the caller falls back then).
struct MapConvArgs {
    const double* in;
    double* out;
    const double* k;           // (nky, nkx), already divided by its sum
    int64_t ny, nx;
    int nky, nkx;
};

__global__ __launch_bounds__(256) void map_conv2d_f64_kernel(const MapConvArgs A) {
    const int64_t x = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t y = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= A.nx || y >= A.ny) return;
    const int hy = A.nky / 2, hx = A.nkx / 2;
    double acc = 0.0;
    for (int jy = 0; jy < A.nky; ++jy) {
        const int64_t iy = y + hy - jy;                    // true convolution: kernel flipped
        if (iy < 0 || iy >= A.ny) continue;
        const double* row = A.in + iy * A.nx;
        const double* kr = A.k + (int64_t)jy * A.nkx;
        for (int jx = 0; jx < A.nkx; ++jx) {
            const int64_t ix = x + hx - jx;
            if (ix >= 0 && ix < A.nx) acc = fma(kr[jx], row[ix], acc);
        }
    }
    A.out[y * A.nx + x] = acc;
}

// The same convolution for a kernel that is an outer product ky (x) kx of at most 33 taps per axis (Gaussian2DKernel: exactly):
// a block stages a (16 + nky - 1) x (64 + nkx - 1) window in LDS (zeros outside the map), convolves its rows along x into a second
// LDS array, then the columns along y: nkx + nky (16 + nky - 1) / 16 multiply-adds per output instead of nky nkx (29 x 29: 109
// against 841; the 2048^2 map of the config-3 pipeline 1.64 -> ~0.2 ms).
constexpr int kMsTX = 64, kMsTY = 16, kMsMaxK = 33;
struct MapSepArgs {
    const double* in;
    double* out;
    const double* ky;          // nky taps, then nkx taps (ky[i] kx[j] = the normalised kernel)
    int64_t ny, nx;
    int nky, nkx;
};

__global__ __launch_bounds__(256) void map_conv2d_sep_f64_kernel(const MapSepArgs A) {
    __shared__ double stage[kMsTY + kMsMaxK - 1][kMsTX + kMsMaxK - 1];
    __shared__ double rows[kMsTY + kMsMaxK - 1][kMsTX];
    __shared__ double wy[kMsMaxK], wx[kMsMaxK];
    const int t = threadIdx.x;
    const int hy = A.nky / 2, hx = A.nkx / 2;
    const int64_t x0 = (int64_t)blockIdx.x * kMsTX, y0 = (int64_t)blockIdx.y * kMsTY;
    const int nr = kMsTY + 2 * hy, nc = kMsTX + 2 * hx;
    if (t < A.nky) wy[t] = A.ky[t];
    if (t >= 64 && t - 64 < A.nkx) wx[t - 64] = A.ky[A.nky + t - 64];
    for (int i = t; i < nr * nc; i += 256) {
        const int r = i / nc, c = i - r * nc;
        const int64_t gy = y0 - hy + r, gx = x0 - hx + c;
        stage[r][c] = (gy >= 0 && gy < A.ny && gx >= 0 && gx < A.nx) ? A.in[gy * A.nx + gx] : 0.0;
    }
    __syncthreads();
    // along x: rows[r][lx] = sum_jx kx[jx] in[r][x + hx - jx]  (staged column lx + 2 hx - jx)
    for (int i = t; i < nr * kMsTX; i += 256) {
        const int r = i / kMsTX, lx = i - r * kMsTX;
        double acc = 0.0;
        for (int jx = 0; jx < A.nkx; ++jx) acc = fma(wx[jx], stage[r][lx + 2 * hx - jx], acc);
        rows[r][lx] = acc;
    }
    __syncthreads();
    // along y: out[ly][lx] = sum_jy ky[jy] rows[ly + 2 hy - jy][lx]
    for (int i = t; i < kMsTY * kMsTX; i += 256) {
        const int ly = i / kMsTX, lx = i - ly * kMsTX;
        const int64_t gy = y0 + ly, gx = x0 + lx;
        if (gy >= A.ny || gx >= A.nx) continue;
        double acc = 0.0;
        for (int jy = 0; jy < A.nky; ++jy) acc = fma(wy[jy], rows[ly + 2 * hy - jy][lx], acc);
        A.out[gy * A.nx + gx] = acc;
    }
}

// ---- sigma clipping support: filled copy + clip pass --------
// astropy.stats.sigma_clip(axis=0, masked=False) behind DaskSpectralCubeMixin.
// sigma_clip_spectrally (spectral_cube/dask_spectral_cube.py:851-878) iterates
//   bounds = centre -/+ sigma * std per ray  ->  values outside become NaN
// until nothing changes (or maxiters).  The per-ray centre (median: spc_percentile_axis0_f32)
// and std (spc_stats_axis_f32) come from the kernels above; these two kernels are the
// elementwise ends: the NaN-filled working copy and the clip itself (with a change counter).
struct ClipArgs {
    const float* in;
    float* out;
    int64_t nz, ny, nx, row_stride, plane_stride, out_row_stride, out_plane_stride;
    MaskDev mask;
    float fill;
    const float* lo;          // (ny, nx)
    const float* hi;
    unsigned long long* nchanged;
};

template <bool ARR>
__global__ __launch_bounds__(256) void fill_masked_kernel(const ClipArgs A) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= A.nx) return;
    for (int64_t z = blockIdx.z; z < A.nz; z += gridDim.z) {
        const float v = A.in[z * A.plane_stride + y * A.row_stride + x];
        bool ok = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, v);
        if (ARR) ok = ok && A.mask.arr[z * A.mask.plane_stride + y * A.mask.row_stride + x] != 0;
        A.out[z * A.out_plane_stride + y * A.out_row_stride + x] = ok ? v : A.fill;
    }
}

// include map of a mask evaluated on the cube it is bound to: out[z][y][x] = included ? 1 : 0 (uint8,
// C-contiguous).  `nan_excluded`: a NaN sample counts as excluded as well (what ~isnan-style masks say).
template <bool ARR>
__global__ __launch_bounds__(256) void mask_include_kernel(const ClipArgs A, uint8_t* out, int nan_excluded) {
    const int64_t x = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t y = blockIdx.y;
    if (x >= A.nx) return;
    for (int64_t z = blockIdx.z; z < A.nz; z += gridDim.z) {
        const float v = A.in[z * A.plane_stride + y * A.row_stride + x];
        bool ok = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, v);
        if (ARR) ok = ok && A.mask.arr[z * A.mask.plane_stride + y * A.mask.row_stride + x] != 0;
        if (nan_excluded) ok = ok && (v == v);
        out[(z * A.ny + y) * A.nx + x] = ok ? 1 : 0;
    }
}

// filled copy with the two spatial axes exchanged: out[z][x][y] = included ? in[z][y][x] : fill.
// Puts the rays of an order statistic along x (median(axis=2)) along y, where the selection
// kernels stream them coalesced.  64 x 64 tile through LDS (pitch 65: conflict-free both ways),
// rows of the tile are read and written as 256-byte segments.
template <bool ARR>
__global__ __launch_bounds__(256) void fill_masked_transpose_kernel(const ClipArgs A) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t x0 = (int64_t)blockIdx.x * 64, y0 = (int64_t)blockIdx.y * 64;
    for (int64_t z = blockIdx.z; z < A.nz; z += gridDim.z) {
#pragma unroll 4
        for (int r = w; r < 64; r += 4) {
            const int64_t y = y0 + r, x = x0 + lane;
            float v = A.fill;
            if (y < A.ny && x < A.nx) {
                const float q = A.in[z * A.plane_stride + y * A.row_stride + x];
                bool ok = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, q);
                if (ARR) ok = ok && A.mask.arr[z * A.mask.plane_stride + y * A.mask.row_stride + x] != 0;
                v = ok ? q : A.fill;
            }
            tile[r][lane] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int r = w; r < 64; r += 4) {
            const int64_t x = x0 + r, y = y0 + lane;
            if (x < A.nx && y < A.ny) A.out[z * A.out_plane_stride + x * A.out_row_stride + y] = tile[lane][r];
        }
        __syncthreads();
    }
}

// bounds of one sigma-clipping iteration, per ray: [centre - sigma_lower * std, centre + sigma_upper * std]
// from the maps the other kernels left on the device (count / sum / sum of squares -> mean and std in
// float64 exactly as numpy would from the same sums; or a median map as centre and a mad_std map as
// spread), so that an iteration makes no host round trip besides the change counter.
struct BoundsArgs {
    const int32_t* cnt; const double* sum; const double* ssq;
    const float* center; const float* spread;
    double lo_s, hi_s;
    float* lo; float* hi;
    int64_t n;
};
__global__ __launch_bounds__(256) void clip_bounds_kernel(const BoundsArgs A) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    double mean = nan, sd = nan;
    if (A.cnt) {
        const int c = A.cnt[i];
        if (c > 0) {
            mean = A.sum[i] / (double)c;
            const double var = __dsub_rn(A.ssq[i] / (double)c, __dmul_rn(mean, mean));    // no FMA contraction: numpy rounds twice
            sd = (var != var) ? nan : sqrt(var > 0.0 ? var : 0.0);
        }
    }
    const double cen = A.center ? (double)A.center[i] : mean;
    if (A.spread) sd = (double)A.spread[i];
    A.lo[i] = (float)__dsub_rn(cen, __dmul_rn(A.lo_s, sd));
    A.hi[i] = (float)__dadd_rn(cen, __dmul_rn(A.hi_s, sd));
}

template <int VEC>
__global__ __launch_bounds__(256) void clip_outside_kernel(const ClipArgs A) {
    // lane = VEC adjacent x of one row, marching over its share of the planes (16-byte accesses when VEC == 4)
    const int64_t x = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    const int64_t y = blockIdx.y;
    unsigned long long mine = 0;
    if (x < A.nx) {
        float lo[VEC], hi[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) { lo[c] = A.lo[y * A.nx + x + c]; hi[c] = A.hi[y * A.nx + x + c]; }
        for (int64_t z = blockIdx.z; z < A.nz; z += gridDim.z) {
            float* q = A.out + z * A.out_plane_stride + y * A.out_row_stride + x;
            if (VEC == 4) {
                f32x4 v = *reinterpret_cast<const f32x4*>(q);
                bool any = false;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (v[c] < lo[c] || v[c] > hi[c]) { v[c] = NAN; ++mine; any = true; }   // NaN compares false
                if (any) *reinterpret_cast<f32x4*>(q) = v;
            } else {
                const float v = *q;
                if (v < lo[0] || v > hi[0]) { *q = NAN; ++mine; }
            }
        }
    }
    // one atomic per wave
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(A.nchanged, mine);
}

// the block records of stats_global_kernel -> one record, on the device (40 bytes go to the host instead of 80 KiB, and the host
// does not loop): thread t adds records t, t + 256, ... in that order, then a fixed tree - the same sum from launch to launch
__global__ __launch_bounds__(256) void stats_finish_kernel(const double* partial, int nblocks, double* out) {
    __shared__ double s[5][256];
    const int t = threadIdx.x;
    double cnt = 0.0, mn = INFINITY, mx = -INFINITY, sum = 0.0, ssq = 0.0;
    for (int b = t; b < nblocks; b += 256) {
        const double* r = partial + (int64_t)b * 5;
        cnt += r[0]; mn = fmin(mn, r[1]); mx = fmax(mx, r[2]); sum += r[3]; ssq += r[4];
    }
    s[0][t] = cnt; s[1][t] = mn; s[2][t] = mx; s[3][t] = sum; s[4][t] = ssq;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) {
            s[0][t] += s[0][t + w]; s[1][t] = fmin(s[1][t], s[1][t + w]); s[2][t] = fmax(s[2][t], s[2][t + w]);
            s[3][t] += s[3][t + w]; s[4][t] += s[4][t + w];
        }
        __syncthreads();
    }
    if (t < 5) out[t] = s[t][0];
}

int env_blocks() {
    const char* e = getenv("SPC_STATS_BLOCKS");              // tuning hook; 2048 = 8 blocks per CU
    const int v = e ? atoi(e) : 2048;
    return v > 0 ? std::min(v, 4095) : 2048;                 // (the workspace holds 4096 records: one is the finished one)
}

int fill_common(StatArgs& A, const spc_cube_f32* cube, const spc_mask* mask) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    return SPC_OK;
}

}  // namespace

extern "C" {

int spc_stats_global_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                         double* h_stats, void* d_workspace, size_t workspace_bytes) {
    SPC_REQUIRE(h_stats != nullptr, "h_stats is NULL");
    StatArgs A{};
    int rc = fill_common(A, cube, mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const bool contig = (A.row_stride == A.nx) && (A.plane_stride == A.ny * A.nx) &&
                        (!arr || (A.mask.row_stride == A.nx && A.mask.plane_stride == A.ny * A.nx));
    if (contig) { A.nrows = 1; A.rowlen = A.nz * A.ny * A.nx; A.row_a = 0; A.row_b = 0; }
    else { A.nrows = A.nz * A.ny; A.rowlen = A.nx; A.row_a = A.plane_stride; A.row_b = A.row_stride; }
    // enough blocks to fill the chip, few enough that the host-side finish is trivial
    const int64_t per_block = 256 * 4 * 8;                   // one group: 8 chunks of 256 x 16 bytes
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(env_blocks(), contig ? (A.rowlen + per_block - 1) / per_block : (A.nrows + 3) / 4));
    SpcWorkspace ws(d_workspace, workspace_bytes);
    SPC_WS_TAKE(d_partial, ws, double, 5 * (size_t)(nblocks + 1));
    A.partial = d_partial;
    const bool thr = (A.mask.flags & (SPC_MASK_GT | SPC_MASK_GE | SPC_MASK_LT | SPC_MASK_LE)) != 0;
    if (arr && thr) hipLaunchKernelGGL((stats_global_kernel<true, true>), dim3(nblocks), dim3(256), 0, st, A);
    else if (arr) hipLaunchKernelGGL((stats_global_kernel<true, false>), dim3(nblocks), dim3(256), 0, st, A);
    else if (thr) hipLaunchKernelGGL((stats_global_kernel<false, true>), dim3(nblocks), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((stats_global_kernel<false, false>), dim3(nblocks), dim3(256), 0, st, A);
    hipError_t e = hipGetLastError();
    double* d_fin = d_partial + 5 * (size_t)nblocks;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(stats_finish_kernel, dim3(1), dim3(256), 0, st, d_partial, nblocks, d_fin);
        e = hipGetLastError();
    }
    double h[5] = {0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d_fin, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);         // the record goes back to the host: wait for THIS stream
    SPC_HIP(e);
    const double cnt = h[0], mn = h[1], mx = h[2], sum = h[3], ssq = h[4];
    h_stats[0] = cnt;
    h_stats[1] = cnt > 0 ? mn : NAN;
    h_stats[2] = cnt > 0 ? mx : NAN;
    h_stats[3] = (double)sum;
    h_stats[4] = (double)ssq;
    return SPC_OK;
}

int spc_stats_planes_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask, double* h_stats,
                         void* d_workspace, size_t workspace_bytes) {
    SPC_REQUIRE(h_stats != nullptr, "h_stats is NULL");
    StatArgs A{};
    int rc = fill_common(A, cube, mask);
    if (rc) return rc;
    SPC_REQUIRE(A.nz <= 65535, "more than 65535 channels per call not supported (split the call)");
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    // enough blocks for the chip even with few channels; a segment should still have a few thousand samples
    const int64_t plane = A.ny * A.nx;
    int nseg = (int)std::max<int64_t>(1, std::min<int64_t>((2048 + A.nz - 1) / A.nz, (plane + 16383) / 16384));
    nseg = std::min(nseg, 64);
    const size_t nrec = (size_t)A.nz * nseg;
    SpcWorkspace ws(d_workspace, workspace_bytes);
    SPC_WS_TAKE(d_partial, ws, double, 5 * nrec);
    A.partial = d_partial;
    dim3 grid((unsigned)nseg, (unsigned)A.nz);
    if (arr) hipLaunchKernelGGL(stats_planes_kernel<true>, grid, dim3(256), 0, st, A);
    else hipLaunchKernelGGL(stats_planes_kernel<false>, grid, dim3(256), 0, st, A);
    hipError_t e = hipGetLastError();
    std::vector<double> h(5 * nrec);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_partial, sizeof(double) * 5 * nrec, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    SPC_HIP(e);
    for (int64_t z = 0; z < A.nz; ++z) {
        double cnt = 0.0, mn = INFINITY, mx = -INFINITY;
        long double sum = 0.0L, ssq = 0.0L;
        for (int sgm = 0; sgm < nseg; ++sgm) {
            const double* r = &h[((size_t)z * nseg + sgm) * 5];
            cnt += r[0]; mn = std::min(mn, r[1]); mx = std::max(mx, r[2]); sum += r[3]; ssq += r[4];
        }
        double* o = h_stats + z * 5;
        o[0] = cnt; o[1] = cnt > 0 ? mn : NAN; o[2] = cnt > 0 ? mx : NAN; o[3] = (double)sum; o[4] = (double)ssq;
    }
    return SPC_OK;
}

int spc_stats_axis_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask, int axis,
                       const spc_stats_outputs* out) {
    SPC_REQUIRE(out != nullptr, "outputs struct is NULL");
    SPC_REQUIRE(axis >= 0 && axis <= 2, "axis must be 0, 1 or 2");
    SPC_REQUIRE(out->d_count || out->d_min || out->d_max || out->d_sum || out->d_sumsq, "no output requested");
    StatArgs A{};
    int rc = fill_common(A, cube, mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    A.o_cnt = out->d_count; A.o_min = out->d_min; A.o_max = out->d_max; A.o_sum = out->d_sum; A.o_ssq = out->d_sumsq;
    if (axis == 2) {
        const int64_t rows = A.nz * A.ny;
        SPC_REQUIRE((rows + 3) / 4 <= 0x7fffffffLL, "too many rows for one launch");
        dim3 grid((unsigned)((rows + 3) / 4));
        if (arr) hipLaunchKernelGGL(stats_rows_kernel<true>, grid, dim3(256), 0, st, A);
        else hipLaunchKernelGGL(stats_rows_kernel<false>, grid, dim3(256), 0, st, A);
        SPC_LAUNCH_CHECK();
        return SPC_OK;
    }
    if (axis == 0) {
        A.n_outer = A.ny; A.outer_stride = A.row_stride; A.n_march = A.nz; A.march_stride = A.plane_stride;
        A.m_outer_stride = A.mask.row_stride; A.m_march_stride = A.mask.plane_stride;
    } else {
        A.n_outer = A.nz; A.outer_stride = A.plane_stride; A.n_march = A.ny; A.march_stride = A.row_stride;
        A.m_outer_stride = A.mask.plane_stride; A.m_march_stride = A.mask.row_stride;
    }
    SPC_REQUIRE(A.n_outer <= 65535, "more than 65535 output rows per call not supported (split the call)");
    const bool v4 = (A.nx % 4 == 0) && (A.row_stride % 4 == 0) && (A.plane_stride % 4 == 0) &&
                    ((((uintptr_t)A.cube) & 15) == 0) &&
                    (!arr || ((A.mask.row_stride % 4 == 0) && (A.mask.plane_stride % 4 == 0) && ((((uintptr_t)A.mask.arr) & 3) == 0)));
    if (v4) {
        dim3 grid((unsigned)((A.nx + 255) / 256), (unsigned)A.n_outer);
        if (arr) hipLaunchKernelGGL((stats_march_kernel<4, true>), grid, dim3(256), 0, st, A);
        else hipLaunchKernelGGL((stats_march_kernel<4, false>), grid, dim3(256), 0, st, A);
    } else {
        dim3 grid((unsigned)((A.nx + 63) / 64), (unsigned)A.n_outer);
        if (arr) hipLaunchKernelGGL((stats_march_kernel<1, true>), grid, dim3(256), 0, st, A);
        else hipLaunchKernelGGL((stats_march_kernel<1, false>), grid, dim3(256), 0, st, A);
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_argextrema_axis_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask, int axis,
                            int64_t* d_argmin, int64_t* d_argmax) {
    SPC_REQUIRE(axis == 1 || axis == 2, "axis must be 1 or 2 (axis 0 is an output of spc_moments_f32)");
    SPC_REQUIRE(d_argmin || d_argmax, "no output requested");
    ArgArgs B{};
    StatArgs& A = B.s;
    int rc = fill_common(A, cube, mask);
    if (rc) return rc;
    SPC_REQUIRE(A.nx < (1ll << 31) && A.ny < (1ll << 31), "axis longer than 2^31 samples");
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    B.o_argmin = d_argmin; B.o_argmax = d_argmax;
    if (axis == 2) {
        const int64_t rows = A.nz * A.ny;
        SPC_REQUIRE((rows + 3) / 4 <= 0x7fffffffLL, "too many rows for one launch");
        dim3 grid((unsigned)((rows + 3) / 4));
        if (arr) hipLaunchKernelGGL(arg_rows_kernel<true>, grid, dim3(256), 0, st, B);
        else hipLaunchKernelGGL(arg_rows_kernel<false>, grid, dim3(256), 0, st, B);
        SPC_LAUNCH_CHECK();
        return SPC_OK;
    }
    A.n_outer = A.nz; A.outer_stride = A.plane_stride; A.n_march = A.ny; A.march_stride = A.row_stride;
    A.m_outer_stride = A.mask.plane_stride; A.m_march_stride = A.mask.row_stride;
    SPC_REQUIRE(A.n_outer <= 65535, "more than 65535 output rows per call not supported (split the call)");
    const bool v4 = (A.nx % 4 == 0) && (A.row_stride % 4 == 0) && (A.plane_stride % 4 == 0) &&
                    ((((uintptr_t)A.cube) & 15) == 0) &&
                    (!arr || ((A.mask.row_stride % 4 == 0) && (A.mask.plane_stride % 4 == 0) && ((((uintptr_t)A.mask.arr) & 3) == 0)));
    if (v4) {
        dim3 grid((unsigned)((A.nx + 255) / 256), (unsigned)A.n_outer);
        if (arr) hipLaunchKernelGGL((arg_march_kernel<4, true>), grid, dim3(256), 0, st, B);
        else hipLaunchKernelGGL((arg_march_kernel<4, false>), grid, dim3(256), 0, st, B);
    } else {
        dim3 grid((unsigned)((A.nx + 63) / 64), (unsigned)A.n_outer);
        if (arr) hipLaunchKernelGGL((arg_march_kernel<1, true>), grid, dim3(256), 0, st, B);
        else hipLaunchKernelGGL((arg_march_kernel<1, false>), grid, dim3(256), 0, st, B);
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_map_conv2d_f64(int device, void* stream, const double* d_in, int64_t ny, int64_t nx,
                       const double* h_kernel, int nky, int nkx, double* d_out, void* d_workspace, size_t workspace_bytes) {
    SPC_REQUIRE(d_in && d_out && h_kernel, "NULL pointer argument");
    SPC_REQUIRE(ny > 0 && nx > 0 && nky > 0 && nkx > 0 && (nky & 1) && (nkx & 1), "bad map / kernel shape");
    double sum = 0.0;
    for (int i = 0; i < nky * nkx; ++i) sum += h_kernel[i];
    SPC_REQUIRE(!(sum < 1e-8 && sum > -1e-8) && sum >= 1e-8,
                "The kernel can't be normalized, because its sum is close to zero");
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    std::vector<double> k((size_t)nky * nkx);
    for (size_t i = 0; i < k.size(); ++i) k[i] = h_kernel[i] / sum;
    SpcWorkspace ws(d_workspace, workspace_bytes);
    SPC_WS_TAKE(d_k, ws, double, k.size());
    SPC_HIP(spc_table_upload(d_k, k.data(), sizeof(double) * k.size(), st));
    // an outer product of at most 33 x 33 taps (the centre row and column reproduce every tap to 1e-14 of the largest): two 1-D passes
    if (nky <= kMsMaxK && nkx <= kMsMaxK && nky * nkx >= nky + nkx && getenv("SPC_MAP_CONV_DIRECT") == nullptr) {
        const int hy = nky / 2, hx = nkx / 2;
        const double c = k[(size_t)hy * nkx + hx];
        double kmax = 0.0, dev = 0.0;
        for (double v : k) kmax = std::max(kmax, fabs(v));
        if (c != 0.0) {
            std::vector<double> f((size_t)nky + nkx);
            for (int i = 0; i < nky; ++i) f[i] = k[(size_t)i * nkx + hx] / c;
            for (int j = 0; j < nkx; ++j) f[nky + j] = k[(size_t)hy * nkx + j];
            for (int i = 0; i < nky; ++i)
                for (int j = 0; j < nkx; ++j) dev = std::max(dev, fabs(f[i] * f[nky + j] - k[(size_t)i * nkx + j]));
            if (dev <= 1e-14 * kmax) {
                SPC_HIP(spc_table_upload(d_k, f.data(), sizeof(double) * f.size(), st));
                MapSepArgs S{d_in, d_out, d_k, ny, nx, nky, nkx};
                hipLaunchKernelGGL(map_conv2d_sep_f64_kernel, dim3((unsigned)((nx + kMsTX - 1) / kMsTX), (unsigned)((ny + kMsTY - 1) / kMsTY)),
                                   dim3(256), 0, st, S);
                SPC_LAUNCH_CHECK();
                return SPC_OK;
            }
        }
    }
    MapConvArgs A{d_in, d_out, d_k, ny, nx, nky, nkx};
    hipLaunchKernelGGL(map_conv2d_f64_kernel, dim3((unsigned)((nx + 63) / 64), (unsigned)((ny + 3) / 4)), dim3(256), 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

// ---- elementwise arithmetic on float64 maps (the algebra around spc_map_conv2d_f64 without a trip to the host)
namespace {
__global__ __launch_bounds__(256) void map_check_kernel(const int32_t* counts, int32_t expect, const double* values, int64_t n, uint32_t* flags) {
    unsigned bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (counts && counts[i] != expect) bad |= 1u;
        if (values && !(fabs(values[i]) <= 1.7976931348623157e308)) bad |= 2u;
    }
    if (__any(bad & 1u) && (threadIdx.x & 63) == 0) atomicOr(flags, 1u);
    if (__any(bad & 2u) && (threadIdx.x & 63) == 0) atomicOr(flags, 2u);
}

__global__ __launch_bounds__(256) void map_arith_kernel(int op, const double* a, const double* b, const double* c, double s,
                                                        double* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b ? b[i] : 0.0, z = c ? c[i] : 0.0;
    double r;
    switch (op) {
        case SPC_MAP_MUL: r = x * y; break;
        case SPC_MAP_SECOND_MOMENT_SUM: r = (x + y * y) * z; break;       // (m2 + mu^2) * S0 = S2
        case SPC_MAP_DIV_ADD: r = x / y + s; break;                      // S1' / S0' + offset
        case SPC_MAP_DIV_SUB_SQ: r = x / y - z * z; break;               // S2' / S0' - mu'^2
        default: r = x; break;
    }
    out[i] = r;
}
}  // namespace

int spc_map_arith_f64(int device, void* stream, int op, const double* d_a, const double* d_b, const double* d_c, double s,
                      double* d_out, int64_t n) {
    SPC_REQUIRE(d_a && d_out && n >= 0, "NULL pointer argument");
    SPC_REQUIRE(op >= SPC_MAP_MUL && op <= SPC_MAP_DIV_SUB_SQ, "unknown map operation");
    SPC_REQUIRE(d_b != nullptr, "d_b is NULL");
    SPC_REQUIRE(d_c != nullptr || (op != SPC_MAP_SECOND_MOMENT_SUM && op != SPC_MAP_DIV_SUB_SQ), "d_c is NULL");
    if (n == 0) return SPC_OK;
    SPC_DEVICE(device);
    hipLaunchKernelGGL(map_arith_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, d_a, d_b, d_c, s, d_out, n);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_map_check(int device, void* stream, const int32_t* d_counts, int32_t expect, const double* d_values, int64_t n,
                  uint32_t* d_flags) {
    SPC_REQUIRE(d_flags && n >= 0, "NULL pointer argument");
    SPC_DEVICE(device);
    SPC_HIP(hipMemsetAsync(d_flags, 0, sizeof(uint32_t), (hipStream_t)stream));
    if (n == 0 || (!d_counts && !d_values)) return SPC_OK;
    const unsigned nb = (unsigned)std::min<int64_t>((n + 1023) / 1024, 4096);
    hipLaunchKernelGGL(map_check_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, d_counts, expect, d_values, n, d_flags);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_fill_masked_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask, float fill,
                        float* d_out, int64_t out_row_stride, int64_t out_plane_stride) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    ClipArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_REQUIRE(cube->ny <= 65535, "more than 65535 rows per call not supported (split the call)");
    SPC_DEVICE(device);
    A.in = cube->d_data; A.out = d_out; A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.out_row_stride = out_row_stride ? out_row_stride : cube->nx;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : cube->ny * A.out_row_stride;
    A.fill = fill;
    dim3 grid((unsigned)((cube->nx + 255) / 256), (unsigned)cube->ny, (unsigned)std::min<int64_t>(cube->nz, 64));
    if (A.mask.flags & SPC_MASK_ARRAY) hipLaunchKernelGGL(fill_masked_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(fill_masked_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_mask_include_u8(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                        int nan_excluded, uint8_t* d_out) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    ClipArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_REQUIRE(cube->ny <= 65535, "too many rows for one launch");
    SPC_DEVICE(device);
    A.in = cube->d_data; A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    dim3 grid((unsigned)((cube->nx + 255) / 256), (unsigned)cube->ny, (unsigned)std::min<int64_t>(cube->nz, 64));
    if (A.mask.flags & SPC_MASK_ARRAY) hipLaunchKernelGGL(mask_include_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, A, d_out, nan_excluded);
    else hipLaunchKernelGGL(mask_include_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, A, d_out, nan_excluded);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_fill_masked_transpose_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                  float fill, float* d_out) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    ClipArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_REQUIRE((cube->ny + 63) / 64 <= 65535, "too many rows for one launch");
    SPC_DEVICE(device);
    A.in = cube->d_data; A.out = d_out; A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.out_row_stride = cube->ny;                      // out is (nz, nx, ny), C-contiguous
    A.out_plane_stride = cube->nx * cube->ny;
    A.fill = fill;
    dim3 grid((unsigned)((cube->nx + 63) / 64), (unsigned)((cube->ny + 63) / 64), (unsigned)std::min<int64_t>(cube->nz, 1024));
    if (A.mask.flags & SPC_MASK_ARRAY) hipLaunchKernelGGL(fill_masked_transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(fill_masked_transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_clip_bounds_f32(int device, void* stream, int64_t n, const int32_t* d_count, const double* d_sum,
                        const double* d_sumsq, const float* d_center, const float* d_spread,
                        double sigma_lower, double sigma_upper, float* d_lo, float* d_hi) {
    SPC_REQUIRE(n > 0 && d_lo && d_hi, "bad arguments");
    SPC_REQUIRE((d_count && d_sum && d_sumsq) || (d_center && d_spread), "need the statistics maps or centre + spread maps");
    SPC_REQUIRE(!d_count || (d_sum && d_sumsq), "count without sum / sumsq");
    SPC_DEVICE(device);
    BoundsArgs A{d_count, d_sum, d_sumsq, d_center, d_spread, sigma_lower, sigma_upper, d_lo, d_hi, n};
    hipLaunchKernelGGL(clip_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_clip_outside_f32(int device, void* stream, float* d_cube, int64_t nz, int64_t ny, int64_t nx,
                         const float* d_lo, const float* d_hi, uint64_t* h_nchanged, void* d_workspace, size_t workspace_bytes) {
    SPC_REQUIRE(d_cube && d_lo && d_hi && h_nchanged, "NULL pointer argument");
    SPC_REQUIRE(nz > 0 && ny > 0 && nx > 0 && ny <= 65535, "bad shape");
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    SpcWorkspace ws(d_workspace, workspace_bytes);
    SPC_WS_TAKE(d_n, ws, unsigned long long, 1);
    hipError_t e = hipMemsetAsync(d_n, 0, sizeof(unsigned long long), st);
    if (e == hipSuccess) {
        ClipArgs A{};
        A.out = d_cube; A.nz = nz; A.ny = ny; A.nx = nx; A.out_row_stride = nx; A.out_plane_stride = ny * nx;
        A.lo = d_lo; A.hi = d_hi; A.nchanged = d_n;
        if (nx % 4 == 0 && ((((uintptr_t)d_cube) & 15) == 0)) {
            dim3 grid((unsigned)((nx / 4 + 255) / 256), (unsigned)ny, (unsigned)std::min<int64_t>(nz, 64));
            hipLaunchKernelGGL(clip_outside_kernel<4>, grid, dim3(256), 0, st, A);
        } else {
            dim3 grid((unsigned)((nx + 255) / 256), (unsigned)ny, (unsigned)std::min<int64_t>(nz, 64));
            hipLaunchKernelGGL(clip_outside_kernel<1>, grid, dim3(256), 0, st, A);
        }
        e = hipGetLastError();
    }
    unsigned long long h = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d_n, sizeof h, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);         // the count goes back to the host
    SPC_HIP(e);
    *h_nchanged = (uint64_t)h;
    return SPC_OK;
}

}  // extern "C"

size_t spc_ws_stats(int kind, int64_t nz, int64_t ny, int64_t nx, int64_t p0, int64_t p1) {
    switch (kind) {
        case SPC_WS_STATS_GLOBAL: return spc_ws_round(sizeof(double) * 5 * 4096) + 256;          // <= 4096 partial records
        case SPC_WS_STATS_PLANES: return spc_ws_round(sizeof(double) * 5 * 64 * (size_t)nz) + 256; // <= 64 segments per channel
        case SPC_WS_MAP_CONV2D: return spc_ws_round(sizeof(double) * (size_t)(p0 * p1)) + 256;     // normalised taps
        case SPC_WS_CLIP_OUTSIDE: return 512;                                                   // one counter
    }
    return 0;
}
