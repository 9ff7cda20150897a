// spc_wide_ops.hip - the operators next to the moments for float64 cubes (gfx950).
//
// The reference keeps a float64 source in float64 through every operator (np.result_type(dtype, 0.0),
// spectral_cube/masks.py:225; the Dask class keeps the chunk dtype, dask_spectral_cube.py:829): statistics() and the
// nan-reductions (dask_spectral_cube.py:641-814), spectral_smooth (:880-917), spatial_smooth (:962-993),
// spectral_interpolate (:1342-1353).  Rounds 1 - 4 narrowed such cubes to float32 on their way to HBM for everything but
// the spectral moments (spc_moments_f64.hip).  These kernels read and write the samples as they are: 8 bytes per voxel,
// thresholds compared in float64, sums in float64 - the arithmetic of the float32 kernels (astropy's convolve: float64
// top / bot, one division; scipy's interp1d slope form) without the final rounding to float32.
//
// They are plain HBM streams (lanes along x) and NOT tuned like the float32 stencils: float64 cubes are outside
// BASELINE.json's configurations; what matters here is that a float64 cube gives the reference's float64 values.
#include "spc_common.h"
#include "spc_wide.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

namespace {

struct Cube64 {
    const double* p;
    int64_t nz, ny, nx, row_stride, plane_stride;
};

__device__ __forceinline__ bool inc64(const Cube64& c, const MaskDev64& m, int64_t z, int64_t y, int64_t x, double& v) {
    v = c.p[z * c.plane_stride + y * c.row_stride + x];
    bool ok = pred64(m, v);
    if (m.flags & SPC_MASK_ARRAY) ok = ok && m.arr[z * m.plane_stride + y * m.row_stride + x] != 0;
    return ok;
}

// ---- statistics ---------------------------------------------------------------------------------------------
struct Rec64 { double n, mn, mx, s, q; };
__device__ __forceinline__ void rec_add(Rec64& r, double v) {
    r.n += 1.0; r.mn = fmin(r.mn, v); r.mx = fmax(r.mx, v); r.s += v; r.q = fma(v, v, r.q);
}
__device__ __forceinline__ void rec_merge(Rec64& a, const Rec64& b) {
    a.n += b.n; a.mn = fmin(a.mn, b.mn); a.mx = fmax(a.mx, b.mx); a.s += b.s; a.q += b.q;
}
__device__ __forceinline__ Rec64 rec_empty() { return Rec64{0.0, INFINITY, -INFINITY, 0.0, 0.0}; }
__device__ __forceinline__ Rec64 rec_shfl_down(const Rec64& r, int d) {
    return Rec64{__shfl_down(r.n, d, 64), __shfl_down(r.mn, d, 64), __shfl_down(r.mx, d, 64), __shfl_down(r.s, d, 64), __shfl_down(r.q, d, 64)};
}
// block of 256: lane partials -> one record in thread 0 (fixed order: the result does not depend on scheduling)
__device__ __forceinline__ Rec64 rec_block_reduce(Rec64 r, Rec64* sh) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const Rec64 o = rec_shfl_down(r, d); rec_merge(r, o); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[w] = r;
    __syncthreads();
    if (threadIdx.x == 0) for (int k = 1; k < (int)(blockDim.x >> 6); ++k) rec_merge(r, sh[k]);
    return r;
}

// rows of the cube (nz * ny of them) dealt to the blocks round robin; partial[b] = the block's record
// VEC = 2: rows of an even length on 16-byte boundaries - a lane asks for two adjacent samples (16 bytes, and 2 mask bytes) per
// request, four requests together: a wave moves 1 KiB per instruction where the one-sample form moves 512 bytes
template <int VEC>
__global__ __launch_bounds__(256) void stats64_global_kernel(const Cube64 C, const MaskDev64 M, Rec64* partial) {
    __shared__ Rec64 sh[4];
    Rec64 r = rec_empty();
    const int64_t nrows = C.nz * C.ny;
    const bool arr = (M.flags & SPC_MASK_ARRAY) != 0;
    constexpr int kIn = 4;                                       // requests together per lane (clamped into the row)
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    for (int64_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int64_t z = row / C.ny, y = row - z * C.ny;
        const double* pd = C.p + z * C.plane_stride + y * C.row_stride;
        const uint8_t* pm = arr ? M.arr + z * M.plane_stride + y * M.row_stride : nullptr;
        for (int64_t x0 = (int64_t)threadIdx.x * VEC; x0 < C.nx; x0 += (int64_t)blockDim.x * kIn * VEC) {
            double vv[kIn][VEC];
            unsigned mk[kIn];
#pragma unroll
            for (int q = 0; q < kIn; ++q) {
                const int64_t xc = min(x0 + (int64_t)q * blockDim.x * VEC, C.nx - VEC);
                if (VEC == 2) {
                    const f64x2 t = *reinterpret_cast<const f64x2*>(pd + xc);
                    vv[q][0] = t.x; vv[q][VEC - 1] = t.y;
                    mk[q] = arr ? (unsigned)*reinterpret_cast<const uint16_t*>(pm + xc) : 0x0101u;
                } else {
                    vv[q][0] = pd[xc];
                    mk[q] = arr ? pm[xc] : 1u;
                }
            }
#pragma unroll
            for (int q = 0; q < kIn; ++q) {
                const bool in = x0 + (int64_t)q * blockDim.x * VEC < C.nx;      // (VEC = 2: nx is even, both samples or neither)
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (in && pred64(M, vv[q][e]) && ((mk[q] >> (8 * e)) & 0xffu) != 0u) rec_add(r, vv[q][e]);
            }
        }
    }
    r = rec_block_reduce(r, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ __launch_bounds__(256) void stats64_finish_kernel(const Rec64* partial, int nblocks, double* out) {
    __shared__ Rec64 sh[4];
    Rec64 r = rec_empty();
    for (int k = threadIdx.x; k < nblocks; k += blockDim.x) rec_merge(r, partial[k]);
    r = rec_block_reduce(r, sh);
    if (threadIdx.x == 0) {
        out[0] = r.n; out[1] = r.n > 0 ? r.mn : NAN; out[2] = r.n > 0 ? r.mx : NAN; out[3] = r.s; out[4] = r.q;
    }
}

struct StatOut64 {
    int32_t* count; double* mn; double* mx; double* sum; double* sumsq;
};
__device__ __forceinline__ void stat_store(const StatOut64& O, int64_t o, const Rec64& r) {
    const bool any = r.n > 0;
    if (O.count) O.count[o] = (int32_t)r.n;
    if (O.mn) O.mn[o] = any ? r.mn : NAN;
    if (O.mx) O.mx[o] = any ? r.mx : NAN;
    if (O.sum) O.sum[o] = any ? r.s : NAN;              // nansum_allbadtonan (dask_spectral_cube.py:54-59)
    if (O.sumsq) O.sumsq[o] = any ? r.q : NAN;
}
// AXIS 0: a lane per (y, x), marching z; AXIS 1: a lane per (z, x), marching y - lanes along x either way
template <int AXIS>
__global__ __launch_bounds__(256) void stats64_march_kernel(const Cube64 C, const MaskDev64 M, const StatOut64 O) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nouter = AXIS == 0 ? C.ny : C.nz;
    if (g >= nouter * C.nx) return;
    const int64_t a = g / C.nx, x = g - a * C.nx;
    const int64_t n = AXIS == 0 ? C.nz : C.ny;
    Rec64 r = rec_empty();
    const bool arr = (M.flags & SPC_MASK_ARRAY) != 0;
    const int64_t dstep = AXIS == 0 ? C.plane_stride : C.row_stride, mstep = AXIS == 0 ? M.plane_stride : M.row_stride;
    const double* pd = C.p + (AXIS == 0 ? a * C.row_stride : a * C.plane_stride) + x;
    const uint8_t* pm = arr ? M.arr + (AXIS == 0 ? a * M.row_stride : a * M.plane_stride) + x : nullptr;
    constexpr int kIn = 8;                                       // samples requested together per lane (clamped into the ray)
    for (int64_t k0 = 0; k0 < n; k0 += kIn) {
        double vv[kIn];
        unsigned mk[kIn];
#pragma unroll
        for (int q = 0; q < kIn; ++q) {
            const int64_t kc = min(k0 + q, n - 1);
            vv[q] = pd[kc * dstep];
            mk[q] = arr ? pm[kc * mstep] : 1u;
        }
#pragma unroll
        for (int q = 0; q < kIn; ++q)
            if (k0 + q < n && pred64(M, vv[q]) && mk[q] != 0u) rec_add(r, vv[q]);
    }
    stat_store(O, g, r);
}
// AXIS 2: a wave per row (z, y)
__global__ __launch_bounds__(256) void stats64_rows_kernel(const Cube64 C, const MaskDev64 M, const StatOut64 O) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= C.nz * C.ny) return;
    const int64_t z = row / C.ny, y = row - z * C.ny;
    Rec64 r = rec_empty();
    for (int64_t x = threadIdx.x & 63; x < C.nx; x += 64) {
        double v;
        if (inc64(C, M, z, y, x, v)) rec_add(r, v);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const Rec64 o = rec_shfl_down(r, d); rec_merge(r, o); }
    if ((threadIdx.x & 63) == 0) stat_store(O, row, r);
}

// ---- spectral_smooth ------------------------------------------------------------------------------------------
// astropy.convolution.convolve(boundary='fill', fill_value=0, nan_treatment='interpolate', normalize_kernel=True) along z:
// out = sum k d [valid] / sum k [valid], samples outside the cube are VALID zeros, taps equal to 0 are skipped (astropy's loop
// multiplies them: 0 x finite = 0 there too; an infinite sample under a zero tap is the one case kept apart), an empty
// window gives the filled centre sample.  A lane owns one spaxel and produces RUN consecutive channels at a time from the
// ntaps + RUN - 1 input planes of the run (each read once per run; the halo of the next run comes from the L2).
template <class F, int... G>
__device__ __forceinline__ void for_each_const(F&& f, std::integer_sequence<int, G...>) { (f(std::integral_constant<int, G>{}), ...); }
constexpr int kRun = 16;
struct Conv64Args {
    Cube64 c;
    MaskDev64 m;
    double* out;
    int64_t out_row_stride, out_plane_stride;
    const double* kern;      // ntaps
    const double* kpad;      // kpad[15 + j] = kern[j], 15 zeros on either side
    int ntaps;
    const unsigned* gate;    // != nullptr: run only when the word is set (the ring kernel met an infinite valid sample)
};
__global__ __launch_bounds__(256) void spectral_conv64_kernel(const Conv64Args A) {
    if (A.gate && *A.gate == 0u) return;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= A.c.ny * A.c.nx) return;
    const int64_t y = g / A.c.nx, x = g - y * A.c.nx;
    const int H = A.ntaps / 2;
    const int64_t o0 = (int64_t)blockIdx.y * kRun;
    double num[kRun], den[kRun];
#pragma unroll
    for (int u = 0; u < kRun; ++u) { num[u] = 0.0; den[u] = 0.0; }
    // Fast form: the weights of input i for the run's 16 outputs are 16 CONSECUTIVE entries of the zero-padded tap table
    // (wave-uniform: scalar loads), no test per (input, output) pair - ~40 instead of ~200 issue slots per input.  A padding zero
    // times an infinite sample would be NaN where astropy has no product at all: a run that meets an infinity is redone below.
    bool inf_seen = false;
    {
        // (eight inputs requested together, their channels clamped into the cube: one load at a time, a run waited out
        //  ntaps + 15 memory latencies - 9.5 ms per 5e8 voxels)
        constexpr int kIn = 8;
        const bool arr = (A.m.flags & SPC_MASK_ARRAY) != 0;
        const int64_t i0 = o0 - H, nin = 2 * (int64_t)H + kRun;
        const double* pd = A.c.p + y * A.c.row_stride + x;
        const uint8_t* pmk = arr ? A.m.arr + y * A.m.row_stride + x : nullptr;
        for (int64_t c0 = 0; c0 < nin; c0 += kIn) {
            double vv[kIn];
            unsigned mk[kIn];
#pragma unroll
            for (int q = 0; q < kIn; ++q) {
                const int64_t ic = min(max(i0 + c0 + q, (int64_t)0), A.c.nz - 1);
                vv[q] = pd[ic * A.c.plane_stride];
                mk[q] = arr ? pmk[ic * A.m.plane_stride] : 1u;
            }
#pragma unroll
            for (int q = 0; q < kIn; ++q) {
                const int64_t i = i0 + c0 + q;
                if (c0 + q < nin) {                                 // (uniform)
                    const bool inr = i >= 0 && i < A.c.nz;          // (uniform) outside the cube: a valid zero
                    const bool ok = pred64(A.m, vv[q]) && mk[q] != 0u;
                    const double v = (inr && ok) ? vv[q] : 0.0, w = inr ? (ok ? 1.0 : 0.0) : 1.0;
                    inf_seen = inf_seen || (fabs(v) == INFINITY);
                    const double* kp = A.kpad + 15 + (o0 - i + H);  // kp[u] = kern[(o0 + u) - i + H], or a padding zero
#pragma unroll
                    for (int u = 0; u < kRun; ++u) {
                        const double kw = kp[u];
                        num[u] = fma(kw, v, num[u]);
                        den[u] = fma(kw, w, den[u]);
                    }
                }
            }
        }
    }
    if (!__any(inf_seen)) goto emit;
#pragma unroll
    for (int u = 0; u < kRun; ++u) { num[u] = 0.0; den[u] = 0.0; }
    // astropy's order: the flipped kernel is walked from its first element, i.e. the inputs of an output from the lowest channel up
    for (int64_t i = o0 - H; i <= o0 + kRun - 1 + H; ++i) {
        double v = 0.0, w = 1.0;
        if (i >= 0 && i < A.c.nz) {
            const bool ok = inc64(A.c, A.m, i, y, x, v);
            if (!ok) { v = 0.0; w = 0.0; }
        }
#pragma unroll
        for (int u = 0; u < kRun; ++u) {
            const int64_t j = (o0 + u) - i + H;                 // weight kern[o - i + H] (true convolution)
            if (j >= 0 && j < A.ntaps) {                        // (wave-uniform)
                const double kw = A.kern[j];
                if (kw != 0.0) { num[u] = fma(kw, v, num[u]); den[u] = fma(kw, w, den[u]); }
            }
        }
    }
emit:
#pragma unroll
    for (int u = 0; u < kRun; ++u) {
        const int64_t o = o0 + u;
        if (o >= A.c.nz) break;
        double res;
        if (den[u] != 0.0) res = num[u] / den[u];
        else { double cv; res = inc64(A.c, A.m, o, y, x, cv) ? cv : NAN; }
        A.out[o * A.out_plane_stride + y * A.out_row_stride + x] = res;
    }
}

// Second form (round 6): ring streaming, the float32 stencil's layout with float64 places.  A lane owns one spaxel and marches
// over z: the R outputs the newest sample still contributes to live in R (numerator, denominator) register pairs; every input
// is read ONCE (the runs of 16 above read ntaps + 15 planes per 16 outputs and spend 3 (ntaps + 15) / 16 FMAs pairs per output
// where this form spends ntaps).  Place r of the step with phase q (= step mod 4) is the register pair q + r: a step accumulates
// in place and the places move down four registers once per four steps (as spatial64_ring_kernel's y pass).  The samples are
// asked for seven steps ahead (a step is ~0.4 us of arithmetic, a fetch from HBM several times that) in straight-line code -
// eight steps per loop iteration, no branch that holds a load - so that the compiler counts the loads in flight.
// Every output adds its inputs from the lowest channel up, like the form above: the same float64 bits.  Taken for kernels of
// up to R taps, none negative, centre tap positive (an empty window then means an invalid centre sample: NaN, no second
// look at the cube); the taps are centred in R entries, and a padding zero times an INFINITE valid sample would be NaN where
// the form above skips it: a lane that meets one raises a flag and the runs-of-16 kernel, queued behind it, redoes the call.
template <int R>
struct SRing64Args {
    Cube64 c;
    MaskDev64 m;
    double* out;
    int64_t out_row_stride, out_plane_stride;
    double k[R];                            // k[j] = weight of input i for output o = i - R / 2 + j  (true convolution)
    int zchunk;                             // outputs per block along z
    unsigned* flag;                         // != nullptr: the mask admits infinite samples
};
template <int R, bool ARR>
__global__ __launch_bounds__(256, 2) void spectral64_ring_kernel(const SRing64Args<R> A) {
    constexpr int H = R / 2, U = 4, NR = R + U - 1, NQ = 8, kPre = NQ - 1;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= A.c.ny * A.c.nx) return;
    const int64_t y = g / A.c.nx, x = g - y * A.c.nx;
    const int nz = (int)A.c.nz;
    const int oa = (int)blockIdx.y * A.zchunk, ob = min(oa + A.zchunk, nz);
    const double* pd = A.c.p + y * A.c.row_stride + x;
    const uint8_t* pmk = ARR ? A.m.arr + y * A.m.row_stride + x : nullptr;
    double* po = A.out + y * A.out_row_stride + x;
    double sn[NR], sd[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { sn[r] = 0.0; sd[r] = 0.0; }
    double vq[NQ];
    unsigned mq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { vq[q] = 0.0; mq[q] = 1u; }
    auto ask = [&](auto slot, int i) {                       // (channels clamped into the cube: what lies outside is not used)
        constexpr int S = decltype(slot)::value;
        const int ic = min(max(i, 0), nz - 1);
        vq[S] = pd[(int64_t)ic * A.c.plane_stride];
        if (ARR) mq[S] = pmk[(int64_t)ic * A.m.plane_stride];
    };
    bool seen_inf = false;
    auto step = [&](auto slot, const int i) {
        constexpr int S = decltype(slot)::value, Q = S % U;
        const double v0 = vq[S];
        const bool inr = i >= 0 && i < nz;                   // (uniform) outside the cube: a valid zero
        const bool ok = pred64(A.m, v0) && mq[S] != 0u;
        const double v = (inr && ok) ? v0 : 0.0, w = inr ? (ok ? 1.0 : 0.0) : 1.0;
        if (A.flag) seen_inf = seen_inf || fabs(v) == INFINITY;
        ask(std::integral_constant<int, (S + kPre) % NQ>{}, i + kPre);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            sn[Q + r] = fma(A.k[r], v, sn[Q + r]);
            sd[Q + r] = fma(A.k[r], w, sd[Q + r]);
        }
        const int o = i - H;                                 // the output that has seen its last input
        if (o >= oa && o < ob) po[(int64_t)o * A.out_plane_stride] = sd[Q] != 0.0 ? sn[Q] / sd[Q] : NAN;
        if (Q == U - 1) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                sn[r] = r + U < NR ? sn[r + U] : 0.0;
                sd[r] = r + U < NR ? sd[r + U] : 0.0;
            }
        }
    };
    const int i_begin = oa - H, i_end = ob + H;
    static_assert(NQ == 8, "eight steps written out");
    ask(std::integral_constant<int, 0>{}, i_begin);     ask(std::integral_constant<int, 1>{}, i_begin + 1);
    ask(std::integral_constant<int, 2>{}, i_begin + 2); ask(std::integral_constant<int, 3>{}, i_begin + 3);
    ask(std::integral_constant<int, 4>{}, i_begin + 4); ask(std::integral_constant<int, 5>{}, i_begin + 5);
    ask(std::integral_constant<int, 6>{}, i_begin + 6);
    for (int i0 = i_begin; i0 < i_end; i0 += NQ) {
        step(std::integral_constant<int, 0>{}, i0);     step(std::integral_constant<int, 1>{}, i0 + 1);
        step(std::integral_constant<int, 2>{}, i0 + 2); step(std::integral_constant<int, 3>{}, i0 + 3);
        step(std::integral_constant<int, 4>{}, i0 + 4); step(std::integral_constant<int, 5>{}, i0 + 5);
        step(std::integral_constant<int, 6>{}, i0 + 6); step(std::integral_constant<int, 7>{}, i0 + 7);
    }
    if (A.flag && seen_inf) atomicOr(A.flag, 1u);
}

// ---- spatial_smooth -------------------------------------------------------------------------------------------
// Per channel, the same convolution in 2-D.  An outer-product kernel runs as two passes over a slab of planes: the x pass
// leaves (sum kx d [valid], sum kx [valid]) per voxel in the caller's workspace, the y pass combines them and divides;
// any other kernel is summed directly (nky x nkx taps per output, neighbours from the caches).
struct Sp64Args {
    Cube64 c;
    MaskDev64 m;
    double* out;
    int64_t out_row_stride, out_plane_stride;
    const double* ky; const double* kx;     // separable: the two factors; direct: ky = the 2-D table, kx unused
    int nky, nkx;
    double* tmp;                            // (planes, ny, nx, 2) of the slab
    int64_t z0;                             // first plane of the slab
    const unsigned* gate;                   // != nullptr: run only when the word is set (the ring kernel met an infinite valid sample)
};
__global__ __launch_bounds__(256) void spatial64_xpass_kernel(const Sp64Args A) {
    if (A.gate && *A.gate == 0u) return;
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t y = blockIdx.y, zl = blockIdx.z, z = A.z0 + zl;
    if (x >= A.c.nx) return;
    const int H = A.nkx / 2;
    double num = 0.0, den = 0.0;
    const bool arr = (A.m.flags & SPC_MASK_ARRAY) != 0;
    const double* pd = A.c.p + z * A.c.plane_stride + y * A.c.row_stride;
    const uint8_t* pmk = arr ? A.m.arr + z * A.m.plane_stride + y * A.m.row_stride : nullptr;
    constexpr int kIn = 8;                                       // loads requested together (clamped into the row)
    for (int j0 = 0; j0 < A.nkx; j0 += kIn) {                    // astropy's order along the fast axis
        double vv[kIn];
        unsigned mk[kIn];
#pragma unroll
        for (int q = 0; q < kIn; ++q) {
            const int64_t ic = min(max(x - H + j0 + q, (int64_t)0), A.c.nx - 1);
            vv[q] = pd[ic];
            mk[q] = arr ? pmk[ic] : 1u;
        }
#pragma unroll
        for (int q = 0; q < kIn; ++q) {
            const int j = j0 + q;
            if (j < A.nkx) {                                     // (uniform)
                const double kw = A.kx[A.nkx - 1 - j];           // flipped kernel, input x - H + j
                const int64_t i = x - H + j;
                const bool inr = i >= 0 && i < A.c.nx;
                const bool ok = pred64(A.m, vv[q]) && mk[q] != 0u;
                const double v = (inr && ok) ? vv[q] : 0.0, w = inr ? (ok ? 1.0 : 0.0) : 1.0;
                if (kw != 0.0) { num = fma(kw, v, num); den = fma(kw, w, den); }
            }
        }
    }
    double* t = A.tmp + ((zl * A.c.ny + y) * A.c.nx + x) * 2;
    t[0] = num; t[1] = den;
}
__global__ __launch_bounds__(256) void spatial64_ypass_kernel(const Sp64Args A) {
    if (A.gate && *A.gate == 0u) return;
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t y = blockIdx.y, zl = blockIdx.z, z = A.z0 + zl;
    if (x >= A.c.nx) return;
    const int H = A.nky / 2;
    // a row outside the image is a row of valid zeros: its x-pass denominator is the whole x kernel
    double ksx = 0.0;
    for (int j = 0; j < A.nkx; ++j) ksx += A.kx[j];
    double num = 0.0, den = 0.0;
    constexpr int kIn = 8;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    for (int j0 = 0; j0 < A.nky; j0 += kIn) {
        f64x2 tt[kIn];
#pragma unroll
        for (int q = 0; q < kIn; ++q) {
            const int64_t ic = min(max(y - H + j0 + q, (int64_t)0), A.c.ny - 1);
            tt[q] = *reinterpret_cast<const f64x2*>(A.tmp + ((zl * A.c.ny + ic) * A.c.nx + x) * 2);
        }
#pragma unroll
        for (int q = 0; q < kIn; ++q) {
            const int j = j0 + q;
            if (j < A.nky) {
                const double kw = A.ky[A.nky - 1 - j];
                const int64_t i = y - H + j;
                const bool inr = i >= 0 && i < A.c.ny;           // (uniform) a row outside the image: valid zeros
                const double tn = inr ? tt[q].x : 0.0, td = inr ? tt[q].y : ksx;
                if (kw != 0.0) { num = fma(kw, tn, num); den = fma(kw, td, den); }
            }
        }
    }
    double res;
    if (den != 0.0) res = num / den;
    else { double cv; res = inc64(A.c, A.m, z, y, x, cv) ? cv : NAN; }
    A.out[z * A.out_plane_stride + y * A.out_row_stride + x] = res;
}
// The two passes through LDS: a block of the x pass stages its row segment once (256 outputs + the halo; the untiled pass asks the
// texture path for 58 samples per output), a block of the y pass a tile of 16 rows x 64 columns of (num, den) pairs + the halo rows
// (464 bytes of L2 traffic per output otherwise).  Same arithmetic, same order.
constexpr int kSpHaloMax = 511, kSpYRows = 16, kSpYHalo = 23;     // (y tiles of up to 62 rows x 64 columns x 16 bytes = 62 KB of dynamic LDS + the taps: inside the 64 KB a launch gets without hipFuncSetAttribute; 49 y taps and more take the untiled pass)
__global__ __launch_bounds__(256) void spatial64_xpass_lds_kernel(const Sp64Args A) {
    if (A.gate && *A.gate == 0u) return;
    __shared__ double sv[256 + 2 * kSpHaloMax];
    __shared__ float sw[256 + 2 * kSpHaloMax];
    __shared__ double sk[2 * kSpHaloMax + 1];                   // the flipped taps (a scalar load per tap waited out its latency in every lane's loop)
    const int t = threadIdx.x, H = A.nkx / 2;
    for (int j = t; j < A.nkx; j += 256) sk[j] = A.kx[A.nkx - 1 - j];
    const int64_t x0 = (int64_t)blockIdx.x * 256, y = blockIdx.y, zl = blockIdx.z, z = A.z0 + zl;
    const bool arr = (A.m.flags & SPC_MASK_ARRAY) != 0;
    const double* pd = A.c.p + z * A.c.plane_stride + y * A.c.row_stride;
    const uint8_t* pmk = arr ? A.m.arr + z * A.m.plane_stride + y * A.m.row_stride : nullptr;
    for (int e = t; e < 256 + 2 * H; e += 256) {
        const int64_t i = x0 - H + e, ic = min(max(i, (int64_t)0), A.c.nx - 1);
        const double v = pd[ic];
        const bool inr = i >= 0 && i < A.c.nx, ok = pred64(A.m, v) && (!arr || pmk[ic] != 0);
        sv[e] = (inr && ok) ? v : 0.0;
        sw[e] = inr ? (ok ? 1.f : 0.f) : 1.f;                    // outside the image: a valid zero
    }
    __syncthreads();
    const int64_t x = x0 + t;
    if (x >= A.c.nx) return;
    double num = 0.0, den = 0.0;
    for (int j = 0; j < A.nkx; ++j) {                            // astropy's order along the fast axis
        const double kw = sk[j];                                 // flipped kernel, input x - H + j
        if (kw != 0.0) { num = fma(kw, sv[t + j], num); den = fma(kw, (double)sw[t + j], den); }
    }
    double* o = A.tmp + ((zl * A.c.ny + y) * A.c.nx + x) * 2;
    o[0] = num; o[1] = den;
}
__global__ __launch_bounds__(256) void spatial64_ypass_lds_kernel(const Sp64Args A) {
    if (A.gate && *A.gate == 0u) return;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    extern __shared__ f64x2 tile[];                              // (kSpYRows + 2 H) x 64 pairs: sized by the launch
    __shared__ double sk[2 * kSpYHalo + 1];
    const int t = threadIdx.x, col = t & 63, rg = t >> 6, H = A.nky / 2;
    if (t < A.nky) sk[t] = A.ky[A.nky - 1 - t];
    const int64_t x = (int64_t)blockIdx.x * 64 + col, y0 = (int64_t)blockIdx.y * kSpYRows, zl = blockIdx.z, z = A.z0 + zl;
    double ksx = 0.0;                                            // a row outside the image: valid zeros under the whole x kernel
    for (int j = 0; j < A.nkx; ++j) ksx += A.kx[j];
    const int64_t xc = min(x, A.c.nx - 1);
    for (int r0 = rg; r0 < kSpYRows + 2 * H; r0 += 16) {         // (four rows per lane requested together, clamped into the image)
        f64x2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t ic = min(max(y0 - H + r0 + 4 * q, (int64_t)0), A.c.ny - 1);
            v[q] = *reinterpret_cast<const f64x2*>(A.tmp + ((zl * A.c.ny + ic) * A.c.nx + xc) * 2);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = r0 + 4 * q;
            const int64_t i = y0 - H + rr;
            if (rr < kSpYRows + 2 * H) tile[rr * 64 + col] = (i >= 0 && i < A.c.ny) ? v[q] : f64x2{0.0, ksx};
        }
    }
    __syncthreads();
    if (x >= A.c.nx) return;
    for (int ro = rg; ro < kSpYRows; ro += 4) {
        const int64_t y = y0 + ro;
        if (y >= A.c.ny) break;
        double num = 0.0, den = 0.0;
        for (int j = 0; j < A.nky; ++j) {
            const double kw = sk[j];
            if (kw != 0.0) { const f64x2 tv = tile[(ro + j) * 64 + col]; num = fma(kw, tv.x, num); den = fma(kw, tv.y, den); }
        }
        double res;
        if (den != 0.0) res = num / den;
        else { double cv; res = inc64(A.c, A.m, z, y, x, cv) ? cv : NAN; }
        A.out[z * A.out_plane_stride + y * A.out_row_stride + x] = res;
    }
}
// Round 6: the same two passes with FOUR adjacent outputs per thread.  A sample read from LDS feeds up to four accumulators
// (the outputs x .. x + 3 / the rows y .. y + 3 see it under the taps j .. j - 3, kept in a four-deep register window), so a
// block reads (nk + 3) samples and taps per four outputs where the one-output passes read nk per output: the passes were bound
// by LDS traffic (3 reads per tap and output), not by the 2 x nk float64 FMAs.  Every output still adds its taps in astropy's
// order (j ascending), zero taps skipped: the same float64 results, bit for bit.
// x pass: a block = 1024 outputs of one row; samples in LDS at s + (s >> 2) - one pad per four - so that the lanes of a wave,
// four samples apart, read 64-bit words five apart: no bank is asked twice within a half wave.
constexpr int kSpX4Halo = 127, kSpX4Out = 1024;
__device__ __forceinline__ int sp64_skew(int s) { return s + (s >> 2); }
__global__ __launch_bounds__(256) void spatial64_xpass_lds4_kernel(const Sp64Args A) {
    if (A.gate && *A.gate == 0u) return;
    constexpr int N = kSpX4Out + 2 * kSpX4Halo + 4;
    __shared__ double sv[N + N / 4 + 1];
    __shared__ float sw[N + N / 4 + 1];
    __shared__ double sk[2 * kSpX4Halo + 1];
    const int t = threadIdx.x, H = A.nkx / 2;
    for (int j = t; j < A.nkx; j += 256) sk[j] = A.kx[A.nkx - 1 - j];
    const int64_t x0 = (int64_t)blockIdx.x * kSpX4Out, y = blockIdx.y, zl = blockIdx.z, z = A.z0 + zl;
    const bool arr = (A.m.flags & SPC_MASK_ARRAY) != 0;
    const double* pd = A.c.p + z * A.c.plane_stride + y * A.c.row_stride;
    const uint8_t* pmk = arr ? A.m.arr + z * A.m.plane_stride + y * A.m.row_stride : nullptr;
    for (int e = t; e < kSpX4Out + 2 * H + 3; e += 256) {
        const int64_t i = x0 - H + e, ic = min(max(i, (int64_t)0), A.c.nx - 1);
        const double v = pd[ic];
        const bool inr = i >= 0 && i < A.c.nx, ok = pred64(A.m, v) && (!arr || pmk[ic] != 0);
        sv[sp64_skew(e)] = (inr && ok) ? v : 0.0;
        sw[sp64_skew(e)] = inr ? (ok ? 1.f : 0.f) : 1.f;         // outside the image: a valid zero
    }
    __syncthreads();
    const int64_t x = x0 + 4 * t;
    if (x >= A.c.nx) return;
    double num[4] = {0.0, 0.0, 0.0, 0.0}, den[4] = {0.0, 0.0, 0.0, 0.0};
    double k0 = 0.0, k1 = 0.0, k2 = 0.0, k3 = 0.0;               // taps i, i - 1, i - 2, i - 3 (0 outside the kernel: skipped like a zero tap)
    for (int i = 0; i < A.nkx + 3; ++i) {
        k3 = k2; k2 = k1; k1 = k0; k0 = i < A.nkx ? sk[i] : 0.0;
        const int p = sp64_skew(4 * t + i);
        const double v = sv[p], w = (double)sw[p];
        if (k0 != 0.0) { num[0] = fma(k0, v, num[0]); den[0] = fma(k0, w, den[0]); }
        if (k1 != 0.0) { num[1] = fma(k1, v, num[1]); den[1] = fma(k1, w, den[1]); }
        if (k2 != 0.0) { num[2] = fma(k2, v, num[2]); den[2] = fma(k2, w, den[2]); }
        if (k3 != 0.0) { num[3] = fma(k3, v, num[3]); den[3] = fma(k3, w, den[3]); }
    }
    double* o = A.tmp + ((zl * A.c.ny + y) * A.c.nx + x) * 2;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (x + q < A.c.nx) { o[2 * q] = num[q]; o[2 * q + 1] = den[q]; }
}
// y pass: the tile of kSpYRows + 2 H rows x 64 columns as before; thread (column, row group) takes the rows 4 g .. 4 g + 3
__global__ __launch_bounds__(256) void spatial64_ypass_lds4_kernel(const Sp64Args A) {
    if (A.gate && *A.gate == 0u) return;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    extern __shared__ f64x2 tile[];                              // (kSpYRows + 2 H) x 64 pairs: sized by the launch
    __shared__ double sk[2 * kSpYHalo + 1];
    const int t = threadIdx.x, col = t & 63, rg = t >> 6, H = A.nky / 2;
    if (t < A.nky) sk[t] = A.ky[A.nky - 1 - t];
    const int64_t x = (int64_t)blockIdx.x * 64 + col, y0 = (int64_t)blockIdx.y * kSpYRows, zl = blockIdx.z, z = A.z0 + zl;
    double ksx = 0.0;                                            // a row outside the image: valid zeros under the whole x kernel
    for (int j = 0; j < A.nkx; ++j) ksx += A.kx[j];
    const int64_t xc = min(x, A.c.nx - 1);
    for (int r0 = rg; r0 < kSpYRows + 2 * H; r0 += 16) {         // (four rows per lane requested together, clamped into the image)
        f64x2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t ic = min(max(y0 - H + r0 + 4 * q, (int64_t)0), A.c.ny - 1);
            v[q] = *reinterpret_cast<const f64x2*>(A.tmp + ((zl * A.c.ny + ic) * A.c.nx + xc) * 2);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = r0 + 4 * q;
            const int64_t i = y0 - H + rr;
            if (rr < kSpYRows + 2 * H) tile[rr * 64 + col] = (i >= 0 && i < A.c.ny) ? v[q] : f64x2{0.0, ksx};
        }
    }
    __syncthreads();
    if (x >= A.c.nx) return;
    static_assert(kSpYRows == 16, "four row groups of four rows");
    double num[4] = {0.0, 0.0, 0.0, 0.0}, den[4] = {0.0, 0.0, 0.0, 0.0};
    double k0 = 0.0, k1 = 0.0, k2 = 0.0, k3 = 0.0;
    for (int i = 0; i < A.nky + 3; ++i) {
        k3 = k2; k2 = k1; k1 = k0; k0 = i < A.nky ? sk[i] : 0.0;
        const f64x2 tv = tile[(4 * rg + i) * 64 + col];
        if (k0 != 0.0) { num[0] = fma(k0, tv.x, num[0]); den[0] = fma(k0, tv.y, den[0]); }
        if (k1 != 0.0) { num[1] = fma(k1, tv.x, num[1]); den[1] = fma(k1, tv.y, den[1]); }
        if (k2 != 0.0) { num[2] = fma(k2, tv.x, num[2]); den[2] = fma(k2, tv.y, den[2]); }
        if (k3 != 0.0) { num[3] = fma(k3, tv.x, num[3]); den[3] = fma(k3, tv.y, den[3]); }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t y = y0 + 4 * rg + q;
        if (y >= A.c.ny) break;
        double res;
        if (den[q] != 0.0) res = num[q] / den[q];
        else { double cv; res = inc64(A.c, A.m, z, y, x, cv) ? cv : NAN; }
        A.out[z * A.out_plane_stride + y * A.out_row_stride + x] = res;
    }
}
// Third form (round 6): ONE kernel, nothing between the passes.  A block = one plane x a strip of 256 columns x a band of rows;
// it marches DOWN the band: a row segment (256 + 2 H columns) is classified once into LDS as (valid ? sample : 0, validity)
// pairs, every thread gathers the x pass of ITS column from there (R reads of 16 bytes, 2 R float64 FMAs, taps in scalar
// registers), and the y pass runs in SCATTER form on registers: the R outputs the new row still contributes to live in R
// (numerator, denominator) accumulator pairs that move down one place per row - acc[r] = fma(k[r], X, acc[r + 1]) accumulates
// AND shifts, no copies - so that acc[0] is the output that has just seen its last row.  The cube is read once and written
// once (the two-pass forms above move another 32 bytes per voxel through the workspace and re-read the y halo of every
// 16-row tile: traffic x2.9 of the algorithmic bytes).  Every output adds its taps in the order of the passes above (x taps
// ascending, then rows ascending): the same float64 values.  Symmetric kernels of up to R taps per axis (zero-padded to R;
// 17 distinct taps per axis fit the scalar registers, 33 would not); a padded tap times an INFINITE valid sample would be NaN
// where the passes above skip zero taps, so a block that meets one (only possible when the mask admits infinities) raises
// a flag and the two-pass kernels - launched behind it, retiring at once while the flag is clear - redo the call.
template <int R>
struct Ring64Args {
    Cube64 c;
    MaskDev64 m;
    double* out;
    int64_t out_row_stride, out_plane_stride;
    double ky[R / 2 + 1], kx[R / 2 + 1];    // k'[i] = k'[R - 1 - i], i = 0 .. R / 2: the taps centred in R entries
    double ksx;                             // the x taps' sum: what a row outside the image (valid zeros) gives
    int nstrips, nbands, band_rows;
    unsigned* flag;                         // != nullptr: the mask admits infinite samples
};
template <int R, bool ARR>
__global__ __launch_bounds__(256, 2) void spatial64_ring_kernel(const Ring64Args<R> A) {
    constexpr int H = R / 2, W = 256, NS = W + 2 * H;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    __shared__ f64x2 buf[2][NS];
    const int t = threadIdx.x;
    int64_t b = blockIdx.x;                                  // strips fastest: x neighbours (2 H shared columns) run side by side
    const int strip = (int)(b % A.nstrips); b /= A.nstrips;
    const int band = (int)(b % A.nbands);
    const int64_t z = b / A.nbands;
    const int nx = (int)A.c.nx, ny = (int)A.c.ny;            // (ny <= 65535 and nx < 2^31: checked by the entry point)
    const int x0 = strip * W;
    const int ya = band * A.band_rows, yb = min(ya + A.band_rows, ny);
    const bool second = t < 2 * H;                           // this thread also stages element W + t (the first wave's lanes: R <= 33)
    const bool wave0 = __builtin_amdgcn_readfirstlane(t >> 6) == 0;
    const int c0 = x0 - H + t, c1 = c0 + W;
    const bool in0 = c0 >= 0 && c0 < nx, in1 = c1 < nx;
    // (lanes without a second element ask for their first one again - every lane issues both loads, see load_row - instead of
    //  fetching 256 columns of the NEXT strip: the kernel read 1.8x its input that way, PMC)
    const unsigned cc0 = (unsigned)min(max(c0, 0), nx - 1), cc1 = second ? (unsigned)min(c1, nx - 1) : cc0;
    const double* pd = A.c.p + z * A.c.plane_stride;
    const uint8_t* pm = ARR ? A.m.arr + z * A.m.plane_stride : nullptr;
    // the ring: logical place r of the row with phase q (= row mod U inside an unrolled group of U rows) is the register pair
    // q + r, so that a row accumulates IN PLACE (v_fmac_f64) and the places move down by U registers once per U rows
    // (64 / U v_mov_b64 per row; a place that moves down every row costs 66, or hand-written three-address FMAs the register
    // allocator answers with scratch)
    constexpr int U = 4, NR = R + U - 1;
    double sn[NR], sd[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { sn[r] = 0.0; sd[r] = 0.0; }
    // the rows' own samples: asked for kPre rows ahead (a row of this wave lasts ~1.5 us, a fetch from HBM under load longer
    // than that), slot = the row's phase.  Straight-line code - rows clamped into the image, every lane asks for both of
    // "its" elements - so that the compiler can count the loads in flight (behind a branch it waits for all of them)
    constexpr int kPre = 3;
    double rq0[U], rq1[U];
    unsigned mq0[U], mq1[U];
#pragma unroll
    for (int q = 0; q < U; ++q) { rq0[q] = 0.0; rq1[q] = 0.0; mq0[q] = 1u; mq1[q] = 1u; }
    auto load_row = [&](auto slot, int u) {                  // (a uniform row pointer + a 32-bit lane offset)
        constexpr int S = decltype(slot)::value;
        const int uc = min(max(u, 0), ny - 1);
        const double* q = pd + (int64_t)uc * A.c.row_stride;
        rq0[S] = q[cc0];
        rq1[S] = q[cc1];
        if (ARR) {
            const uint8_t* qm = pm + (int64_t)uc * A.m.row_stride;
            mq0[S] = qm[cc0];
            mq1[S] = qm[cc1];
        }
    };
    const int u_begin = ya - H, u_end = yb + H;
    static_assert(kPre == 3 && U == 4, "the first rows' slots written out");
    load_row(std::integral_constant<int, 0>{}, u_begin);
    load_row(std::integral_constant<int, 1>{}, u_begin + 1);
    load_row(std::integral_constant<int, 2>{}, u_begin + 2);
    bool seen_inf = false;
    const int x = x0 + t;
    double* po = A.out + z * A.out_plane_stride + x;
    // Software pipeline: the step of row u stages row u, runs the y pass of row u - 1 (Xpn, Xpd: registers only), then the
    // x pass of row u from LDS.  A row outside the image is staged as valid zeros (its x pass gives (0, sum of the x taps)
    // like any other row: one code path); the march runs on to a multiple of U rows past the band (nothing is stored there).
    double Xpn = 0.0, Xpd = 0.0;                             // (before the first row: zeros into an empty ring)
    auto row_step = [&](auto phase, const int u) {
        constexpr int Q = decltype(phase)::value;
        const bool row_in = u >= 0 && u < ny;                // (uniform)
        const double r0 = rq0[Q], r1 = rq1[Q];
        const unsigned m0 = mq0[Q], m1 = mq1[Q];
        f64x2* const Bw = buf[Q & 1];                        // (U is even: the buffer is the phase's parity)
        {
            const bool ok = pred64(A.m, r0) && m0 != 0u;
            Bw[t] = (in0 && row_in) ? (ok ? f64x2{r0, 1.0} : f64x2{0.0, 0.0}) : f64x2{0.0, 1.0};      // outside the image: a valid zero
            if (A.flag) seen_inf = seen_inf || (ok && in0 && row_in && fabs(r0) == INFINITY);
        }
        if (wave0 && second) {                              // (a scalar branch for three of the four waves)
            const bool ok = pred64(A.m, r1) && m1 != 0u;
            Bw[W + t] = (in1 && row_in) ? (ok ? f64x2{r1, 1.0} : f64x2{0.0, 0.0}) : f64x2{0.0, 1.0};
            if (A.flag) seen_inf = seen_inf || (ok && in1 && row_in && fabs(r1) == INFINITY);
        }
        load_row(std::integral_constant<int, (Q + kPre) % U>{}, u + kPre);      // in flight under the arithmetic of three rows
        // (one barrier per row: a buffer is written again two rows later, behind the barrier of the row in between)
        __syncthreads();
        // y pass of row u - 1: place r takes tap R - 1 - r (symmetric: tap r); place 0 has then seen its last row
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double kk = A.ky[r <= H ? r : R - 1 - r];
            sn[Q + r] = fma(kk, Xpn, sn[Q + r]);
            sd[Q + r] = fma(kk, Xpd, sd[Q + r]);
        }
        const int y = u - 1 - H;
        if (y >= ya && y < yb && x < nx) {
            // (no tap negative, centre taps positive: an empty window means an invalid centre sample - NaN, no second look at the cube,
            //  and no load behind a branch)
            po[(int64_t)y * A.out_row_stride] = sd[Q] != 0.0 ? sn[Q] / sd[Q] : NAN;
        }
        // x pass of row u.  Left alone the compiler asks for all R pairs at once - 132 registers beside the ring's 144 - and
        // sends a part of the ring to scratch (a sched_barrier does not hold the reads back: they are hoisted before the
        // machine scheduler runs).  The reads of a group's slots therefore go through a pointer that an empty asm statement
        // "redefines" together with the running sum: they cannot move above the FMAs that emptied their slots.
        constexpr int kD = 8, kG = 4;                        // pairs in flight, taps per group
        typedef const f64x2 __attribute__((address_space(3)))* lds_pairs;
        lds_pairs Bp = (lds_pairs)(Bw + t);
        double Xn = 0.0, Xd = 0.0;
        f64x2 sq[kD];
#pragma unroll
        for (int j = 0; j < kD; ++j) sq[j] = Bp[j];
        auto group = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
#pragma unroll
            for (int q = 0; q < kG; ++q) {
                const int j = g * kG + q;
                if (j < R) {
                    const double kk = A.kx[j <= H ? j : R - 1 - j];
                    Xn = fma(kk, sq[j % kD].x, Xn);
                    Xd = fma(kk, sq[j % kD].y, Xd);
                }
            }
            asm volatile("" : "+v"(Bp), "+v"(Xn), "+v"(Xd));
#pragma unroll
            for (int q = 0; q < kG; ++q) {
                const int j = g * kG + q;
                if (j + kD < R) sq[j % kD] = Bp[j + kD];
            }
        };
        for_each_const(group, std::make_integer_sequence<int, (R + kG - 1) / kG>{});
        Xpn = Xn; Xpd = Xd;
    };
    static_assert(U == 4, "four phases written out");
    for (int u0 = u_begin; u0 <= u_end; u0 += U) {
        row_step(std::integral_constant<int, 0>{}, u0);
        row_step(std::integral_constant<int, 1>{}, u0 + 1);
        row_step(std::integral_constant<int, 2>{}, u0 + 2);
        row_step(std::integral_constant<int, 3>{}, u0 + 3);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            sn[i] = i + U < NR ? sn[i + U] : 0.0;
            sd[i] = i + U < NR ? sd[i + U] : 0.0;
        }
    }
    if (A.flag && seen_inf) atomicOr(A.flag, 1u);
}


__global__ __launch_bounds__(256) void spatial64_direct_kernel(const Sp64Args A) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t y = blockIdx.y, z = A.z0 + blockIdx.z;
    if (x >= A.c.nx) return;
    const int Hy = A.nky / 2, Hx = A.nkx / 2;
    double num = 0.0, den = 0.0;
    for (int jy = 0; jy < A.nky; ++jy) {
        const int64_t iy = y - Hy + jy;
        for (int jx = 0; jx < A.nkx; ++jx) {
            const double kw = A.ky[(A.nky - 1 - jy) * A.nkx + (A.nkx - 1 - jx)];
            if (kw == 0.0) continue;
            const int64_t ix = x - Hx + jx;
            double v = 0.0, w = 1.0;
            if (iy >= 0 && iy < A.c.ny && ix >= 0 && ix < A.c.nx) {
                if (!inc64(A.c, A.m, z, iy, ix, v)) { v = 0.0; w = 0.0; }
            }
            num = fma(kw, v, num); den = fma(kw, w, den);
        }
    }
    double res;
    if (den != 0.0) res = num / den;
    else { double cv; res = inc64(A.c, A.m, z, y, x, cv) ? cv : NAN; }
    A.out[z * A.out_plane_stride + y * A.out_row_stride + x] = res;
}

// ---- spectral_interpolate -------------------------------------------------------------------------------------
struct Lerp64Args {
    Cube64 c;
    MaskDev64 m;
    int64_t nz_out;
    const int32_t* lo; const double* t; const double* inv_dx;
    double fill;
    double* out;
    int64_t out_row_stride, out_plane_stride, jchunk;
};
__global__ __launch_bounds__(256) void spectral_lerp64_kernel(const Lerp64Args A) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= A.c.ny * A.c.nx) return;
    const int64_t y = g / A.c.nx, x = g - y * A.c.nx;
    const int64_t jb = (int64_t)blockIdx.y * A.jchunk, je = min(A.nz_out, jb + A.jchunk);
    int cur = -2;
    double ylo = NAN, yhi = NAN;
    for (int64_t j = jb; j < je; ++j) {
        const int lo = A.lo[j];
        double res = A.fill;
        if (lo >= 0) {
            if (lo != cur) {
                double v;
                if (lo == cur + 1) ylo = yhi; else ylo = inc64(A.c, A.m, lo, y, x, v) ? v : NAN;
                yhi = inc64(A.c, A.m, lo + 1, y, x, v) ? v : NAN;
                cur = lo;
            }
            // scipy: slope = (y_hi - y_lo) / (x_hi - x_lo); y = slope * (x_new - x_lo) + y_lo
            res = (yhi - ylo) * A.inv_dx[j] * A.t[j] + ylo;
        }
        A.out[j * A.out_plane_stride + y * A.out_row_stride + x] = res;
    }
}

// ---- reproject: bilinear / nearest resampling --------------------------------------------------------------------
// bilinear_kernel's semantics (scipy map_coordinates(order = 1 | 0) on the edge-replicated image, NaN outside [-0.5, n - 0.5],
// a NaN neighbour propagates even with weight 0) with float64 samples, float64 weights and a float64 result.  Gather form:
// a lane per output pixel of a compact 16 x 4 tile per wave, marching the channels.
struct Bil64Args {
    Cube64 c;
    MaskDev64 m;
    double fill;
    int64_t ny_out, nx_out;
    const double* xs; const double* ys;
    double* out;
    int64_t out_row_stride, out_plane_stride, zchunk;
    uint8_t* footprint;
    int nearest;
    unsigned int* any_valid;
};
__global__ __launch_bounds__(256) void bilinear64_kernel(const Bil64Args A) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tiles_x = (A.nx_out + 63) / 64;
    const int64_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int64_t xo = bx * 64 + (wave * 16) + (lane & 15), yo = by * 4 + (lane >> 4);
    if (xo >= A.nx_out || yo >= A.ny_out) return;
    const int64_t pix = yo * A.nx_out + xo;
    const double xs = A.xs[pix], ys = A.ys[pix];
    const bool inside = (xs >= -0.5) && (xs <= (double)A.c.nx - 0.5) && (ys >= -0.5) && (ys <= (double)A.c.ny - 0.5);
    if (blockIdx.y == 0 && A.footprint) A.footprint[pix] = inside ? 1 : 0;
    const int64_t zb = (int64_t)blockIdx.y * A.zchunk, ze = min(A.c.nz, zb + A.zchunk);
    double* po = A.out + yo * A.out_row_stride + xo;
    if (!inside) {
        for (int64_t z = zb; z < ze; ++z) po[z * A.out_plane_stride] = NAN;
        return;
    }
    const double xf = A.nearest ? floor(xs + 0.5) : floor(xs), yf = A.nearest ? floor(ys + 0.5) : floor(ys);
    const int64_t x0 = min(max((int64_t)xf, (int64_t)0), A.c.nx - 1), y0 = min(max((int64_t)yf, (int64_t)0), A.c.ny - 1);
    const int64_t x1 = A.nearest ? x0 : min((int64_t)xf + 1, A.c.nx - 1), y1 = A.nearest ? y0 : min((int64_t)yf + 1, A.c.ny - 1);
    const double fx = xs - xf, fy = ys - yf;
    const double w00 = (1.0 - fy) * (1.0 - fx), w01 = (1.0 - fy) * fx, w10 = fy * (1.0 - fx), w11 = fy * fx;
    bool anyv = false;
    for (int64_t z = zb; z < ze; ++z) {
        // excluded voxels are replaced by the cube's fill value (spectral_cube.py:2709-2712); a NaN sample that an array-only
        // mask includes stays what it is (np.where(include, data, fill))
        auto get = [&](int64_t yy, int64_t xx) {
            double v;
            bool ok = inc64(A.c, A.m, z, yy, xx, v);
            if (!ok && v != v && !(A.m.flags & ~SPC_MASK_ARRAY))
                ok = !(A.m.flags & SPC_MASK_ARRAY) || A.m.arr[z * A.m.plane_stride + yy * A.m.row_stride + xx] != 0;
            return (A.m.flags && !ok) ? A.fill : v;
        };
        const double a = get(y0, x0), b = get(y0, x1), c = get(y1, x0), d = get(y1, x1);
        const double r = A.nearest ? a : (a * w00 + b * w01 + c * w10 + d * w11);
        anyv = anyv || (r == r);
        po[z * A.out_plane_stride] = r;
    }
    if (A.any_valid && __any(anyv) && lane == 0) atomicOr(A.any_valid, 1u);
}
__global__ __launch_bounds__(256) void scale64_kernel(double* p, int64_t n, double f) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] *= f;
}

// ---- order statistics along the spectral axis ------------------------------------------------------------------------------
// median / percentile / mad_std (dask_spectral_cube.py:657-731) and sigma_clip_spectrally (:851-878, centre = median | mean,
// spread = std | mad_std) of float64 rays of up to 4096 samples.  No register-resident radix descent like the float32 kernels: a block sorts
// its TS = 4096 / NZP adjacent rays (order-preserving 64-bit keys, excluded / NaN samples last) in 32 KB of LDS with a bitonic
// network, then one lane per ray reads what it needs from the sorted ray - the order statistics directly (numpy's linear rule,
// its `_lerp` form), the clip loop as a shrinking window [a, b) of the sorted samples (clipping removes the two ends of a sorted ray;
// mean and std of a window are two scans of it, nanstd's two-pass form).  The clipped cube is written in a last sweep over the
// rays: a sample survives when its key lies between the window's end keys.
constexpr int kSortKeys = 4096;          // 32 KB of keys per block: four or five blocks per CU (16384 keys, one block: 41 ms per 5e8 voxels;
                                         // 8192: 21.8; 4096: 14.7 - the network's barriers and LDS round trips want other blocks to run beside them)
constexpr unsigned long long kExcl = ~0ull;
__device__ __forceinline__ unsigned long long fkey64(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double funkey64(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}
struct Sort64Args {
    Cube64 c;
    MaskDev64 m;
    int nzp, ts;                 // padded ray length (power of two), rays per block
    double q, scale;
    const double* center;        // NULL, or (ny, nx): keys of |x - center|
    double* out;                 // MODE 0: (ny, nx); MODE 1: (nz, ny, nx) C-contiguous
    double lo_s, hi_s;
    int maxiters, cen_mean, spread_mad;
};
template <int MODE>
__global__ __launch_bounds__(256) void sort64_kernel(const Sort64Args A) {
    __shared__ unsigned long long keys[kSortKeys];           // [sample][ray]: consecutive lanes = consecutive rays
    __shared__ unsigned long long s_wlo[64], s_whi[64];
    const int t = threadIdx.x, TS = A.ts, NZP = A.nzp, L = 256 / TS;
    const int64_t tiles_x = (A.c.nx + TS - 1) / TS;
    const int64_t y = blockIdx.x / tiles_x, x0 = (blockIdx.x % tiles_x) * TS;
    const int r = t % TS, jz = t / TS;
    const bool col_in = x0 + r < A.c.nx;
    const double cen = (A.center && col_in) ? A.center[y * A.c.nx + x0 + r] : 0.0;
    {   // (eight samples requested together per lane, their channels clamped into the ray)
        constexpr int kIn = 8;
        const bool arr = (A.m.flags & SPC_MASK_ARRAY) != 0;
        const int64_t xc = col_in ? x0 + r : A.c.nx - 1;
        const double* pd = A.c.p + y * A.c.row_stride + xc;
        const uint8_t* pmk = arr ? A.m.arr + y * A.m.row_stride + xc : nullptr;
        for (int z0 = jz; z0 < NZP; z0 += L * kIn) {
            double vv[kIn];
            unsigned mk[kIn];
#pragma unroll
            for (int q = 0; q < kIn; ++q) {
                const int64_t zc = min((int64_t)(z0 + q * L), A.c.nz - 1);
                vv[q] = pd[zc * A.c.plane_stride];
                mk[q] = arr ? pmk[zc * A.m.plane_stride] : 1u;
            }
#pragma unroll
            for (int q = 0; q < kIn; ++q) {
                const int z = z0 + q * L;
                if (z < NZP) {
                    double v = vv[q];
                    bool ok = col_in && z < A.c.nz && pred64(A.m, v) && mk[q] != 0u;
                    if (A.center) { v = fabs(v - cen); ok = ok && (v == v); }
                    keys[z * TS + r] = ok ? fkey64(v) : kExcl;
                }
            }
        }
    }
    __syncthreads();
    const int half = (TS * NZP) >> 1;
    for (int k = 2; k <= NZP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int idx = t; idx < half; idx += 256) {
                const int ray = idx % TS, p = idx / TS;
                const int i = ((p / j) * 2 * j) + (p % j), l = i + j;
                const bool up = (i & k) == 0;
                const unsigned long long a = keys[i * TS + ray], b = keys[l * TS + ray];
                if ((a > b) == up) { keys[i * TS + ray] = b; keys[l * TS + ray] = a; }
            }
            __syncthreads();
        }
    }
    if (t < TS && x0 + t < A.c.nx) {
        const int ray = t;
        int lo = 0, hi = NZP;                                  // n = number of keys below the excluded one
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid * TS + ray] == kExcl) hi = mid; else lo = mid + 1; }
        const int n = lo;
        auto val = [&](int i) { return funkey64(keys[i * TS + ray]); };
        auto quantile = [&](int a, int cnt, double q) {        // numpy: virtual index q / 100 (cnt - 1), _lerp between its neighbours
            if (q == 50.0) {
                const int m = a + (cnt - 1) / 2;
                return (cnt & 1) ? val(m) : 0.5 * (val(m) + val(m + 1));
            }
            const double vi = q / 100.0 * (double)(cnt - 1);
            int p = (int)floor(vi);
            p = min(max(p, 0), cnt - 1);
            const int p1 = min(p + 1, cnt - 1);
            const double g = vi - (double)p, va = val(a + p), vb = val(a + p1), d = vb - va;
            return g >= 0.5 ? __dsub_rn(vb, __dmul_rn(d, 1.0 - g)) : __dadd_rn(va, __dmul_rn(d, g));   // (numpy's _lerp: a product, then a sum - never one fused operation)
        };
        // median of |x - med| over the window: the deviations below and above the median are two ascending sequences (walking
        // away from it) - their merge is walked up to the middle ranks
        auto mad = [&](int a, int cnt, double med) {
            int lo = a, hi = a + cnt;                          // first entry above the median
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (val(mid) > med) hi = mid; else lo = mid + 1; }
            int i = lo - 1, j = lo;                            // i walks down (<= med), j walks up (> med)
            const int r1 = (cnt - 1) / 2, r2 = cnt / 2;
            double d1 = 0.0, d2 = 0.0;
            for (int rank = 0; rank <= r2; ++rank) {
                const double dl = i >= a ? med - val(i) : INFINITY, dh = j < a + cnt ? val(j) - med : INFINITY;
                double d;
                if (dl <= dh) { d = dl; --i; } else { d = dh; ++j; }
                if (rank == r1) d1 = d;
                if (rank == r2) d2 = d;
            }
            return (cnt & 1) ? d1 : 0.5 * (d1 + d2);
        };
        if (MODE == 0) {
            A.out[y * A.c.nx + x0 + ray] = n > 0 ? quantile(0, n, A.q) * A.scale : NAN;
        } else {
            int a = 0, b = n, it = 0;
            while (b > a && (A.maxiters < 0 || it < A.maxiters)) {
                ++it;
                const int cnt = b - a;
                double sum = 0.0;
                for (int i = a; i < b; ++i) sum += val(i);
                const double mean = sum / (double)cnt;
                double ss = 0.0;
                for (int i = a; i < b; ++i) { const double dv = val(i) - mean; ss = fma(dv, dv, ss); }
                const double medw = quantile(a, cnt, 50.0);
                const double sd = A.spread_mad ? 1.482602218505602 * mad(a, cnt, medw) : sqrt(ss / (double)cnt);
                const double c = A.cen_mean ? mean : medw;
                const double lob = c - A.lo_s * sd, hib = c + A.hi_s * sd;
                int na = a, nb = b;
                while (na < nb && val(na) < lob) ++na;
                while (nb > na && val(nb - 1) > hib) --nb;
                // (a NaN bound - an infinite sample in the window - clips nothing: the comparisons are false, as numpy's)
                if (na == a && nb == b) break;
                a = na; b = nb;
            }
            s_wlo[ray] = b > a ? keys[a * TS + ray] : 1ull;      // empty window: lo > hi, nothing survives
            s_whi[ray] = b > a ? keys[(b - 1) * TS + ray] : 0ull;
        }
    }
    if (MODE == 1) {
        __syncthreads();
        if (col_in) {
            const unsigned long long wlo = s_wlo[r], whi = s_whi[r];
            for (int z = jz; z < A.c.nz; z += L) {
                double v;
                const bool ok = inc64(A.c, A.m, z, y, x0 + r, v);
                const unsigned long long k = ok ? fkey64(v) : kExcl;
                A.out[((int64_t)z * A.c.ny + y) * A.c.nx + x0 + r] = (k >= wlo && k <= whi) ? v : NAN;
            }
        }
    }
}

// Round 6: median / percentile / mad_std of float64 rays WITHOUT a sort.  The bitonic network moved ~720 bytes through LDS per
// sample (45 barrier-separated stages at 512 channels); here the keys of TS = min(16, 8192 / NZP) adjacent rays go to LDS once
// (64 KB: 128 contiguous bytes per plane and block at TS = 16) and the lower order statistic of numpy's rule is pinned down one
// key BYTE at a time: eight passes count, for every ray, the keys that share the bytes found so far by their next byte (LDS
// atomics on 16 x 128 packed 16-bit counters), 256 / TS threads per ray add the counters up and one of them walks to the bin
// that holds the rank.  A ninth pass gives the number of keys <= the one found and the smallest key above it: the upper order
// statistic is one or the other.  The results are the sorted rays' - numpy's `_lerp` of the same two samples - bit for bit.
constexpr int kSelKeys64 = 8192, kSelRays64 = 16;
__global__ __launch_bounds__(256) void select64_kernel(const Sort64Args A) {
    __shared__ unsigned long long keys[kSelKeys64];          // [sample][ray]
    __shared__ unsigned hist[kSelRays64 * 128];              // [ray][bin & 127]: low half = bins 0 .. 127, high half = bins 128 .. 255
    __shared__ unsigned s_part[256];                         // [ray][thread of the ray]
    __shared__ unsigned long long s_prefix[kSelRays64], s_next[kSelRays64];
    __shared__ int s_rank[kSelRays64], s_n[kSelRays64], s_le[kSelRays64];
    const int t = threadIdx.x, TS = A.ts, NZP = A.nzp, L = 256 / TS;
    const int64_t tiles_x = (A.c.nx + TS - 1) / TS;
    const int64_t y = blockIdx.x / tiles_x, x0 = (blockIdx.x % tiles_x) * TS;
    const int r = t % TS, jz = t / TS;
    const bool col_in = x0 + r < A.c.nx;
    const double cen = (A.center && col_in) ? A.center[y * A.c.nx + x0 + r] : 0.0;
    {   // (eight samples requested together per lane, their channels clamped into the ray)
        constexpr int kIn = 8;
        const bool arr = (A.m.flags & SPC_MASK_ARRAY) != 0;
        const int64_t xc = col_in ? x0 + r : A.c.nx - 1;
        const double* pd = A.c.p + y * A.c.row_stride + xc;
        const uint8_t* pmk = arr ? A.m.arr + y * A.m.row_stride + xc : nullptr;
        for (int z0 = jz; z0 < NZP; z0 += L * kIn) {
            double vv[kIn];
            unsigned mk[kIn];
#pragma unroll
            for (int q = 0; q < kIn; ++q) {
                const int64_t zc = min((int64_t)(z0 + q * L), A.c.nz - 1);
                vv[q] = pd[zc * A.c.plane_stride];
                mk[q] = arr ? pmk[zc * A.m.plane_stride] : 1u;
            }
#pragma unroll
            for (int q = 0; q < kIn; ++q) {
                const int z = z0 + q * L;
                if (z < NZP) {
                    double v = vv[q];
                    bool ok = col_in && z < A.c.nz && pred64(A.m, v) && mk[q] != 0u;
                    if (A.center) { v = fabs(v - cen); ok = ok && (v == v); }
                    keys[z * TS + r] = ok ? fkey64(v) : kExcl;
                }
            }
        }
    }
    if (t < TS) { s_prefix[t] = 0ull; s_rank[t] = 0; s_n[t] = 0; s_le[t] = 0; s_next[t] = kExcl; }
    const int nkeys = TS * NZP;
    const int ray_s = t / L, l = t % L;                        // scan: L threads per ray, TS bins each
    for (int d = 7; d >= 0; --d) {
        const int shift = 8 * d;
        for (int i = t; i < TS * 128; i += 256) hist[i] = 0u;
        __syncthreads();
        for (int i = t; i < nkeys; i += 256) {
            const unsigned long long k = keys[i];
            const int ray = i % TS;
            // (the bytes above the current one: equal to the prefix found so far; d = 7: every key but the excluded ones)
            const bool match = k != kExcl && (d == 7 || (k >> (shift + 8)) == (s_prefix[ray] >> (shift + 8)));
            if (match) {
                const unsigned b = (unsigned)(k >> shift) & 255u;
                atomicAdd(&hist[ray * 128 + (b & 127u)], (b & 128u) ? 65536u : 1u);
            }
        }
        __syncthreads();
        {   // bins [l TS, (l + 1) TS) of ray ray_s: all in one half of the packed counters
            unsigned sum = 0;
            const int b0 = l * TS;
            for (int b = b0; b < b0 + TS; ++b) { const unsigned w = hist[ray_s * 128 + (b & 127)]; sum += (b & 128) ? (w >> 16) : (w & 0xffffu); }
            s_part[ray_s * L + l] = sum;
        }
        __syncthreads();
        if (l == 0) {
            int rank = s_rank[ray_s];
            if (d == 7) {                                      // the ray's count, and the rank of the lower order statistic
                int n = 0;
                for (int g = 0; g < L; ++g) n += (int)s_part[ray_s * L + g];
                s_n[ray_s] = n;
                if (A.q == 50.0) rank = (n - 1) / 2;
                else { rank = (int)floor(A.q / 100.0 * (double)(n - 1)); rank = min(max(rank, 0), n - 1); }
                if (n <= 0) rank = 0;
            }
            int g = 0;
            while (g < L - 1 && rank >= (int)s_part[ray_s * L + g]) { rank -= (int)s_part[ray_s * L + g]; ++g; }
            int b = g * TS;
            for (;;) {
                const unsigned w = hist[ray_s * 128 + (b & 127)];
                const int c = (int)((b & 128) ? (w >> 16) : (w & 0xffffu));
                if (rank < c || b == g * TS + TS - 1) break;
                rank -= c; ++b;
            }
            s_rank[ray_s] = rank;
            s_prefix[ray_s] |= (unsigned long long)b << shift;
        }
        __syncthreads();
    }
    for (int i = t; i < nkeys; i += 256) {                      // keys <= the one found, and the smallest one above it
        const unsigned long long k = keys[i];
        const int ray = i % TS;
        if (k != kExcl) {
            if (k <= s_prefix[ray]) atomicAdd(&s_le[ray], 1);
            else atomicMin(&s_next[ray], k);
        }
    }
    __syncthreads();
    if (t < TS && x0 + t < A.c.nx) {
        const int n = s_n[t];
        double res = NAN;
        if (n > 0) {
            const unsigned long long klo = s_prefix[t];
            int p;
            double g = 0.0;
            if (A.q == 50.0) { p = (n - 1) / 2; g = (n & 1) ? 0.0 : 0.5; }
            else { const double vi = A.q / 100.0 * (double)(n - 1); p = min(max((int)floor(vi), 0), n - 1); g = vi - (double)p; }
            const int p1 = min(p + 1, n - 1);
            const unsigned long long khi = (p1 == p || p1 < s_le[t]) ? klo : s_next[t];
            const double va = funkey64(klo), vb = funkey64(khi);
            if (A.q == 50.0) res = (n & 1) ? va : 0.5 * (va + vb);
            else { const double d = vb - va; res = g >= 0.5 ? __dsub_rn(vb, __dmul_rn(d, 1.0 - g)) : __dadd_rn(va, __dmul_rn(d, g)); }   // (numpy's _lerp, unfused)
        }
        A.out[y * A.c.nx + x0 + t] = res * A.scale;
    }
}

// Rays of up to 1024 samples: the keys stay in REGISTERS (round 6, second form).  A block = 16 adjacent rays (128 contiguous
// bytes per plane), a wave = 4 of them, and the 16 lanes that share a ray are ONE DPP row (lane = 16 ray + slice, slice s holds
// the samples z = s + 16 i): whatever the lanes of a ray have to agree on - the sixteen counters of a digit pass, the count of
// keys below the one found, the smallest key above it - meets through four row rotations (v_add / v_min with row_ror 8, 4, 2, 1),
// with no LDS, no barrier, no atomics; every wave runs on its own.  Radix 16: sixteen passes over the KPL key registers of a
// lane (a bit-field extract, a prefix compare on the word that holds the digit, 8-bit counters in four packed words), after
// which every lane of the row walks the sixteen totals to the digit.
template <int N> struct U32Arr { unsigned v[N]; };
__device__ __forceinline__ unsigned row_ror_u32(unsigned v, int n) {       // lane i of a row of 16 takes lane (i + n) mod 16 ... of its row
    switch (n) {
        case 8: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);
        case 4: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);
        case 2: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);
        default: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);
    }
}
__device__ __forceinline__ unsigned swap16_u32(unsigned v) {           // lane i takes lane i ^ 16 (ds_swizzle, bit mode: and 0x1f, xor 0x10)
    return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401f);
}
// LPR = 16 | 32 lanes per ray: one DPP row, or two neighbouring rows of the same wave (rays of 513 .. 1024 samples: 64 keys per lane
// do not fit the registers)
template <int LPR>
__device__ __forceinline__ unsigned row_sum_u32(unsigned v) {
    v += row_ror_u32(v, 8); v += row_ror_u32(v, 4); v += row_ror_u32(v, 2); v += row_ror_u32(v, 1);
    if (LPR == 32) v += swap16_u32(v);
    return v;
}
template <int LPR>
__device__ __forceinline__ unsigned long long row_min_u64(unsigned long long v) {
#pragma unroll
    for (int n = 8; n >= 1; n >>= 1) {
        const unsigned lo = row_ror_u32((unsigned)v, n), hi = row_ror_u32((unsigned)(v >> 32), n);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    if (LPR == 32) {
        const unsigned long long o = ((unsigned long long)swap16_u32((unsigned)(v >> 32)) << 32) | swap16_u32((unsigned)v);
        v = o < v ? o : v;
    }
    return v;
}
// the one read of the cube: lane (ray, slice s) takes the samples z = s + LPR i of its ray as order-preserving keys (hi / lo words);
// excluded, NaN and out-of-range samples become the all-ones key, which no valid sample maps to and which sorts last
template <int KPL, int LPR>
__device__ __forceinline__ void row_load_keys(const Sort64Args& A, int64_t y, int64_t xc, bool col_in, int sl, double cen,
                                              unsigned (&khi)[KPL], unsigned (&klo)[KPL]) {
    const bool arr = (A.m.flags & SPC_MASK_ARRAY) != 0;
    const double* pd = A.c.p + y * A.c.row_stride + xc;
    const uint8_t* pmk = arr ? A.m.arr + y * A.m.row_stride + xc : nullptr;
    constexpr int CH = KPL < 16 ? KPL : 16;                  // samples requested together per lane (all 64 at once: 128 registers of loads in flight)
#pragma unroll
    for (int i0 = 0; i0 < KPL; i0 += CH) {
        double vv[CH];
        unsigned mk[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int64_t zc = min((int64_t)(sl + LPR * (i0 + q)), A.c.nz - 1);
            vv[q] = pd[zc * A.c.plane_stride];
            mk[q] = arr ? pmk[zc * A.m.plane_stride] : 1u;
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int i = i0 + q;
            double v = vv[q];
            bool ok = col_in && (sl + LPR * i) < A.c.nz && pred64(A.m, v) && mk[q] != 0u;
            if (A.center) { v = fabs(v - cen); ok = ok && (v == v); }
            const unsigned long long k = ok ? fkey64(v) : kExcl;
            khi[i] = (unsigned)(k >> 32); klo[i] = (unsigned)k;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the key of 0-based rank `rank` (< n = the ray's valid keys) among the keys of the row's lanes: sixteen digit passes, HI selects
// the word the digit lies in (static per unrolled body, uniform per pass).  The excluded key - all ones - is counted like any
// other: it is the LARGEST key, so it never moves the bin that holds a rank below n (leaving its test out saves two of ten
// instructions per key and pass).  The descent stops as soon as every ray of the wave has ONE key left under its prefix
// (distinct samples: after nine or ten of the sixteen passes): that key is the smallest - the only - key of the row that
// matches the prefix.  All lanes of the wave call it (wave-uniform control flow); every lane of a row gets the row's key.
template <int KPL, int LPR>
__device__ __forceinline__ unsigned long long row_descend(const unsigned (&khi)[KPL], const unsigned (&klo)[KPL], int rank, int n) {
    int left = n;                                            // keys of the ray under the prefix found so far
    unsigned phi = 0u, plo = 0u;                             // the key's bits found so far
    auto pass = [&](auto hi_c, const int shift) {           // shift: bit offset of the digit inside its word
        constexpr bool HI = decltype(hi_c)::value;
        const unsigned above = shift >= 28 ? 0u : (0xffffffffu << (shift + 4));       // the bits of this word above the digit
        const unsigned pre = (HI ? phi : plo) & above;
        unsigned long long c4 = 0ull, ev = 0ull, od = 0ull;  // sixteen 4-bit counters; 8-bit counters of the even / odd bins
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const unsigned w = HI ? khi[i] : klo[i];
            bool m = (w & above) == pre;
            if (!HI) m = m && (khi[i] == phi);
            const unsigned b4 = __builtin_amdgcn_ubfe(w, (unsigned)shift, 4u) << 2;
            c4 += (unsigned long long)(m ? 1u : 0u) << b4;
            if ((i % 15) == 14 || i == KPL - 1) {            // a 4-bit counter holds 15
                ev += c4 & 0x0f0f0f0f0f0f0f0full;
                od += (c4 >> 4) & 0x0f0f0f0f0f0f0f0full;
                c4 = 0ull;
            }
        }
        // the ray's totals: 16-bit fields (<= 1024 keys per ray) summed over the row.  word[h][g] with h = even / odd bins,
        // g = 0 .. 3: fields (bits 0-15, bits 16-31) = 8-bit fields (f, f + 2) of the h-counters' word (g >> 1), f = g & 1
        unsigned tw[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned long long src = h ? od : ev;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const unsigned word = (unsigned)(src >> (32 * (g >> 1)));
                tw[h][g] = row_sum_u32<LPR>((word >> (8 * (g & 1))) & 0x00ff00ffu);
            }
        }
        unsigned digit = 15u;
        bool found = false;
        left = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            // bin b = 8-bit field f = b >> 1 of the (b & 1)-counters: word f >> 2, field f & 3 = (f & 1) + 2 * ((f >> 1) & 1)
            const int f = b >> 1, g = 2 * (f >> 2) + (f & 1), half = (f >> 1) & 1;
            const int cb = (int)((tw[b & 1][g] >> (16 * half)) & 0xffffu);
            const bool here = !found && rank < cb;
            left = here ? cb : left;
            digit = here ? (unsigned)b : digit;
            rank = (found || here) ? rank : rank - cb;
            found = found || here;
        }
        if (HI) phi |= digit << shift; else plo |= digit << shift;
    };
    // The leading digits EVERY valid key of the ray shares need no pass (positive samples within a factor of a few share their sign,
    // exponent and then some: the bench's 1000 +- 1 shares five of the ten digits a descent otherwise walks): AND and OR of the
    // ray's keys differ where its keys do; the wave starts at the first digit that differs in ANY of its rays.
    int start = 0;
    {
        unsigned ah = 0xffffffffu, al = 0xffffffffu, oh = 0u, ol = 0u;
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const bool ex = (khi[i] & klo[i]) == 0xffffffffu;      // (the excluded key is all ones: neutral for AND, kept out of OR)
            ah &= khi[i]; al &= klo[i];
            oh |= ex ? 0u : khi[i]; ol |= ex ? 0u : klo[i];
        }
        ah &= row_ror_u32(ah, 8); al &= row_ror_u32(al, 8); oh |= row_ror_u32(oh, 8); ol |= row_ror_u32(ol, 8);
        ah &= row_ror_u32(ah, 4); al &= row_ror_u32(al, 4); oh |= row_ror_u32(oh, 4); ol |= row_ror_u32(ol, 4);
        ah &= row_ror_u32(ah, 2); al &= row_ror_u32(al, 2); oh |= row_ror_u32(oh, 2); ol |= row_ror_u32(ol, 2);
        ah &= row_ror_u32(ah, 1); al &= row_ror_u32(al, 1); oh |= row_ror_u32(oh, 1); ol |= row_ror_u32(ol, 1);
        if (LPR == 32) { ah &= swap16_u32(ah); al &= swap16_u32(al); oh |= swap16_u32(oh); ol |= swap16_u32(ol); }
        const unsigned dh = ah ^ oh, dl = al ^ ol;
        int shared = dh ? (__builtin_clz(dh) >> 2) : 8 + (dl ? (__builtin_clz(dl) >> 2) : 8);
        if (n <= 1) shared = 16;                             // nothing to tell apart: this ray holds nobody back
        start = min(__builtin_amdgcn_readlane(shared, 0), __builtin_amdgcn_readlane(shared, 32));
        if (LPR == 16) start = min(start, min(__builtin_amdgcn_readlane(shared, 16), __builtin_amdgcn_readlane(shared, 48)));
        if (start >= 16) return ((unsigned long long)ah << 32) | al;     // every ray of the wave: one distinct key (or none)
        phi = start >= 8 ? ah : (start > 0 ? (ah & (0xffffffffu << (32 - 4 * start))) : 0u);
        plo = start > 8 ? (al & (0xffffffffu << (64 - 4 * start))) : 0u;
    }
    int stop_hi = -1, stop_lo = -1;                          // the shift of the last pass made in either word (-1: none)
    for (int shift = 28 - 4 * start; shift >= 0; shift -= 4) {          // (start >= 8: no pass in the high word)
        pass(std::integral_constant<bool, true>{}, shift);
        stop_hi = shift;
        if (__all(left <= 1)) break;
    }
    if (start >= 8 || !__all(left <= 1)) {
        for (int shift = start > 8 ? 28 - 4 * (start - 8) : 28; shift >= 0; shift -= 4) {
            pass(std::integral_constant<bool, false>{}, shift);
            stop_lo = shift;
            if (__all(left <= 1)) break;
        }
    }
    if (stop_lo != 0) {                                      // stopped early: pick the one key that is left up
        const unsigned mh = stop_lo >= 0 ? 0xffffffffu : (0xffffffffu << stop_hi);
        const unsigned ml = stop_lo >= 0 ? (0xffffffffu << stop_lo) : 0u;
        unsigned long long cand = kExcl;
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const bool mt = ((khi[i] & mh) == phi) && ((klo[i] & ml) == plo);
            const unsigned long long k = ((unsigned long long)khi[i] << 32) | klo[i];
            cand = (mt && k < cand) ? k : cand;
        }
        cand = row_min_u64<LPR>(cand);
        if (left == 1) { phi = (unsigned)(cand >> 32); plo = (unsigned)cand; }   // (a ray with duplicates left ran all sixteen passes)
    }
    return ((unsigned long long)phi << 32) | plo;
}

// keys <= the one found, and the smallest key above it (the upper order statistic of numpy's rule is one or the other)
template <int KPL, int LPR>
__device__ __forceinline__ void row_le_next(const unsigned (&khi)[KPL], const unsigned (&klo)[KPL], unsigned long long kfound,
                                            unsigned& le, unsigned long long& nxt) {
    unsigned l = 0;
    unsigned long long nx = kExcl;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const unsigned long long k = ((unsigned long long)khi[i] << 32) | klo[i];
        l += (k <= kfound) ? 1u : 0u;
        nx = (k > kfound && k < nx) ? k : nx;
    }
    le = row_sum_u32<LPR>(l);
    nxt = row_min_u64<LPR>(nx);
}
template <int KPL, int LPR>
__device__ __forceinline__ int row_count_valid(const unsigned (&khi)[KPL], const unsigned (&klo)[KPL]) {
    unsigned nloc = 0;
#pragma unroll
    for (int i = 0; i < KPL; ++i) nloc += ((khi[i] & klo[i]) != 0xffffffffu) ? 1u : 0u;      // (valid keys never are all ones)
    return (int)row_sum_u32<LPR>(nloc);
}

template <int KPL, int LPR>
__global__ __launch_bounds__(16 * LPR) void select64_reg_kernel(const Sort64Args A) {
    constexpr int RW = 64 / LPR;                             // rays per wave
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, rr = lane / LPR, sl = lane % LPR;
    const int64_t tiles_x = (A.c.nx + 15) / 16;
    const int64_t y = blockIdx.x / tiles_x, x0 = (blockIdx.x % tiles_x) * 16;
    const int64_t x = x0 + RW * wave + rr;
    const bool col_in = x < A.c.nx;
    const int64_t xc = col_in ? x : A.c.nx - 1;
    const double cen = A.center ? A.center[y * A.c.nx + xc] : 0.0;
    unsigned khi[KPL], klo[KPL];
    row_load_keys<KPL, LPR>(A, y, xc, col_in, sl, cen, khi, klo);
    // valid samples of the ray, and the rank of the lower order statistic (numpy's rule)
    const int n = row_count_valid<KPL, LPR>(khi, klo);
    int p = 0;
    double g = 0.0;
    if (A.q == 50.0) { p = (n - 1) / 2; g = (n & 1) ? 0.0 : 0.5; }
    else { const double vi = A.q / 100.0 * (double)(n - 1); p = min(max((int)floor(vi), 0), max(n - 1, 0)); g = vi - (double)p; }
    if (n <= 0) p = 0;
    const unsigned long long kfound = row_descend<KPL, LPR>(khi, klo, p, n);
    unsigned le;
    unsigned long long nxt;
    row_le_next<KPL, LPR>(khi, klo, kfound, le, nxt);
    if (sl == 0 && col_in) {
        double res = NAN;
        if (n > 0) {
            const int p1 = min(p + 1, n - 1);
            const unsigned long long kh = (p1 == p || p1 < (int)le) ? kfound : nxt;
            const double va = funkey64(kfound), vb = funkey64(kh);
            if (A.q == 50.0) res = (n & 1) ? va : 0.5 * (va + vb);
            else { const double d = vb - va; res = g >= 0.5 ? __dsub_rn(vb, __dmul_rn(d, 1.0 - g)) : __dadd_rn(va, __dmul_rn(d, g)); }
        }
        A.out[y * A.c.nx + x] = res * A.scale;
    }
}

// sigma_clip_spectrally on the same resident keys (centre = median | mean, spread = std): an iteration is the window's count, its
// mean and its sum of squared deviations (float64 lane partials added over the row in a fixed order: the result does not depend
// on scheduling), the median by row_descend, astropy's bounds; a clipped sample's key becomes the excluded key.  A wave iterates
// until none of ITS rays changes (a converged ray recomputes the same bounds and clips nothing) or maxiters is reached.
template <int LPR>
__device__ __forceinline__ double row_sum_f64(double v) {
#pragma unroll
    for (int n = 8; n >= 1; n >>= 1) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        const unsigned lo = row_ror_u32((unsigned)u, n), hi = row_ror_u32((unsigned)(u >> 32), n);
        v += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    if (LPR == 32) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(v);
        v += __longlong_as_double((long long)(((unsigned long long)swap16_u32((unsigned)(u >> 32)) << 32) | swap16_u32((unsigned)u)));
    }
    return v;
}
template <int KPL, int LPR>
__global__ __launch_bounds__(16 * LPR) void sigma_clip64_reg_kernel(const Sort64Args A) {
    constexpr int RW = 64 / LPR;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, rr = lane / LPR, sl = lane % LPR;
    const int64_t tiles_x = (A.c.nx + 15) / 16;
    const int64_t y = blockIdx.x / tiles_x, x0 = (blockIdx.x % tiles_x) * 16;
    const int64_t x = x0 + RW * wave + rr;
    const bool col_in = x < A.c.nx;
    const int64_t xc = col_in ? x : A.c.nx - 1;
    unsigned khi[KPL], klo[KPL];
    row_load_keys<KPL, LPR>(A, y, xc, col_in, sl, 0.0, khi, klo);
    int it = 0;
    for (;;) {
        if (!(A.maxiters < 0 || it < A.maxiters)) break;     // (uniform)
        ++it;
        const int cnt = row_count_valid<KPL, LPR>(khi, klo);
        double sum = 0.0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const unsigned long long k = ((unsigned long long)khi[i] << 32) | klo[i];
            sum += (k != kExcl) ? funkey64(k) : 0.0;
        }
        const double mean = row_sum_f64<LPR>(sum) / (double)cnt;
        double ss = 0.0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const unsigned long long k = ((unsigned long long)khi[i] << 32) | klo[i];
            const double dv = funkey64(k) - mean;
            ss = (k != kExcl) ? fma(dv, dv, ss) : ss;
        }
        const double sd = sqrt(row_sum_f64<LPR>(ss) / (double)cnt);
        double c = mean;
        if (!A.cen_mean) {                                   // (uniform)
            const int p = max((cnt - 1) / 2, 0);
            const unsigned long long kf = row_descend<KPL, LPR>(khi, klo, p, cnt);
            unsigned le;
            unsigned long long nxt;
            row_le_next<KPL, LPR>(khi, klo, kf, le, nxt);
            const unsigned long long kh = ((cnt & 1) || p + 1 < (int)le) ? kf : nxt;
            c = (cnt & 1) ? funkey64(kf) : 0.5 * (funkey64(kf) + funkey64(kh));
        }
        const double lob = c - A.lo_s * sd, hib = c + A.hi_s * sd;
        unsigned changed = 0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const unsigned long long k = ((unsigned long long)khi[i] << 32) | klo[i];
            const double v = funkey64(k);
            // (a NaN bound - an infinite sample in the window - clips nothing: the comparisons are false, as numpy's)
            const bool out = (k != kExcl) && (cnt > 0) && ((v < lob) || (v > hib));
            khi[i] = out ? 0xffffffffu : khi[i]; klo[i] = out ? 0xffffffffu : klo[i];
            changed |= out ? 1u : 0u;
        }
        if (!__any(changed != 0u)) break;                    // none of this wave's rays changed
    }
    if (col_in) {
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int z = sl + LPR * i;
            if (z < A.c.nz) {
                const unsigned long long k = ((unsigned long long)khi[i] << 32) | klo[i];
                A.out[((int64_t)z * A.c.ny + y) * A.c.nx + x] = (k != kExcl) ? funkey64(k) : NAN;
            }
        }
    }
}

static int sort64_launch(Sort64Args& A, const spc_cube_f64* cube, bool clip, hipStream_t st) {
    if (cube->nz > 4096) {
        spc_set_error("rays of more than 4096 samples have no float64 order statistics (got %lld)", (long long)cube->nz);
        return SPC_ERR_UNSUPPORTED;
    }
    int nzp = 2;
    while (nzp < cube->nz) nzp <<= 1;
    // (SPC_SELECT64: 0 = the order statistics from sorted rays as in round 5; the clip loop always sorts)
    static const bool radix = [] { const char* e = getenv("SPC_SELECT64"); return e ? atoi(e) != 0 : true; }();
    const bool sel = !clip && radix;
    A.nzp = nzp; A.ts = sel ? std::min(kSelRays64, kSelKeys64 / nzp) : std::min(64, kSortKeys / nzp);
    const int64_t nb = ((cube->nx + A.ts - 1) / A.ts) * cube->ny;
    SPC_REQUIRE(nb < (1LL << 31), "map too large for one launch");
    // (SPC_SELECT64 = 2, the default: rays of up to 1024 samples keep their keys in registers; 1: the keys in LDS for every length)
    static const int radix_form = [] { const char* e = getenv("SPC_SELECT64"); return e ? atoi(e) : 2; }();
    if (clip && radix_form >= 2 && cube->nz <= 1024 && !A.spread_mad) {      // (stdfunc = mad_std: a second descent over |x - median| - the sorted form below)
        const int64_t nbr = ((cube->nx + 15) / 16) * cube->ny;
        SPC_REQUIRE(nbr < (1LL << 31), "map too large for one launch");
        const int kpl = (int)((cube->nz + 15) / 16);
        dim3 grid((unsigned)nbr);
        if (kpl <= 8) hipLaunchKernelGGL((sigma_clip64_reg_kernel<8, 16>), grid, dim3(256), 0, st, A);
        else if (kpl <= 16) hipLaunchKernelGGL((sigma_clip64_reg_kernel<16, 16>), grid, dim3(256), 0, st, A);
        else if (kpl <= 32) hipLaunchKernelGGL((sigma_clip64_reg_kernel<32, 16>), grid, dim3(256), 0, st, A);
        else hipLaunchKernelGGL((sigma_clip64_reg_kernel<32, 32>), grid, dim3(512), 0, st, A);
        SPC_LAUNCH_CHECK();
        return SPC_OK;
    }
    if (sel && radix_form >= 2 && cube->nz <= 1024) {
        const int64_t nbr = ((cube->nx + 15) / 16) * cube->ny;
        SPC_REQUIRE(nbr < (1LL << 31), "map too large for one launch");
        const int kpl = (int)((cube->nz + 15) / 16);
        dim3 grid((unsigned)nbr);
        if (kpl <= 8) hipLaunchKernelGGL((select64_reg_kernel<8, 16>), grid, dim3(256), 0, st, A);
        else if (kpl <= 16) hipLaunchKernelGGL((select64_reg_kernel<16, 16>), grid, dim3(256), 0, st, A);
        else if (kpl <= 32) hipLaunchKernelGGL((select64_reg_kernel<32, 16>), grid, dim3(256), 0, st, A);
        else hipLaunchKernelGGL((select64_reg_kernel<32, 32>), grid, dim3(512), 0, st, A);      // 32 lanes per ray, 32 keys per lane
        SPC_LAUNCH_CHECK();
        return SPC_OK;
    }
    if (clip) hipLaunchKernelGGL(sort64_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, A);
    else if (sel) hipLaunchKernelGGL(select64_kernel, dim3((unsigned)nb), dim3(256), 0, st, A);
    else hipLaunchKernelGGL(sort64_kernel<0>, dim3((unsigned)nb), dim3(256), 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

// ---- small elementwise helpers --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void narrow64_kernel(const Cube64 C, float* out, int64_t out_row_stride, int64_t out_plane_stride) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t y = blockIdx.y, z = blockIdx.z;
    if (x >= C.nx) return;
    out[z * out_plane_stride + y * out_row_stride + x] = (float)C.p[z * C.plane_stride + y * C.row_stride + x];
}
__global__ __launch_bounds__(256) void include64_kernel(const Cube64 C, const MaskDev64 M, int nan_excluded, uint8_t* out) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t y = blockIdx.y, z = blockIdx.z;
    if (x >= C.nx) return;
    double v;
    bool ok = inc64(C, M, z, y, x, v);
    if (!nan_excluded && !(M.flags & ~SPC_MASK_ARRAY) && v != v) {
        // (without a predicate a NaN sample is included when its mask byte says so: the mask, not the data, is reported)
        ok = (M.flags & SPC_MASK_ARRAY) ? M.arr[z * M.plane_stride + y * M.row_stride + x] != 0 : true;
    }
    out[(z * C.ny + y) * C.nx + x] = ok ? 1 : 0;
}

// astropy: "The kernel can't be normalized, because its sum is close to zero" (convolve.py; the float32 entry points' check_kernel)
static int check_kernel_sum(const double* k, size_t n) {
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) sum += k[i];
    SPC_REQUIRE(!(sum < 1e-8 && sum > -1e-8) && sum >= 1.0 / 1e8, "The kernel can't be normalized, because its sum is close to zero");
    return SPC_OK;
}

static int cube64_args(const spc_cube_f64* cube, const spc_mask_f64* mask, Cube64* C, MaskDev64* M, bool any_order = false) {
    int rc = any_order ? check_cube64_any_order(cube) : check_cube64(cube);
    if (rc) return rc;
    rc = mask64_to_dev(mask, cube, M);
    if (rc) return rc;
    C->p = cube->d_data; C->nz = cube->nz; C->ny = cube->ny; C->nx = cube->nx;
    C->row_stride = cube->row_stride; C->plane_stride = cube->plane_stride;
    return SPC_OK;
}

}  // namespace

template <int R>
int launch_sring64(hipStream_t st, const Conv64Args& S, const double* h_kernel, int ntaps, unsigned* d_flag) {
    SRing64Args<R> A{};
    A.c = S.c; A.m = S.m; A.out = S.out; A.out_row_stride = S.out_row_stride; A.out_plane_stride = S.out_plane_stride;
    const int pad = (R - ntaps) / 2;                            // the taps centred in R entries
    for (int j = 0; j < R; ++j) A.k[j] = (j >= pad && j < pad + ntaps) ? h_kernel[j - pad] : 0.0;
    A.flag = d_flag;
    // the whole spectral axis per block when the map alone fills the chip, else chunks of >= 4 R outputs (R - 1 more inputs each)
    const int64_t nb = (S.c.ny * S.c.nx + 255) / 256;
    SPC_REQUIRE(nb < (1LL << 31), "map too large for one launch");
    int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>((2048 + nb - 1) / nb, S.c.nz / (4 * R)));
    A.zchunk = (int)((S.c.nz + nsplit - 1) / nsplit);
    nsplit = (S.c.nz + A.zchunk - 1) / A.zchunk;
    SPC_REQUIRE(nsplit <= 65535, "too many chunks");
    if (S.m.flags & SPC_MASK_ARRAY) hipLaunchKernelGGL((spectral64_ring_kernel<R, true>), dim3((unsigned)nb, (unsigned)nsplit), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((spectral64_ring_kernel<R, false>), dim3((unsigned)nb, (unsigned)nsplit), dim3(256), 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

template <int R>
int launch_ring64(hipStream_t st, const Sp64Args& S, const double* h_ky, int nky, const double* h_kx, int nkx, unsigned* d_flag) {
    Ring64Args<R> A{};
    A.c = S.c; A.m = S.m; A.out = S.out; A.out_row_stride = S.out_row_stride; A.out_plane_stride = S.out_plane_stride;
    const int py = (R - nky) / 2, px = (R - nkx) / 2;           // the taps centred in R entries; symmetric: the first half names them all
    for (int i = 0; i <= R / 2; ++i) {
        A.ky[i] = (i >= py) ? h_ky[i - py] : 0.0;
        A.kx[i] = (i >= px) ? h_kx[i - px] : 0.0;
    }
    double ksx = 0.0;
    for (int j = 0; j < nkx; ++j) ksx = fma(h_kx[nkx - 1 - j], 1.0, ksx);      // the x pass of a row of valid zeros, in its order
    A.ksx = ksx;
    A.flag = d_flag;
    A.nstrips = (int)((S.c.nx + 255) / 256);
    // bands: the whole column when the planes and strips alone fill the chip, else bands of >= 64 rows (2 (R / 2) halo rows each)
    const int64_t base = S.c.nz * A.nstrips;
    int64_t nb = std::max<int64_t>(1, std::min<int64_t>((2048 + base - 1) / base, (S.c.ny + 63) / 64));
    A.band_rows = (int)((S.c.ny + nb - 1) / nb);
    A.nbands = (int)((S.c.ny + A.band_rows - 1) / A.band_rows);
    const int64_t nblocks = base * A.nbands;
    SPC_REQUIRE(nblocks < (1ll << 31), "too many blocks");
    if (S.m.flags & SPC_MASK_ARRAY) hipLaunchKernelGGL((spatial64_ring_kernel<R, true>), dim3((unsigned)nblocks), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((spatial64_ring_kernel<R, false>), dim3((unsigned)nblocks), dim3(256), 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

size_t spc_ws_wide(int kind, int64_t nz, int64_t ny, int64_t nx, int64_t p0, int64_t p1) {
    switch (kind) {
        case SPC_WS_STATS_GLOBAL_F64: return spc_ws_round(sizeof(Rec64) * 4096) + spc_ws_round(5 * sizeof(double)) + 256;
        case SPC_WS_SPECTRAL_CONV_F64: return spc_ws_round(sizeof(double) * (size_t)(2 * std::max<int64_t>(p0, 1) + 30)) + 256 + 256;   // taps + the zero-padded table + the ring form's flag word
        case SPC_WS_SPATIAL_CONV_F64: {
            // taps + the (num, den) planes of a slab: at most 256 MiB (the caller keeps its scratch), at least one plane
            const size_t plane = (size_t)ny * (size_t)nx * 2 * sizeof(double);
            const size_t slab = std::max<size_t>(plane, std::min<size_t>((size_t)nz * plane, (size_t)1 << 28));
            return spc_ws_round(sizeof(double) * (size_t)(std::max<int64_t>(p0, 1) * std::max<int64_t>(p1, 1) + p0 + p1)) + spc_ws_round(slab) + 512 + 256;   // (+ the ring form's flag word)
        }
    }
    return 0;
}

extern "C" {

int spc_stats_global_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, double* h_stats,
                         void* d_workspace, size_t workspace_bytes) {
    Cube64 C; MaskDev64 M;
    int rc = cube64_args(cube, mask, &C, &M);
    if (rc) return rc;
    SPC_REQUIRE(h_stats != nullptr, "h_stats is NULL");
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    SpcWorkspace ws(d_workspace, workspace_bytes);
    const int nblocks = (int)std::min<int64_t>(4096, C.nz * C.ny);
    SPC_WS_TAKE(d_partial, ws, Rec64, 4096);
    SPC_WS_TAKE(d_out, ws, double, 5);
    const bool vec2 = !(C.nx & 1) && !(C.row_stride & 1) && !(C.plane_stride & 1) && !(((uintptr_t)C.p) & 15) &&
                      (!(M.flags & SPC_MASK_ARRAY) || (!(M.row_stride & 1) && !(M.plane_stride & 1) && !(((uintptr_t)M.arr) & 1)));
    if (vec2) hipLaunchKernelGGL(stats64_global_kernel<2>, dim3((unsigned)nblocks), dim3(256), 0, st, C, M, d_partial);
    else hipLaunchKernelGGL(stats64_global_kernel<1>, dim3((unsigned)nblocks), dim3(256), 0, st, C, M, d_partial);
    SPC_LAUNCH_CHECK();
    hipLaunchKernelGGL(stats64_finish_kernel, dim3(1), dim3(256), 0, st, d_partial, nblocks, d_out);
    SPC_LAUNCH_CHECK();
    SPC_HIP(hipMemcpyAsync(h_stats, d_out, 5 * sizeof(double), hipMemcpyDeviceToHost, st));
    SPC_HIP(hipStreamSynchronize(st));
    return SPC_OK;
}

int spc_stats_axis_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, int axis,
                       const spc_stats_outputs_f64* out) {
    Cube64 C; MaskDev64 M;
    int rc = cube64_args(cube, mask, &C, &M);
    if (rc) return rc;
    SPC_REQUIRE(out != nullptr, "outputs struct is NULL");
    SPC_REQUIRE(axis >= 0 && axis <= 2, "axis must be 0, 1 or 2, got %d", axis);
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    const StatOut64 O{out->d_count, out->d_min, out->d_max, out->d_sum, out->d_sumsq};
    if (axis == 2) {
        const int64_t nb = (C.nz * C.ny + 3) / 4;
        SPC_REQUIRE(nb < (1LL << 31), "too many rows for one launch");
        hipLaunchKernelGGL(stats64_rows_kernel, dim3((unsigned)nb), dim3(256), 0, st, C, M, O);
    } else {
        const int64_t n = (axis == 0 ? C.ny : C.nz) * C.nx, nb = (n + 255) / 256;
        SPC_REQUIRE(nb < (1LL << 31), "map too large for one launch");
        if (axis == 0) hipLaunchKernelGGL(stats64_march_kernel<0>, dim3((unsigned)nb), dim3(256), 0, st, C, M, O);
        else hipLaunchKernelGGL(stats64_march_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, C, M, O);
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_spectral_conv_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, const double* h_kernel,
                          int ntaps, double* d_out, int64_t out_row_stride, int64_t out_plane_stride, void* d_workspace,
                          size_t workspace_bytes) {
    Conv64Args A{};
    int rc = cube64_args(cube, mask, &A.c, &A.m);
    if (rc) return rc;
    SPC_REQUIRE(h_kernel && d_out, "NULL pointer argument");
    SPC_REQUIRE(ntaps >= 1 && (ntaps & 1) && ntaps <= 8191, "the kernel needs an odd number of taps in [1, 8191], got %d", ntaps);
    rc = check_kernel_sum(h_kernel, (size_t)ntaps);
    if (rc) return rc;
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    SpcWorkspace ws(d_workspace, workspace_bytes);
    SPC_WS_TAKE(d_k, ws, double, 2 * ntaps + 30);
    {
        std::vector<double> hk((size_t)(2 * ntaps + 30), 0.0);
        for (int i = 0; i < ntaps; ++i) { hk[(size_t)i] = h_kernel[i]; hk[(size_t)(ntaps + 15 + i)] = h_kernel[i]; }
        SPC_HIP(spc_table_upload(d_k, hk.data(), sizeof(double) * hk.size(), st));
    }
    A.kern = d_k; A.kpad = d_k + ntaps; A.ntaps = ntaps; A.out = d_out;
    A.out_row_stride = out_row_stride ? out_row_stride : cube->nx;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : cube->ny * A.out_row_stride;
    A.gate = nullptr;
    // up to 33 taps, none negative, centre positive: the ring form (SPC_SPECTRAL64_RING=0: the runs of 16)
    const bool ring_on = [] { const char* e = getenv("SPC_SPECTRAL64_RING"); return e ? atoi(e) != 0 : true; }();      // (read per call: the tests compare both forms in one process)
    bool ring = ring_on && ntaps <= 33 && h_kernel[ntaps / 2] > 0.0;
    for (int i = 0; ring && i < ntaps; ++i) ring = h_kernel[i] >= 0.0;
    if (ring) {
        unsigned* d_flag = nullptr;
        if (!(A.m.flags & SPC_MASK_FINITE)) {                   // the mask admits infinite samples: see spectral64_ring_kernel
            SPC_WS_TAKE(d_f, ws, unsigned, 64);
            d_flag = d_f;
            SPC_HIP(hipMemsetAsync(d_flag, 0, sizeof(unsigned), st));
        }
        if (ntaps <= 17) rc = launch_sring64<17>(st, A, h_kernel, ntaps, d_flag);
        else rc = launch_sring64<33>(st, A, h_kernel, ntaps, d_flag);
        if (rc) return rc;
        if (!d_flag) return SPC_OK;
        A.gate = d_flag;                                        // the kernel below runs only if the ring kernel raised the flag
    }
    const int64_t nb = (cube->ny * cube->nx + 255) / 256, runs = (cube->nz + kRun - 1) / kRun;
    SPC_REQUIRE(nb < (1LL << 31) && runs <= 65535, "cube too large for one launch (convolve a slab of channels / rows)");
    hipLaunchKernelGGL(spectral_conv64_kernel, dim3((unsigned)nb, (unsigned)runs), dim3(256), 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_spatial_conv_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, const double* h_ky, int nky,
                         const double* h_kx, int nkx, int separable, double* d_out, int64_t out_row_stride, int64_t out_plane_stride,
                         void* d_workspace, size_t workspace_bytes) {
    Sp64Args A{};
    int rc = cube64_args(cube, mask, &A.c, &A.m);
    if (rc) return rc;
    SPC_REQUIRE(h_ky && d_out && (!separable || h_kx), "NULL pointer argument");
    SPC_REQUIRE(nky >= 1 && (nky & 1) && nkx >= 1 && (nkx & 1) && nky <= 1023 && nkx <= 1023,
                "kernel axes must be odd and at most 1023 (got %d x %d)", nky, nkx);
    SPC_REQUIRE(cube->ny <= 65535, "too many rows for one launch");
    if (separable) {                                            // (the sum of an outer product is the product of the sums)
        double sy = 0.0, sx = 0.0;
        for (int i = 0; i < nky; ++i) sy += h_ky[i];
        for (int i = 0; i < nkx; ++i) sx += h_kx[i];
        const double prod = sy * sx;
        rc = check_kernel_sum(&prod, 1);
    } else {
        rc = check_kernel_sum(h_ky, (size_t)nky * (size_t)nkx);
    }
    if (rc) return rc;
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    SpcWorkspace ws(d_workspace, workspace_bytes);
    const size_t ntab = separable ? (size_t)(nky + nkx) : (size_t)nky * (size_t)nkx;
    SPC_WS_TAKE(d_k, ws, double, ntab);
    if (separable) {
        SPC_HIP(spc_table_upload(d_k, h_ky, sizeof(double) * (size_t)nky, st));
        SPC_HIP(spc_table_upload(d_k + nky, h_kx, sizeof(double) * (size_t)nkx, st));
        A.ky = d_k; A.kx = d_k + nky;
    } else {
        SPC_HIP(spc_table_upload(d_k, h_ky, sizeof(double) * ntab, st));
        A.ky = d_k; A.kx = nullptr;
    }
    A.nky = nky; A.nkx = nkx; A.out = d_out;
    A.out_row_stride = out_row_stride ? out_row_stride : cube->nx;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : cube->ny * A.out_row_stride;
    const unsigned gx = (unsigned)((cube->nx + 255) / 256);
    if (!separable) {
        for (int64_t z0 = 0; z0 < cube->nz; z0 += 65535) {
            A.z0 = z0;
            const unsigned gz = (unsigned)std::min<int64_t>(65535, cube->nz - z0);
            hipLaunchKernelGGL(spatial64_direct_kernel, dim3(gx, (unsigned)cube->ny, gz), dim3(256), 0, st, A);
            SPC_LAUNCH_CHECK();
        }
        return SPC_OK;
    }
    // symmetric factors of up to 33 taps, none negative, centre taps positive: the one-kernel ring form (SPC_SPATIAL64_RING=0: the
    // two-pass forms)
    const bool ring_on = [] { const char* e = getenv("SPC_SPATIAL64_RING"); return e ? atoi(e) != 0 : true; }();     // (read per call: the tests compare both forms in one process)
    bool ring = ring_on && nky <= 33 && nkx <= 33 && h_ky[nky / 2] > 0.0 && h_kx[nkx / 2] > 0.0;
    for (int i = 0; ring && i < nky / 2; ++i) ring = h_ky[i] == h_ky[nky - 1 - i] && h_ky[i] >= 0.0;
    for (int i = 0; ring && i < nkx / 2; ++i) ring = h_kx[i] == h_kx[nkx - 1 - i] && h_kx[i] >= 0.0;
    A.gate = nullptr;
    if (ring) {
        unsigned* d_flag = nullptr;
        if (!(A.m.flags & SPC_MASK_FINITE)) {                   // the mask admits infinite samples: see spatial64_ring_kernel
            SPC_WS_TAKE(d_f, ws, unsigned, 64);
            d_flag = d_f;
            SPC_HIP(hipMemsetAsync(d_flag, 0, sizeof(unsigned), st));
        }
        if (std::max(nky, nkx) <= 17) rc = launch_ring64<17>(st, A, h_ky, nky, h_kx, nkx, d_flag);
        else rc = launch_ring64<33>(st, A, h_ky, nky, h_kx, nkx, d_flag);
        if (rc) return rc;
        if (!d_flag) return SPC_OK;
        A.gate = d_flag;                                        // the passes below run only if the ring kernel raised the flag
    }
    const size_t plane = (size_t)cube->ny * (size_t)cube->nx * 2 * sizeof(double);
    const size_t avail = ws.size > ws.used + 512 ? ws.size - ws.used - 512 : 0;
    const int64_t planes = (int64_t)std::min<size_t>(std::min<size_t>(avail / plane, (size_t)cube->nz), 65535);
    SPC_REQUIRE(planes >= 1, "d_workspace too small: the (num, den) planes of one channel need %zu bytes more (spc_workspace_bytes)", plane);
    SPC_WS_TAKE(d_tmp, ws, double, (size_t)planes * (size_t)cube->ny * (size_t)cube->nx * 2);
    A.tmp = d_tmp;
    for (int64_t z0 = 0; z0 < cube->nz; z0 += planes) {
        A.z0 = z0;
        const unsigned gz = (unsigned)std::min<int64_t>(planes, cube->nz - z0);
        static const int tiled = [] { const char* e = getenv("SPC_SPATIAL64_LDS"); return e ? atoi(e) : 2; }();
        // (SPC_SPATIAL64_LDS: 0 = the untiled passes, 1 = LDS tiles with one output per thread (round 5), 2 = four outputs per thread)
        if (tiled == 2 && nkx / 2 <= kSpX4Halo)
            hipLaunchKernelGGL(spatial64_xpass_lds4_kernel, dim3((unsigned)((cube->nx + kSpX4Out - 1) / kSpX4Out), (unsigned)cube->ny, gz), dim3(256), 0, st, A);
        else if (tiled && nkx / 2 <= kSpHaloMax)
            hipLaunchKernelGGL(spatial64_xpass_lds_kernel, dim3(gx, (unsigned)cube->ny, gz), dim3(256), 0, st, A);
        else
            hipLaunchKernelGGL(spatial64_xpass_kernel, dim3(gx, (unsigned)cube->ny, gz), dim3(256), 0, st, A);
        SPC_LAUNCH_CHECK();
        if (tiled == 2 && nky / 2 <= kSpYHalo && (cube->ny + kSpYRows - 1) / kSpYRows <= 65535)
            hipLaunchKernelGGL(spatial64_ypass_lds4_kernel, dim3((unsigned)((cube->nx + 63) / 64), (unsigned)((cube->ny + kSpYRows - 1) / kSpYRows), gz),
                               dim3(256), (size_t)(kSpYRows + 2 * (nky / 2)) * 64 * 16, st, A);
        else if (tiled && nky / 2 <= kSpYHalo && (cube->ny + kSpYRows - 1) / kSpYRows <= 65535)
            hipLaunchKernelGGL(spatial64_ypass_lds_kernel, dim3((unsigned)((cube->nx + 63) / 64), (unsigned)((cube->ny + kSpYRows - 1) / kSpYRows), gz),
                               dim3(256), (size_t)(kSpYRows + 2 * (nky / 2)) * 64 * 16, st, A);
        else
            hipLaunchKernelGGL(spatial64_ypass_kernel, dim3(gx, (unsigned)cube->ny, gz), dim3(256), 0, st, A);
        SPC_LAUNCH_CHECK();
    }
    return SPC_OK;
}

int spc_spectral_lerp_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, int64_t nz_out,
                          const int32_t* d_lo, const double* d_t, const double* d_inv_dx, double fill, double* d_out,
                          int64_t out_row_stride, int64_t out_plane_stride) {
    Lerp64Args A{};
    int rc = cube64_args(cube, mask, &A.c, &A.m);
    if (rc) return rc;
    SPC_REQUIRE(nz_out > 0 && d_lo && d_t && d_inv_dx && d_out, "NULL pointer argument / nz_out must be positive");
    SPC_DEVICE(device);
    A.nz_out = nz_out; A.lo = d_lo; A.t = d_t; A.inv_dx = d_inv_dx; A.fill = fill; A.out = d_out;
    A.out_row_stride = out_row_stride ? out_row_stride : cube->nx;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : cube->ny * A.out_row_stride;
    const int64_t nb = (cube->ny * cube->nx + 255) / 256;
    SPC_REQUIRE(nb < (1LL << 31), "map too large for one launch");
    int nsplit = 1;
    if (nb < 2048) nsplit = (int)std::max<int64_t>(1, std::min<int64_t>((2048 + nb - 1) / nb, nz_out / 8));
    A.jchunk = (nz_out + nsplit - 1) / nsplit;
    nsplit = (int)((nz_out + A.jchunk - 1) / A.jchunk);
    hipLaunchKernelGGL(spectral_lerp64_kernel, dim3((unsigned)nb, (unsigned)nsplit), dim3(256), 0, (hipStream_t)stream, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_resample_bilinear_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, double fill,
                              int64_t ny_out, int64_t nx_out, const double* d_xs, const double* d_ys, double* d_out,
                              int64_t out_row_stride, int64_t out_plane_stride, uint8_t* d_footprint, int order,
                              uint32_t* d_any_valid) {
    Bil64Args A{};
    int rc = cube64_args(cube, mask, &A.c, &A.m);
    if (rc) return rc;
    SPC_REQUIRE(ny_out > 0 && nx_out > 0, "output shape must be positive");
    SPC_REQUIRE(d_xs && d_ys && d_out, "NULL pointer argument");
    SPC_REQUIRE(order == 0 || order == 1, "order must be 1 (bilinear) or 0 (nearest neighbour), got %d", order);
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    A.fill = fill; A.ny_out = ny_out; A.nx_out = nx_out; A.xs = d_xs; A.ys = d_ys; A.out = d_out;
    A.out_row_stride = out_row_stride ? out_row_stride : nx_out;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : ny_out * A.out_row_stride;
    A.footprint = d_footprint; A.nearest = order == 0; A.any_valid = d_any_valid;
    if (d_any_valid) SPC_HIP(hipMemsetAsync(d_any_valid, 0, sizeof(uint32_t), st));
    const int64_t nblocks = ((nx_out + 63) / 64) * ((ny_out + 3) / 4);
    SPC_REQUIRE(nblocks < (1LL << 31), "output map too large for one launch");
    int nsplit = 1;
    if (nblocks < 2048) nsplit = (int)std::max<int64_t>(1, std::min<int64_t>((2048 + nblocks - 1) / nblocks, cube->nz / 8));
    A.zchunk = (cube->nz + nsplit - 1) / nsplit;
    nsplit = (int)((cube->nz + A.zchunk - 1) / A.zchunk);
    hipLaunchKernelGGL(bilinear64_kernel, dim3((unsigned)nblocks, (unsigned)nsplit), dim3(256), 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_scale_f64(int device, void* stream, double* d_data, int64_t n, double factor) {
    SPC_REQUIRE(d_data != nullptr && n >= 0, "NULL pointer / negative count");
    SPC_DEVICE(device);
    const int64_t nb = (n + 255) / 256;
    SPC_REQUIRE(nb < (1LL << 31), "array too large for one launch");
    if (nb) hipLaunchKernelGGL(scale64_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, d_data, n, factor);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_percentile_axis0_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, double q,
                             const double* d_center, double scale, double* d_out) {
    Sort64Args A{};
    int rc = cube64_args(cube, mask, &A.c, &A.m, true);       // (rays along y: the view with the first two axes exchanged)
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    SPC_REQUIRE(q >= 0.0 && q <= 100.0, "percentile must lie in [0, 100], got %g", q);
    SPC_DEVICE(device);
    A.q = q; A.scale = scale; A.center = d_center; A.out = d_out;
    return sort64_launch(A, cube, false, (hipStream_t)stream);
}

int spc_sigma_clip_axis0_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, double sigma_lower,
                             double sigma_upper, int maxiters, int center_is_mean, int spread_is_mad, double* d_out) {
    Sort64Args A{};
    int rc = cube64_args(cube, mask, &A.c, &A.m);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    SPC_DEVICE(device);
    A.lo_s = sigma_lower; A.hi_s = sigma_upper; A.maxiters = maxiters; A.cen_mean = center_is_mean; A.spread_mad = spread_is_mad; A.out = d_out;
    A.q = 50.0; A.scale = 1.0;
    return sort64_launch(A, cube, true, (hipStream_t)stream);
}

int spc_narrow_f64_to_f32(int device, void* stream, const spc_cube_f64* cube, float* d_out, int64_t out_row_stride,
                          int64_t out_plane_stride) {
    Cube64 C; MaskDev64 M;
    int rc = cube64_args(cube, nullptr, &C, &M);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    SPC_REQUIRE(cube->ny <= 65535 && cube->nz <= 65535, "too many rows / channels for one launch");
    SPC_DEVICE(device);
    const int64_t rs = out_row_stride ? out_row_stride : cube->nx, ps = out_plane_stride ? out_plane_stride : cube->ny * rs;
    hipLaunchKernelGGL(narrow64_kernel, dim3((unsigned)((cube->nx + 255) / 256), (unsigned)cube->ny, (unsigned)cube->nz), dim3(256), 0,
                       (hipStream_t)stream, C, d_out, rs, ps);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_mask_include_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, int nan_excluded,
                         uint8_t* d_out) {
    Cube64 C; MaskDev64 M;
    int rc = cube64_args(cube, mask, &C, &M);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    SPC_REQUIRE(cube->ny <= 65535 && cube->nz <= 65535, "too many rows / channels for one launch");
    SPC_DEVICE(device);
    hipLaunchKernelGGL(include64_kernel, dim3((unsigned)((cube->nx + 255) / 256), (unsigned)cube->ny, (unsigned)cube->nz), dim3(256), 0,
                       (hipStream_t)stream, C, M, nan_excluded, d_out);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

}  // extern "C"
