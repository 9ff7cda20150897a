// Kernel templates of the ring-streaming spectral stencil; included by the
// per-ring-size translation units spc_spectral_conv_r*.hip (one TU per R so
// `make -j` builds the fully unrolled kernels in parallel).
//
// Ring streaming.  One lane owns VEC (1 or 2) adjacent spaxels and marches over
// z.  The R = 2H+1 outputs that the newest input still contributes to live in R
// (num, den) accumulator registers.  Every input is loaded ONCE; at step s it
// is FMA-ed into all R slots (slot m holds the output of age a = (s-m) mod R
// and takes weight k[2H-a]), then the oldest output (slot (s+1) mod R)
// completes and is emitted.  The loop is unrolled by R ("one revolution") so
// slots and weights are static registers.
//
// Arithmetic = astropy's: astropy.convolution.convolve promotes the array and the
// kernel to float64, accumulates top += val * ker / bot += ker in float64, divides
// and rounds ONCE to the input dtype (float32).  The rings therefore hold float64
// accumulators fed by v_fma_f64 with the float64 taps in SGPR pairs, the division
// is a correctly rounded float64 division (a reciprocal multiply + one Newton
// residual step when the denominator is the full kernel sum), and the result is
// rounded to float32 once.  What differs from astropy is only the ORDER of the
// float64 additions (1e-16 relative), so the float32 outputs are bit-identical
// except where the float64 value sits within ~1e-16 of a float32 rounding
// boundary - which is what makes the argmax of a smoothed cube an exact integer
// map and keeps ill-conditioned moments of the smoothed cube inside 1e-5.
//
// What bounds it: VALU issue, not HBM.  Measured on MI355X
// (tests/micro/valu_rate.hip): a wave64 VALU instruction occupies its SIMD for
// ~4 cycles whether it is v_fmac_f32, v_pk_fma_f32 or v_fma_f64.  Hence:
//   * the FMAs are in-place `v_fma_f64` inline asm with the weight in an SGPR pair:
//     left alone, LLVM sinks each slot's FMA chain to its emission point (R-long
//     dependent chains, twice the registers);
//   * the NaN-renormalising denominator den = sum of the weights of the VALID samples of
//     a window is a function of the window's R validity bits only.  It is not accumulated:
//     every lane keeps the validity of its last 64 samples in one 64-bit register and the
//     block keeps, in LDS, the partial sums of the taps for every pattern of 11 adjacent
//     bits (ceil(R / 11) tables of 2048 float64 sums, built by the block when it starts).
//     A denominator is then <= 3 LDS reads + 2 additions instead of R FMAs and R live
//     registers; a revolution whose windows are all valid (wave-uniform) divides by the
//     kernel sum without looking anything up;
//   * ONE unrolled body with wave-uniform run-time flags: two specialised bodies
//     (or a per-step validity branch) make the register allocator duplicate /
//     copy the whole ring (measured: 2x VGPRs or 2R v_mov per step);
//   * loads/stores go through buffer descriptors built from readfirstlane'd
//     plane bases (no per-load 64-bit VGPR address, no waterfall loops).
#pragma once
#include "spc_common.h"
#include <algorithm>

namespace spc_sconv {

constexpr int kMaxTaps = 33;                   // three denominator tables of 11 validity bits
typedef float float2v __attribute__((ext_vector_type(2)));

struct ConvArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    float* out;
    int64_t out_row_stride, out_plane_stride;
    int64_t zchunk;            // outputs per gridDim.y slice
    // fused-moment part
    const double* cen;
    double cen_c0, cen_dc;     // linear-axis form c[z] = c0 + z*dc when cen_linear
    int cen_linear;
    double dv, m1_add;
    spc_moment_outputs mo;
    int64_t mo_row_stride;
    // all-valid fast pass: one byte per 128-column tile, 0 = done by the fast kernel
    unsigned char* status;
    double ksum, inv_ksum;     // sum(k) in tap order, 1 / sum(k)
    alignas(16) double k[72];  // taps padded to R, centred (R <= 65)
};

// ---- denominators from validity bits ------------------------------------------------------
// Bit b of a lane's validity history = the sample that arrived b steps ago; when an output
// completes, that sample carried tap k[b] (the newest sample takes k[0], see ring_body).
constexpr int kLutBits = 11, kLutSize = 1 << kLutBits;
constexpr int lut_tables(int R) { return (R + kLutBits - 1) / kLutBits; }
// threads per block of the general kernel: the tables are per block (48 KB for 33 taps), so the block size sets
// how many waves share a CU: 2 x 512 threads (4 waves per SIMD) where the registers allow it (no mask array, not
// fused: 120 VGPRs), 3 x 256 threads otherwise (139 VGPRs with a mask array; measured 3.7 ms against 4.1 ms)
constexpr int general_block(bool arr, bool fuse) { return (arr || fuse) ? 256 : 512; }

template <int R>
__device__ __forceinline__ void lut_build(const ConvArgs& A, double* lut) {
#pragma unroll
    for (int t = 0; t < lut_tables(R); ++t) {
        for (int w = threadIdx.x; w < kLutSize; w += blockDim.x) {
            double acc = 0.0;
#pragma unroll
            for (int b = 0; b < kLutBits; ++b)
                if (kLutBits * t + b < R) acc += ((w >> b) & 1) ? A.k[kLutBits * t + b] : 0.0;
            lut[t * kLutSize + w] = acc;
        }
    }
}

// byte-offset mask of a table of which `bits` (<= 11) index bits are in use
constexpr unsigned lut_mask(int bits) { return (((bits >= kLutBits) ? (unsigned)kLutSize : (bits > 0 ? (1u << (bits > 0 ? bits : 0)) : 1u)) - 1u) * 8u; }

template <int R>
__device__ __forceinline__ double lut_den(const double* lut, unsigned long long hist) {
    const unsigned lo = (unsigned)hist, hi = (unsigned)(hist >> 32);
    const char* base = reinterpret_cast<const char*>(lut);
    double den = *reinterpret_cast<const double*>(base + ((lo << 3) & lut_mask(R)));
    if (R > kLutBits)
        den += *reinterpret_cast<const double*>(base + kLutSize * 8 + ((lo >> (kLutBits - 3)) & lut_mask(R - kLutBits)));
    if (R > 2 * kLutBits)
        den += *reinterpret_cast<const double*>(base + 2 * kLutSize * 8 + (__builtin_amdgcn_alignbit(hi, lo, 2 * kLutBits - 3) & lut_mask(R - 2 * kLutBits)));
    return den;
}

// fused-moment running state of one spaxel
struct MomState {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    int nvalid = 0;
    float bmax = -INFINITY, bmin = INFINITY;
    int imax = 0, imin = 0;
};

// ---- in-place FMA / MUL with a scalar (SGPR pair) float64 weight ------------------------
__device__ __forceinline__ void fma_w(double& acc, const ConvArgs& A, int j, double x) {
    asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc) : "s"(A.k[j]), "v"(x));
}
__device__ __forceinline__ void mul_w(double& acc, const ConvArgs& A, int j, double x) {
    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(acc) : "s"(A.k[j]), "v"(x));
}

// num / sum(k), correctly rounded: q = num * (1/ksum) is within an ulp; one residual step
// r = num - q * ksum (exact in the FMA), q += r / ksum brings it to the rounded quotient
// (the step every software division ends with) - 3 VALU slots instead of ~14.
__device__ __forceinline__ double div_ksum(double num, const ConvArgs& A) {
    const double q = num * A.inv_ksum;
    // (an infinite quotient has a NaN residual: max() drops the NaN, the infinity survives the FMA)
    const double r = fmax(fma(-q, A.ksum, num), -1.7976931348623157e308);
    return fma(r, A.inv_ksum, q);
}

// Buffer descriptor over one plane.  The base must be wave-uniform AND provably
// so, otherwise hipcc wraps every buffer op in a waterfall loop (guide T20):
// pass both halves through readfirstlane.
__device__ __forceinline__ auto plane_srd(const void* base) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)0xffffffffu,
                                             0x00020000);
}

// loads / stores relative to a per-revolution descriptor: soffset (SGPR) = plane delta in bytes
template <int VEC> struct Io;
template <> struct Io<1> {
    using T = float;
    static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, /*nt*/ 2));
    }
    static __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t rs, int voff, int soff, const float (&v)[1]) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[0]), rs, voff, soff, /*nt*/ 2);
    }
    static __device__ __forceinline__ float get(T v, int) { return v; }
};
template <> struct Io<2> {
    using T = float2v;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ T ld(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
        return __builtin_bit_cast(float2v, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, /*nt*/ 2));
    }
    static __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t rs, int voff, int soff, const float (&v)[2]) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, float2v{v[0], v[1]}), rs, voff, soff, /*nt*/ 2);
    }
    static __device__ __forceinline__ float get(T v, int c) { return v[c]; }
};

__device__ __forceinline__ float ld1(const float* plane, int voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(plane_srd(plane), voff, 0, /*nt*/ 2));
}
__device__ __forceinline__ unsigned ldm1(const uint8_t* plane, int moff) {
    return __builtin_amdgcn_raw_buffer_load_b8(plane_srd(plane), moff, 0, 2);
}

// The R x R update + emission part of one revolution.
//   v[s]  : classified input: the value when valid, NaN when invalid, 0 when out of range
//   incb  : (FUSE only) bit s = sample s is in range and included by the mask
//   okhist: validity of the lane's last 64 samples, bit 0 = newest (out-of-range samples are
//           valid zeros, boundary='fill')
// FULL (wave-uniform, run-time): this and the previous revolution were all valid, so every
// output completing now has the whole kernel as its denominator.
template <int R, bool ARR, bool FUSE, bool EXT, bool SYM>
__device__ __forceinline__ void ring_body(const ConvArgs& A, double (&num)[R], const double* lut,
                                          const float (&v)[R], unsigned long long incb, unsigned long long& okhist,
                                          unsigned long long& inc_hist, MomState& ms, int voff_out, int i0,
                                          int zb, int ze, const bool FULL) {
    constexpr int H = R / 2;
    // outputs of this revolution: planes i0 - H .. i0 + H, addressed from the first one that exists
    const int ob = max(i0 - H, 0);
    const int obytes = FUSE ? 0 : (int)(A.out_plane_stride * 4);
    const auto ro = plane_srd(FUSE ? (const void*)A.cube : (const void*)(A.out + (int64_t)ob * A.out_plane_stride));
#pragma unroll
    for (int s = 0; s < R; ++s) {
        const bool ok = v[s] == v[s];
        const double x = ok ? (double)v[s] : 0.0;
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int a = (s - m + R) % R;          // age of the output living in slot m
            // symmetric kernels: half the distinct weights, all of them stay in SGPRs
            const int j = SYM ? (a <= H ? a : 2 * H - a) : 2 * H - a;
            if (a == 0) mul_w(num[m], A, j, x);
            else fma_w(num[m], A, j, x);
        }
        if (!FULL) okhist = (okhist << 1) | (ok ? 1ull : 0ull);
        if (FUSE && A.mask.flags) inc_hist = (inc_hist << 1) | ((incb >> s) & 1ull);
        // ---- the output that just received its last contribution
        const int e = (s + 1) % R;
        const int o = i0 + s - H;
        if (o >= zb && o < ze) {
            float res;
            if (FULL) {
                res = (float)div_ksum(num[e], A);
            } else {
                const double dtot = lut_den<R>(lut, okhist);
                // astropy returns the (filled) centre sample for an empty window; the host
                // only dispatches kernels with a non-zero centre tap here, for which an empty
                // window implies an invalid centre, i.e. NaN
                res = (dtot != 0.0) ? (float)(num[e] / dtot) : NAN;
            }
            if (!FUSE) {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res), ro, voff_out, (int)((unsigned)(o - ob) * (unsigned)obytes), 0);
            } else {
                // no mask at all: every in-range channel is included
                const bool inc_o = A.mask.flags ? (((inc_hist >> H) & 1ull) != 0ull) : true;
                const bool okm = inc_o && (res == res);
                const double wd = okm ? (double)res : 0.0;
                const double c = A.cen_linear ? fma((double)o, A.cen_dc, A.cen_c0) : A.cen[o];
                ms.s0 += wd;
                ms.s1 = fma(wd, c, ms.s1);
                ms.s2 = fma(wd, c * c, ms.s2);
                ms.nvalid += okm ? 1 : 0;
                if (EXT) {
                    const float hi = okm ? res : -INFINITY;
                    const float lo = okm ? res : INFINITY;
                    if (hi > ms.bmax) { ms.bmax = hi; ms.imax = (int)o; }
                    if (lo < ms.bmin) { ms.bmin = lo; ms.imin = (int)o; }
                }
            }
        }
    }
    if (FULL) okhist = ~0ull;                        // R valid samples went by
}

template <int R, bool ARR, bool FUSE, bool EXT, bool SYM>
// (fused: 3 waves per SIMD asked for - 168 VGPRs, a handful of spilled dwords - instead of the 180 the allocator
// would take: 40.7 -> 31.7 ms for the masked C3 smooth -> moments)
__global__ __launch_bounds__(general_block(ARR, FUSE), FUSE ? 3 : 1) void spectral_conv_kernel(const ConvArgs A) {
    constexpr int H = R / 2;
    static_assert(R <= 3 * kLutBits, "three denominator tables cover 33 taps");
    __shared__ double lut[lut_tables(R) * kLutSize];
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // tiles (128 columns) already finished by the all-valid fast kernel
    const bool live = (col < A.ny * A.nx) &&
                      !(A.status && spc_flag_get(A.status + __builtin_amdgcn_readfirstlane((int)(min(col, A.ny * A.nx - 1) >> 7))) == 0);
    if (!__syncthreads_or(live ? 1 : 0)) return;
    lut_build<R>(A, lut);
    __syncthreads();
    if (!live) return;
    const int64_t y = col / A.nx, x = col - y * A.nx;
    const int nz = (int)A.nz;
    const int zb = (int)(blockIdx.y * A.zchunk);
    const int ze = min(nz, zb + (int)A.zchunk);
    const int voff_out = FUSE ? 0 : (int)((y * A.out_row_stride + x) * 4);
    const uint32_t flags = A.mask.flags;
    const float tlo = A.mask.thr_lo, thi = A.mask.thr_hi;
    const int voff = (int)((y * A.row_stride + x) * 4);              // < 2 GiB per plane (checked on the host)
    const int moff = ARR ? (int)(y * A.mask.row_stride + x) : 0;
    double num[R];
#pragma unroll
    for (int m = 0; m < R; ++m) num[m] = 0.0;
    unsigned long long inc_hist = 0ull;  // include bit of the last 64 inputs (bit 0 = newest)
    unsigned long long okhist = ~0ull;   // validity of the last 64 inputs (what lies before the slice start is never emitted)
    MomState ms;
    bool prev_allv = false;
    const int pbytes = (int)(A.plane_stride * 4), mbytes = ARR ? (int)A.mask.plane_stride : 0;

    const int T = (ze - zb) + 2 * H;      // number of input steps
    for (int t0 = 0; t0 < T; t0 += R) {
        const int i0 = __builtin_amdgcn_readfirstlane(zb - H + t0);   // first input channel of this revolution
        float v[R];
        unsigned mk[R];
        // ONE descriptor per revolution (base = first plane read) + a scalar byte offset per load
        const int pb = min(max(i0, 0), nz - 1);
        const auto rs = plane_srd(A.cube + (int64_t)pb * A.plane_stride);
        const auto rm = plane_srd(ARR ? (const void*)(A.mask.arr + (int64_t)pb * A.mask.plane_stride) : (const void*)A.cube);
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int dz = min(max(i0 + s, 0), nz - 1) - pb;                 // clamped, uniform
            v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)((unsigned)dz * (unsigned)pbytes), /*nt*/ 2));
            if (ARR) mk[s] = __builtin_amdgcn_raw_buffer_load_b8(rm, moff, (int)((unsigned)dz * (unsigned)mbytes), 2);
        }
        // ---- classify: value | NaN (invalid) | 0 (out of range = valid zero, boundary='fill')
        unsigned long long incb = 0ull;
        bool bad = false;
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const bool in = (i0 + s >= 0) && (i0 + s < nz);                  // uniform
            bool inc = spc_pred(flags, tlo, thi, v[s]);
            if (ARR) inc = inc && (mk[s] != 0);
            const bool ok = inc && (v[s] == v[s]);
            bad = bad || (in && !ok);
            v[s] = in ? (ok ? v[s] : NAN) : 0.f;
            if (FUSE && flags) incb |= ((in && inc) ? 1ull : 0ull) << s;
        }
        const bool allv = !__any(bad);
        // (the first revolution of a z slice starts from zeroed rings whose first H outputs lie
        // before zb and are never emitted, so "previous revolution all valid" holds vacuously)
        const bool full = allv && (prev_allv || t0 == 0);
        ring_body<R, ARR, FUSE, EXT, SYM>(A, num, lut, v, incb, okhist, inc_hist, ms, voff_out, i0, zb, ze, full);
        prev_allv = allv;
    }

    if (FUSE) {
        const int64_t o = y * A.mo_row_stride + x;
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        const double mu = ms.s1 / ms.s0;
        if (A.mo.d_m0) A.mo.d_m0[o] = ms.nvalid > 0 ? A.dv * ms.s0 : nan;
        if (A.mo.d_m1) A.mo.d_m1[o] = mu + A.m1_add;
        if (A.mo.d_m2) A.mo.d_m2[o] = ms.s2 / ms.s0 - mu * mu;
        if (A.mo.d_mu) A.mo.d_mu[o] = mu;
        if (A.mo.d_s0) A.mo.d_s0[o] = ms.s0;
        if (A.mo.d_argmax) A.mo.d_argmax[o] = ms.nvalid > 0 ? (int64_t)ms.imax : 0;
        if (A.mo.d_argmin) A.mo.d_argmin[o] = ms.nvalid > 0 ? (int64_t)ms.imin : 0;
        if (A.mo.d_vmax) A.mo.d_vmax[o] = ms.nvalid > 0 ? ms.bmax : NAN;
        if (A.mo.d_vmin) A.mo.d_vmin[o] = ms.nvalid > 0 ? ms.bmin : NAN;
        if (A.mo.d_nvalid) A.mo.d_nvalid[o] = ms.nvalid;
    }
}

// ---- all-valid fast kernel ------------------------------------------------------------
// Speculative first pass for data WITHOUT invalid samples (the common case: cubes whose
// only NaNs are blanked edges): numerators only, out = num / sum(k) exactly like
// astropy's NaN-free branch (float64 accumulation, correctly rounded division, one
// rounding to float32).  A wavefront that meets an invalid sample marks its tile dirty
// and quits; the general kernel then redoes only the dirty tiles.  VEC = spaxels per lane.
template <int R, bool FUSE, int VEC, bool SYM>
__global__ __launch_bounds__(256) void spectral_conv_fast_kernel(const ConvArgs A) {
    constexpr int H = R / 2;
    using IO = Io<VEC>;
    const int64_t gpr = A.nx / VEC;
    const int64_t ngroups = A.ny * gpr;
    const int64_t g0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // status bytes are per 128 columns of the linear spaxel index
    const bool live = g0 < ngroups;
    const int64_t g = live ? g0 : ngroups - 1;
    const int64_t y = g / gpr, x = (g - y * gpr) * VEC;
    const int nz = (int)A.nz;
    const int voff = (int)((y * A.row_stride + x) * 4);
    const int voff_out = FUSE ? 0 : (int)((y * A.out_row_stride + x) * 4);
    const bool EXT = FUSE && (A.mo.d_argmax || A.mo.d_argmin || A.mo.d_vmax || A.mo.d_vmin);

    double num[VEC][R];
#pragma unroll
    for (int c = 0; c < VEC; ++c)
#pragma unroll
        for (int m = 0; m < R; ++m) num[c][m] = 0.0;
    MomState ms[VEC];

    // One revolution's inputs are loaded up front.
    // Addressing: ONE buffer descriptor per revolution + a scalar byte offset per access.
    const int pbytes = (int)(A.plane_stride * 4), obytes = FUSE ? 0 : (int)(A.out_plane_stride * 4);
    const int T = nz + 2 * H;
    for (int t0 = 0; t0 < T; t0 += R) {
        const int i0 = __builtin_amdgcn_readfirstlane(t0 - H);
        const int pb = min(max(i0, 0), nz - 1);                            // base plane of the loads
        const auto rs = plane_srd(A.cube + (int64_t)pb * A.plane_stride);
        const int ob = max(i0 - H, 0);                                     // base plane of the outputs
        const auto ro = plane_srd(FUSE ? (const float*)A.cube : A.out + (int64_t)ob * A.out_plane_stride);
        typename IO::T v[R];
#pragma unroll
        for (int s = 0; s < R; ++s) v[s] = IO::ld(rs, voff, (int)((unsigned)(min(max(i0 + s, 0), nz - 1) - pb) * (unsigned)pbytes));
        // The fast pass only runs for masks that reject exactly the non-finite samples (none /
        // isfinite), and those propagate through the FMAs: chk += 0 * (finished output) is NaN
        // iff a NaN or Inf went into it - one FMA per output instead of compares on every
        // sample.  Planes outside the cube (clamped duplicate loads) become valid zeros under
        // a wave-uniform branch that only the first / last revolution takes.
        if ((i0 < 0) || (i0 + R > nz)) {
#pragma unroll
            for (int s = 0; s < R; ++s)
                if (!((i0 + s >= 0) && (i0 + s < nz))) v[s] = typename IO::T{};
        }
        float chk = 0.f;
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int e = (s + 1) % R;
            const int o = i0 + s - H;
            float res[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const double x64 = (double)IO::get(v[s], c);
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    const int a = (s - m + R) % R;
                    const int j = SYM ? (a <= H ? a : 2 * H - a) : 2 * H - a;
                    if (a == 0) mul_w(num[c][m], A, j, x64);
                    else fma_w(num[c][m], A, j, x64);
                }
                res[c] = (float)div_ksum(num[c][e], A);
                chk = fmaf(res[c], 0.f, chk);
            }
            if (o >= 0 && o < nz) {
                if (!FUSE) {
                    if (live) IO::st(ro, voff_out, (int)((unsigned)(o - ob) * (unsigned)obytes), res);
                } else {
                    const double cz = A.cen_linear ? fma((double)o, A.cen_dc, A.cen_c0) : A.cen[o];
                    const double czz = cz * cz;
#pragma unroll
                    for (int c = 0; c < VEC; ++c) {
                        const double wd = (double)res[c];
                        MomState& w = ms[c];
                        w.s0 += wd;
                        w.s1 = fma(wd, cz, w.s1);
                        w.s2 = fma(wd, czz, w.s2);
                        if (EXT) {
                            if (res[c] > w.bmax) { w.bmax = res[c]; w.imax = o; }
                            if (res[c] < w.bmin) { w.bmin = res[c]; w.imin = o; }
                        }
                    }
                }
            }
        }
        // wave-uniform: a non-finite sample went into this revolution's outputs -> the general
        // kernel redoes the tile (whatever this wave already stored is overwritten).  A wave of
        // VEC = 1 covers 64 columns = half a 128-column status tile; flagging the whole tile is
        // harmless (the general kernel recomputes it from the inputs).
        if (__any(!(chk == chk) && live)) {
            if ((threadIdx.x & 63) == 0) spc_flag_set(A.status + ((y * A.nx + x) >> 7));
            return;
        }
    }
    if (FUSE && live) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const int64_t o = y * A.mo_row_stride + x + c;
            MomState w = ms[c];
            w.nvalid = nz;                                 // every sample of a clean tile is valid
            const double mu = w.s1 / w.s0;
            if (A.mo.d_m0) A.mo.d_m0[o] = A.dv * w.s0;
            if (A.mo.d_m1) A.mo.d_m1[o] = mu + A.m1_add;
            if (A.mo.d_m2) A.mo.d_m2[o] = w.s2 / w.s0 - mu * mu;
            if (A.mo.d_mu) A.mo.d_mu[o] = mu;
            if (A.mo.d_s0) A.mo.d_s0[o] = w.s0;
            if (A.mo.d_argmax) A.mo.d_argmax[o] = (int64_t)w.imax;
            if (A.mo.d_argmin) A.mo.d_argmin[o] = (int64_t)w.imin;
            if (A.mo.d_vmax) A.mo.d_vmax[o] = w.bmax;
            if (A.mo.d_vmin) A.mo.d_vmin[o] = w.bmin;
            if (A.mo.d_nvalid) A.mo.d_nvalid[o] = w.nvalid;
        }
    }
}

template <int R, bool FUSE, bool SYM>
int launch_rs(const ConvArgs& A, hipStream_t st, int64_t ncols, unsigned nsplit, bool arr, bool ext) {
    if (arr) {
        constexpr int B = general_block(true, FUSE);
        dim3 grid((unsigned)((ncols + B - 1) / B), nsplit), block(B);
        if (FUSE && ext) hipLaunchKernelGGL((spectral_conv_kernel<R, true, FUSE, FUSE, SYM>), grid, block, 0, st, A);
        else hipLaunchKernelGGL((spectral_conv_kernel<R, true, FUSE, false, SYM>), grid, block, 0, st, A);
    } else {
        constexpr int B = general_block(false, FUSE);
        dim3 grid((unsigned)((ncols + B - 1) / B), nsplit), block(B);
        if (FUSE && ext) hipLaunchKernelGGL((spectral_conv_kernel<R, false, FUSE, FUSE, SYM>), grid, block, 0, st, A);
        else hipLaunchKernelGGL((spectral_conv_kernel<R, false, FUSE, false, SYM>), grid, block, 0, st, A);
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

// entry point instantiated once per ring size in spc_spectral_conv_r<R>.hip
// fast: run the all-valid kernel first (A.status must point at ceil(ny*nx/128) zeroed
// bytes; fast = spaxels per lane, 1 or 2), then the general kernel on the dirty tiles.
template <int R>
int launch(const ConvArgs& A, hipStream_t st, int fast, bool fuse) {
    bool sym = true;
    for (int i = 0; i < R / 2; ++i) sym = sym && (A.k[i] == A.k[R - 1 - i]);
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const bool ext = fuse && (A.mo.d_argmax || A.mo.d_argmin || A.mo.d_vmax || A.mo.d_vmin);
    if (fast) {
        const int64_t ngroups = A.ny * (A.nx / fast);
        dim3 fgrid((unsigned)((ngroups + 255) / 256), 1), block(256);
        if (fast == 2) {
            if (fuse) { if (sym) hipLaunchKernelGGL((spectral_conv_fast_kernel<R, true, 2, true>), fgrid, block, 0, st, A);
                        else hipLaunchKernelGGL((spectral_conv_fast_kernel<R, true, 2, false>), fgrid, block, 0, st, A); }
            else { if (sym) hipLaunchKernelGGL((spectral_conv_fast_kernel<R, false, 2, true>), fgrid, block, 0, st, A);
                   else hipLaunchKernelGGL((spectral_conv_fast_kernel<R, false, 2, false>), fgrid, block, 0, st, A); }
        } else {
            if (fuse) { if (sym) hipLaunchKernelGGL((spectral_conv_fast_kernel<R, true, 1, true>), fgrid, block, 0, st, A);
                        else hipLaunchKernelGGL((spectral_conv_fast_kernel<R, true, 1, false>), fgrid, block, 0, st, A); }
            else { if (sym) hipLaunchKernelGGL((spectral_conv_fast_kernel<R, false, 1, true>), fgrid, block, 0, st, A);
                   else hipLaunchKernelGGL((spectral_conv_fast_kernel<R, false, 1, false>), fgrid, block, 0, st, A); }
        }
        SPC_LAUNCH_CHECK();
    }
    const int64_t ncols = A.ny * A.nx;
    const int64_t nsplit = (A.nz + A.zchunk - 1) / A.zchunk;
    if (fuse) return sym ? launch_rs<R, true, true>(A, st, ncols, (unsigned)nsplit, arr, ext) : launch_rs<R, true, false>(A, st, ncols, (unsigned)nsplit, arr, ext);
    return sym ? launch_rs<R, false, true>(A, st, ncols, (unsigned)nsplit, arr, ext) : launch_rs<R, false, false>(A, st, ncols, (unsigned)nsplit, arr, ext);
}

// Rings wider than 33 taps exist for the all-valid pass only (symmetric kernels: the distinct taps have to fit the
// SGPR file; one spaxel per lane: R float64 accumulators + R staged inputs per lane): dirty tiles go to the
// runs-of-16 kernel of spc_spectral_conv.hip.  A.status must point at zeroed tile flags.
template <int R>
int launch_fast_only(const ConvArgs& A, hipStream_t st) {
    const int64_t ngroups = A.ny * A.nx;
    hipLaunchKernelGGL((spectral_conv_fast_kernel<R, false, 1, true>), dim3((unsigned)((ngroups + 255) / 256), 1), dim3(256), 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

}  // namespace spc_sconv
