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
// (tools/micro/valu_rate.hip): a wave64 VALU instruction occupies its SIMD for
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
    int skip_dead;             // general ring kernel with a mask array: revolutions without an included sample in the wave are skipped
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
    float pred_lim, pred_lo, pred_hi, pred_pad;   // spc_canonical_pred of the mask's predicate terms
    alignas(16) double k[72];  // taps padded to R, centred (R <= 65)
};

// ---- denominators from validity bits ------------------------------------------------------
// Bit b of a lane's history = the sample that arrived b steps ago was INVALID (0 = valid; out-of-range samples are
// valid zeros, boundary='fill'); when an output completes, that sample carried tap k[b] (the newest sample takes
// k[0], see the ring).  Two 32-bit words (bits 0..31, bit 32): 64-bit shifts are not single instructions.
constexpr int kLutBits = 11, kLutSize = 1 << kLutBits;
constexpr int lut_tables(int R) { return (R + kLutBits - 1) / kLutBits; }
// threads per block of the general kernel: the tables are per block (48 KB for 33 taps), so the block size sets
// how many waves share a CU: 2 x 512 threads (4 waves per SIMD) where the registers allow it (no mask array, not
// fused), 3 x 256 threads otherwise
constexpr int general_block(bool arr, bool fuse) { return (arr || fuse) ? 256 : 512; }

// table t, entry w: sum of the taps 11 t + b whose bit b of w is CLEAR (valid)
template <int R>
__device__ __forceinline__ void lut_build(const ConvArgs& A, double* lut) {
#pragma unroll
    for (int t = 0; t < lut_tables(R); ++t) {
        for (int w = threadIdx.x; w < kLutSize; w += blockDim.x) {
            double acc = 0.0;
#pragma unroll
            for (int b = 0; b < kLutBits; ++b)
                if (kLutBits * t + b < R) acc += ((w >> b) & 1) ? 0.0 : A.k[kLutBits * t + b];
            lut[t * kLutSize + w] = acc;
        }
    }
}

// byte-offset mask of a table of which `bits` (<= 11) index bits are in use
constexpr unsigned lut_mask(int bits) { return (((bits >= kLutBits) ? (unsigned)kLutSize : (bits > 0 ? (1u << (bits > 0 ? bits : 0)) : 1u)) - 1u) * 8u; }

template <int R>
__device__ __forceinline__ double lut_den(const double* lut, unsigned lo, unsigned hi) {
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

// num / den for a looked-up denominator (a sum of taps: 1e-5 .. 1 times the kernel sum, never subnormal): the hardware
// reciprocal (2^-23 relative), one Newton step (2^-46), the quotient and ONE residual step (exact in the FMA) - the
// step every software division ends with; ~2^-90 from the rounded quotient, i.e. the same float32 after the final
// rounding except within 2^-90 of a rounding boundary.  10 VALU slots (v_rcp_f64 is quarter rate) against the ~17 of
// the compiler's IEEE division (two v_div_scale, two Newton steps, v_div_fmas, v_div_fixup).  den = 0 (an empty
// window: num = 0 as well) gives NaN, which is astropy's answer for an invalid centre sample.
__device__ __forceinline__ double div_den(double num, double den) {
    double y = __builtin_amdgcn_rcp(den);
    y = fma(fma(-den, y, 1.0), y, y);
    const double q = num * y;
    const double r = fmax(fma(-q, den, num), -1.7976931348623157e308);     // (infinite num: see div_ksum)
    return fma(r, y, q);
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

// ---- general kernel (masks / NaNs) -----------------------------------------------------------
// One lane = one spaxel marching over z; a revolution = R input planes.  The loads of a revolution are issued
// up front (one descriptor + a scalar plane offset each); every ring step then classifies ITS sample, feeds the R
// numerators, and emits the output that just completed.  Classification lives inside the step on purpose: done as a
// separate pass, the R per-lane validity masks (SGPR pairs) and the R uniform in-range conditions stayed live across
// the whole ring - 431 spilled SGPRs, ~18 v_readlane / v_writelane per voxel next to the 33 FMAs, and another ~60
// scalar instructions per voxel (rounds 1 - 2: 102 VALU instructions per voxel, 145 fused; now ~62 / ~75).
// Out-of-range samples (the first / last revolution of a z slice) are valid zeros; that bookkeeping sits behind a
// wave-uniform branch and costs interior revolutions nothing.
template <int R, bool ARR, bool FUSE, bool EXT, bool SYM>
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
    // fused: the steps read the channel coordinates from lanes 0 .. R - 1 of the wave (v_readlane), so a wave keeps
    // ALL its lanes marching as long as one of them is live; the others redo the last spaxel and write nothing
    if (FUSE ? !__any(live) : !live) return;
    const int64_t colc = min(col, A.ny * A.nx - 1);
    const int64_t y = colc / A.nx, x = colc - y * A.nx;
    const int nz = (int)A.nz;
    const int zb = (int)(blockIdx.y * A.zchunk);
    const int ze = min(nz, zb + (int)A.zchunk);
    const int voff_out = FUSE ? 0 : (int)((y * A.out_row_stride + x) * 4);
    const uint32_t flags = A.mask.flags;
    // predicate terms (isfinite / thresholds) in their three-compare canonical form; without any, a sample the array
    // term includes may still be NaN: included (it counts for the fused reduction's mask) but not valid
    const bool has_pred = (flags & (SPC_MASK_FINITE | SPC_MASK_GT | SPC_MASK_GE | SPC_MASK_LT | SPC_MASK_LE)) != 0;   // uniform
    const float plim = A.pred_lim, plo = A.pred_lo, phi = A.pred_hi;
    const int voff = (int)((y * A.row_stride + x) * 4);              // < 2 GiB per plane (checked on the host)
    const int moff = ARR ? (int)(y * A.mask.row_stride + x) : 0;
    double num[R];
#pragma unroll
    for (int m = 0; m < R; ++m) num[m] = 0.0;
    unsigned bad_lo = 0u, bad_hi = 0u;   // invalid bit of the last 33 inputs, bit 0 = newest (what lies before the slice start is never emitted)
    unsigned inc_lo = 0u;                // include bit of the last 32 inputs (the fused reduction asks for the one of age H <= 16)
    // Round 5: a signal mask (data > n sigma) leaves long runs of channels in which NO lane of the wave has its mask byte set.
    // A revolution (R channels) whose mask bytes are all zero, behind another such revolution, is skipped whole: its outputs'
    // windows lie inside the two - NaN, nothing for the moment sums - and the numerator slots it would open already hold the
    // zeros the previous dead revolution left.  Decided once per revolution from the OR of its R mask bytes (R vector
    // instructions); a first version that tested every STEP cut the hot loop into R basic blocks and cost the noise-level
    // mask of the bench 15 % (19.5 -> 22.7 ms at C3).
    bool prev_dead = false;
    MomState ms;
    const int pbytes = (int)(A.plane_stride * 4), mbytes = ARR ? (int)A.mask.plane_stride : 0;
    const int obytes = FUSE ? 0 : (int)(A.out_plane_stride * 4);

    const int T = (ze - zb) + 2 * H;      // number of input steps
    for (int t0 = 0; t0 < T; t0 += R) {
        const int i0 = __builtin_amdgcn_readfirstlane(zb - H + t0);   // first input channel of this revolution
        float v[R];
        unsigned mk[R];
        // ONE descriptor per revolution (base = first plane read) + a scalar byte offset per load
        const int pb = min(max(i0, 0), nz - 1);
        const auto rs = plane_srd(A.cube + (int64_t)pb * A.plane_stride);
        const auto rm = plane_srd(ARR ? (const void*)(A.mask.arr + (int64_t)pb * A.mask.plane_stride) : (const void*)A.cube);
        const bool edge = (i0 < 0) || (i0 + R > nz);                  // uniform: some sample of this revolution is out of range
        if (!edge) {
#pragma unroll
            for (int s = 0; s < R; ++s) {
                v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)((unsigned)s * (unsigned)pbytes), /*nt*/ 2));
                if (ARR) mk[s] = __builtin_amdgcn_raw_buffer_load_b8(rm, moff, (int)((unsigned)s * (unsigned)mbytes), 2);
            }
        } else {
#pragma unroll
            for (int s = 0; s < R; ++s) {
                const int dz = min(max(i0 + s, 0), nz - 1) - pb;                 // clamped, uniform
                const bool in = (i0 + s >= 0) && (i0 + s < nz);                  // uniform
                const float ld = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)((unsigned)dz * (unsigned)pbytes), /*nt*/ 2));
                v[s] = in ? ld : 0.f;                                             // out of range = a valid zero (boundary='fill')
                if (ARR) mk[s] = __builtin_amdgcn_raw_buffer_load_b8(rm, moff, (int)((unsigned)dz * (unsigned)mbytes), 2);
            }
        }
        // outputs of this revolution: planes i0 - H .. i0 + H, addressed from the first one that exists
        const int ob = max(i0 - H, 0);
        const auto ro = plane_srd(FUSE ? (const void*)A.cube : (const void*)(A.out + (int64_t)ob * A.out_plane_stride));
        const bool emit_all = (i0 - H >= zb) && (i0 + R - 1 - H < ze);          // uniform
        if (ARR && A.skip_dead) {
            unsigned anym = 0u;
#pragma unroll
            for (int s = 0; s < R; ++s) anym |= mk[s];
            const bool dead = !edge && !__any(anym != 0u);                      // uniform: no lane includes any sample of this revolution
            if (dead && prev_dead && emit_all) {
                if (!FUSE) {
#pragma unroll
                    for (int s = 0; s < R; ++s)
                        __builtin_amdgcn_raw_buffer_store_b32(0x7fc00000u, ro, voff_out, (int)((unsigned)(i0 + s - H - ob) * (unsigned)obytes), 0);
                }
                bad_lo = 0xffffffffu; bad_hi = 0xffffffffu; inc_lo = 0u;          // R invalid, excluded samples went by
                continue;
            }
            prev_dead = dead;
        }
        // fused reduction: the R channel coordinates of this revolution's outputs in ONE vector load (lane l holds
        // c[i0 - H + l]); a step takes its own with two v_readlane - a scalar load per output would put a memory
        // round trip (and a wait on the counter the table reads share) on every step
        double cvec = 0.0;
        if (FUSE) cvec = A.cen[min(max(i0 - H + (int)(threadIdx.x & 63), 0), nz - 1)];
#pragma unroll
        for (int s = 0; s < R; ++s) {
            // ---- classify sample s: value | invalid | out of range (= valid zero)
            float vs = v[s];
            // (without predicate terms lim = +inf and lo = hi = NaN: the three compares then only reject NaN)
            const bool arrbit = ARR ? (mk[s] != 0) : true;
            const bool ok = arrbit && (__builtin_fabsf(vs) <= plim) && !(vs <= plo) && !(vs >= phi);
            // included by the mask (what the fused reduction's mask asks): a predicate term rejects NaN, so included ==
            // valid; without one an included sample may still be NaN
            const bool inc = FUSE ? (ok || (!has_pred && arrbit)) : ok;       // (mask logic: scalar and / or of the lane masks)
            float xs = ok ? vs : 0.f;                                 // (an out-of-range sample was loaded as 0: it adds nothing)
            unsigned badbit = ok ? 0u : 1u, incbit = inc ? 1u : 0u;   // as VECTOR integers: no lane mask stays live
            if (edge) {                                               // uniform, and kept a BRANCH (the empty asm cannot be
                asm volatile("");                                     // speculated): interior revolutions pay one s_cbranch
                if (!((i0 + s >= 0) && (i0 + s < nz))) { badbit = 0u; incbit = 0u; }   // uniform: a valid zero, not part of the cube
            }
            asm volatile("" : "+v"(xs));                              // select in float32, THEN widen (one v_cndmask, not two)
            const double xd = (double)xs;
            bad_hi = __builtin_amdgcn_alignbit(bad_hi, bad_lo, 31);
            bad_lo = (bad_lo << 1) | badbit;
            if (FUSE) inc_lo = (inc_lo << 1) | incbit;
#pragma unroll
            for (int m = 0; m < R; ++m) {
                const int a = (s - m + R) % R;          // age of the output living in slot m
                // symmetric kernels: half the distinct weights, all of them stay in SGPRs
                const int j = SYM ? (a <= H ? a : 2 * H - a) : 2 * H - a;
                if (a == 0) mul_w(num[m], A, j, xd);
                else fma_w(num[m], A, j, xd);
            }
            // ---- the output that just received its last contribution
            const int e = (s + 1) % R;
            const int o = i0 + s - H;
            if (emit_all || (o >= zb && o < ze)) {
                // every sample of this output's window valid, in every lane: the denominator is the whole kernel.  Tested
                // only without a mask array (the dirty tiles of the all-valid pass: sparse NaNs in mostly clean data) - a
                // cube that brings its own mask array practically never has 64 clean windows side by side, and the test
                // costs every output three vector instructions and a branch
                const unsigned anybad = (R > 32) ? ((bad_hi & 1u) | bad_lo) : (bad_lo & (unsigned)((1ull << R) - 1ull));
                float res;
                if (!ARR && __all(anybad == 0u)) {
                    res = (float)div_ksum(num[e], A);
                    asm volatile("; whole kernel" : "+v"(res));       // (distinct tails: merged, the scalar 1 / sum(k) of this
                } else {                                              //  arm is copied to vector registers for every output)
                    // astropy returns the (filled) centre sample for an empty window; the host only dispatches kernels
                    // with a non-zero centre tap here, for which an empty window implies an invalid centre: NaN = 0 / 0
                    res = (float)div_den(num[e], lut_den<R>(lut, bad_lo, bad_hi));
                    asm volatile("; looked-up denominator" : "+v"(res));
                }
                if (!FUSE) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res), ro, voff_out, (int)((unsigned)(o - ob) * (unsigned)obytes), 0);
                } else {
                    // (no mask at all: every in-range channel carries an include bit)
                    const bool inc_o = ((inc_lo >> H) & 1u) != 0u;
                    const bool okm = inc_o && (res == res);
                    float wf = okm ? res : 0.f;
                    asm volatile("" : "+v"(wf));                    // select in float32, then widen
                    const double wd = (double)wf;
                    const unsigned long long cb = __builtin_bit_cast(unsigned long long, cvec);
                    const double c = __builtin_bit_cast(double, ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(cb >> 32), s) << 32) |
                                                                    (unsigned)__builtin_amdgcn_readlane((int)cb, s));
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
    }

    if (FUSE && live) {
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

// ---- general kernel of the 49- / 65-tap rings (masks / NaNs, symmetric kernels; round 5) ---------------------------
// The ring of spectral_conv_kernel with what 49 / 65 float64 numerators per lane leave room for (2 waves per SIMD):
//   * the inputs of a revolution arrive in FOUR chunks (13 13 13 10 / 17 17 17 14 samples) through two alternating
//     sets of staging registers - the chunk after the one being consumed is in flight, across the revolution boundary
//     as well - instead of R staged samples + R mask bytes;
//   * the validity history is 65 bits in three words; the denominator tables take 10 bits each (5 / 7 tables of
//     1024 float64 sums: 40 / 56 KB, two blocks per CU);
//   * the distinct taps (25 / 33 SGPR pairs) exist for symmetric kernels only: others stay with the runs-of-16 kernel.
// Honours the tile flags of the all-valid ring pass.  Not fused.  Before: spectral_conv_wide_kernel, 13.5 / 16.1 ms per
// 1024^3 at 49 / 65 taps (two FMAs per tap and voxel, a (ntaps - 1)-plane halo per run of 16).
constexpr int kWideLutBits = 10;
constexpr int wide_chunk(int R) { return (R + 3) / 4; }

// 32 history bits starting at bit P (b0: bits 0..31, b1: 32..63, b2: 64..)
template <int P>
__device__ __forceinline__ unsigned hist_from(unsigned b0, unsigned b1, unsigned b2) {
    if (P == 0) return b0;
    if (P < 32) return __builtin_amdgcn_alignbit(b1, b0, P);
    if (P == 32) return b1;
    if (P < 64) return __builtin_amdgcn_alignbit(b2, b1, P - 32);
    return b2;
}

template <int R, int T>
__device__ __forceinline__ double wide_lut_term(const double* lut, unsigned b0, unsigned b1, unsigned b2) {
    constexpr int LB = kWideLutBits, P = LB * T, bits = (R - P) < LB ? (R - P) : LB;
    constexpr unsigned m = ((1u << bits) - 1u) << 3;
    const unsigned off = (P >= 3) ? (hist_from<(P >= 3 ? P - 3 : 0)>(b0, b1, b2) & m) : ((b0 << 3) & m);
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(lut) + (T << LB) * 8 + off);
}

template <int R>
__device__ __forceinline__ double wide_lut_den(const double* lut, unsigned b0, unsigned b1, unsigned b2) {
    constexpr int NT = (R + kWideLutBits - 1) / kWideLutBits;
    double den = wide_lut_term<R, 0>(lut, b0, b1, b2);
    den += wide_lut_term<R, 1>(lut, b0, b1, b2);
    den += wide_lut_term<R, 2>(lut, b0, b1, b2);
    den += wide_lut_term<R, 3>(lut, b0, b1, b2);
    den += wide_lut_term<R, 4>(lut, b0, b1, b2);
    if (NT > 5) den += wide_lut_term<R, (NT > 5 ? 5 : 0)>(lut, b0, b1, b2);
    if (NT > 6) den += wide_lut_term<R, (NT > 6 ? 6 : 0)>(lut, b0, b1, b2);
    return den;
}

// loads of N samples starting at input channel ch0 (clamped into the cube: what lies outside is fixed up by the consumer)
template <int G, int N, bool ARR>
__device__ __forceinline__ void wide_load(const ConvArgs& A, float (&v)[G], unsigned (&mk)[G], int ch0, int nz, int voff, int moff,
                                          int pbytes, int mbytes) {
    const int pb = min(max(ch0, 0), nz - 1);
    const auto rs = plane_srd(A.cube + (int64_t)pb * A.plane_stride);
    const auto rm = plane_srd(ARR ? (const void*)(A.mask.arr + (int64_t)pb * A.mask.plane_stride) : (const void*)A.cube);
    if (ch0 >= 0 && ch0 + N <= nz) {
#pragma unroll
        for (int u = 0; u < N; ++u) {
            v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)((unsigned)u * (unsigned)pbytes), /*nt*/ 2));
            if (ARR) mk[u] = __builtin_amdgcn_raw_buffer_load_b8(rm, moff, (int)((unsigned)u * (unsigned)mbytes), 2);
        }
    } else {
#pragma unroll
        for (int u = 0; u < N; ++u) {
            const int dz = min(max(ch0 + u, 0), nz - 1) - pb;
            v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)((unsigned)dz * (unsigned)pbytes), /*nt*/ 2));
            if (ARR) mk[u] = __builtin_amdgcn_raw_buffer_load_b8(rm, moff, (int)((unsigned)dz * (unsigned)mbytes), 2);
        }
    }
}

template <int R, bool ARR>
__global__ __launch_bounds__(256, 2) void spectral_conv_ring_wide_kernel(const ConvArgs A) {
    constexpr int H = R / 2, LB = kWideLutBits, NT = (R + LB - 1) / LB, G = wide_chunk(R), NC = 4;
    static_assert(R > 40 && R <= 65 && G * (NC - 1) < R && NT >= 5 && NT <= 7, "49- and 65-tap rings");
    __shared__ double lut[NT << LB];
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = (col < A.ny * A.nx) &&
                      !(A.status && spc_flag_get(A.status + __builtin_amdgcn_readfirstlane((int)(min(col, A.ny * A.nx - 1) >> 7))) == 0);
    if (!__syncthreads_or(live ? 1 : 0)) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        for (int w = threadIdx.x; w < (1 << LB); w += blockDim.x) {
            double acc = 0.0;
#pragma unroll
            for (int b = 0; b < LB; ++b)
                if (LB * t + b < R) acc += ((w >> b) & 1) ? 0.0 : A.k[(LB * t + b) <= H ? (LB * t + b) : 2 * H - (LB * t + b)];
            lut[(t << LB) + w] = acc;
        }
    }
    __syncthreads();
    if (!live) return;
    const int64_t y = col / A.nx, x = col - y * A.nx;
    const int nz = (int)A.nz;
    const int zb = (int)(blockIdx.y * A.zchunk);
    const int ze = min(nz, zb + (int)A.zchunk);
    const int voff_out = (int)((y * A.out_row_stride + x) * 4);
    const float plim = A.pred_lim, plo = A.pred_lo, phi = A.pred_hi;
    const int voff = (int)((y * A.row_stride + x) * 4);
    const int moff = ARR ? (int)(y * A.mask.row_stride + x) : 0;
    const int pbytes = (int)(A.plane_stride * 4), mbytes = ARR ? (int)A.mask.plane_stride : 0;
    const int obytes = (int)(A.out_plane_stride * 4);
    double num[R];
#pragma unroll
    for (int m = 0; m < R; ++m) num[m] = 0.0;
    unsigned b0 = 0u, b1 = 0u, b2 = 0u;          // invalid bit of the last 65 inputs, bit 0 = newest
    float va[G], vb[G];
    unsigned ma[G], mb[G];
#pragma unroll
    for (int u = 0; u < G; ++u) { va[u] = 0.f; vb[u] = 0.f; ma[u] = 1u; mb[u] = 1u; }

    const int T = (ze - zb) + 2 * H;
    wide_load<G, G, ARR>(A, va, ma, zb - H, nz, voff, moff, pbytes, mbytes);
    for (int t0 = 0; t0 < T; t0 += R) {
        const int i0 = __builtin_amdgcn_readfirstlane(zb - H + t0);
        const bool edge = (i0 < 0) || (i0 + R > nz);
        const int ob = max(i0 - H, 0);
        const auto ro = plane_srd((const void*)(A.out + (int64_t)ob * A.out_plane_stride));
        const bool emit_all = (i0 - H >= zb) && (i0 + R - 1 - H < ze);
#pragma unroll
        for (int s = 0; s < R; ++s) {
            constexpr int dummy = 0; (void)dummy;
            const int c = s / G, u = s % G;
            if (u == 0) {
                // the chunk after this one: into the other set of staging registers
                if (c == 0) wide_load<G, G, ARR>(A, vb, mb, i0 + G, nz, voff, moff, pbytes, mbytes);
                else if (c == 1) wide_load<G, G, ARR>(A, va, ma, i0 + 2 * G, nz, voff, moff, pbytes, mbytes);
                else if (c == 2) wide_load<G, R - 3 * G, ARR>(A, vb, mb, i0 + 3 * G, nz, voff, moff, pbytes, mbytes);
                else if (t0 + R < T) wide_load<G, G, ARR>(A, va, ma, i0 + R, nz, voff, moff, pbytes, mbytes);
            }
            const float vs = (c & 1) ? vb[u] : va[u];
            const bool arrbit = ARR ? (((c & 1) ? mb[u] : ma[u]) != 0) : true;
            const bool ok = arrbit && (__builtin_fabsf(vs) <= plim) && !(vs <= plo) && !(vs >= phi);
            float xs = ok ? vs : 0.f;
            unsigned badbit = ok ? 0u : 1u;
            if (edge) {
                asm volatile("");
                if (!((i0 + s >= 0) && (i0 + s < nz))) { badbit = 0u; xs = 0.f; }   // outside the cube: a valid zero
            }
            asm volatile("" : "+v"(xs));
            const double xd = (double)xs;
            if (R > 64) b2 = __builtin_amdgcn_alignbit(b2, b1, 31);
            b1 = __builtin_amdgcn_alignbit(b1, b0, 31);
            b0 = (b0 << 1) | badbit;
#pragma unroll
            for (int m = 0; m < R; ++m) {
                const int a = (s - m + R) % R;
                const int j = a <= H ? a : 2 * H - a;
                if (a == 0) mul_w(num[m], A, j, xd);
                else fma_w(num[m], A, j, xd);
            }
            const int e = (s + 1) % R;
            const int o = i0 + s - H;
            if (emit_all || (o >= zb && o < ze)) {
                const unsigned anybad = (R > 64) ? (b0 | b1 | (b2 & 1u)) : (b0 | (b1 & (unsigned)((1ull << (R - 32)) - 1ull)));
                float res;
                if (!ARR && __all(anybad == 0u)) {
                    res = (float)div_ksum(num[e], A);
                    asm volatile("; whole kernel" : "+v"(res));
                } else {
                    res = (float)div_den(num[e], wide_lut_den<R>(lut, b0, b1, b2));
                    asm volatile("; looked-up denominator" : "+v"(res));
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res), ro, voff_out, (int)((unsigned)(o - ob) * (unsigned)obytes), 0);
            }
        }
    }
}

template <int R>
int launch_ring_wide(const ConvArgs& A, hipStream_t st) {
    const int64_t ncols = A.ny * A.nx;
    const dim3 grid((unsigned)((ncols + 255) / 256), (unsigned)((A.nz + A.zchunk - 1) / A.zchunk)), block(256);
    if (A.mask.flags & SPC_MASK_ARRAY) hipLaunchKernelGGL((spectral_conv_ring_wide_kernel<R, true>), grid, block, 0, st, A);
    else hipLaunchKernelGGL((spectral_conv_ring_wide_kernel<R, false>), grid, block, 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
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
