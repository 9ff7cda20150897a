// Masked separable spatial_smooth with the denominator on the matrix cores, fused with moment 0
// (spc_spatial_conv_sep_mfma_f32; reference: dask_spectral_cube.py:962-993 - the per-channel astropy convolution with
// boundary='fill', nan_treatment='interpolate' - followed by the nansum of :1083-1104 under the ORIGINAL mask).
//
//   out[z, y, x] = sum_ij ky[i] kx[j] d m / sum_ij ky[i] kx[j] m            (m: validity, 0 / 1; outside the plane: d = 0, m = 1)
//
// The ring kernels of spc_spatial_conv_impl.h carry numerator and denominator together through the vector ALU: 58
// packed FMAs per voxel, and that - not memory - is what bounds them (profiles/r03_masked_spatial_ablation_after.log).
// Here the vector ALU only does the numerator, two COLUMNS per packed FMA (29 instead of 58 issue slots per voxel); the
// denominator is a convolution of a 0 / 1 array and goes to the matrix pipe, which runs beside the vector pipe:
//
//   den^T = Tx . (V^T . Ty^T)        V: validity bits of the band, Ty / Tx: banded Toeplitz matrices of the taps
//
// on v_mfma_f32_16x16x32_f16 with float32 accumulation.  V is exact in fp16.  The taps are scaled by a power of two
// (so that the far Gaussian tail stays a NORMAL fp16 number) and split into fp16 hi + lo; the intermediate Q = V^T Ty^T
// (float32 accumulators) is split the same way: 2 + 3 products per K step, ~22 significant bits end to end
// (tests: <= 1e-6 relative on the denominator against float64).  The C layout of the first product IS the B layout of
// the second (same lane, same output row), so Q never leaves the registers.
//
// One block = a band of 16 output rows x a strip of 480 output columns (512 input columns, two per lane), walking a
// chunk of channels; per channel:  y pass (44 input rows -> 16 rows, static taps, parked in LDS as row pairs) | barrier |
// matrix phase (validity bits from LDS -> den in LDS) | barrier | x pass + division (+ moment-0 sums kept in registers
// across the chunk) | barrier.  Two blocks share a CU: while one is in its matrix phase the other issues FMAs.
#include "spc_common.h"
#include <algorithm>
#include <cmath>

namespace {

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int R = 29, H = 14;
constexpr int kThreads = 256;
constexpr int kCols = 512;                    // input columns per strip (2 per lane)
constexpr int kTxo = 480;                     // output columns per strip = 30 matrix tiles of 16
constexpr int kBand = 16;                     // output rows per block and channel
constexpr int kInRows = kBand + 2 * H;        // 44
constexpr int kRun = 8;
constexpr int kPitch2 = kCols + kCols / 8;    // float2 per LDS pair row, 9-per-8 padded
constexpr int kDenPitch = kTxo + 4;           // floats per den row: 484 = 36 (mod 64) spreads the 16 rows of a tile store over the banks
constexpr int kBitGroups = 8;                 // 8-row groups of validity bits the K = 64 window of the first product spans

struct SmArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    const uint8_t* marr;                      // uint8 mask array or nullptr
    int64_t mrow_stride, mplane_stride;
    float* out;                               // smoothed cube or nullptr
    int64_t out_row_stride, out_plane_stride;
    float* partial;                           // (nchunk, ny, nx) float32 sums of a chunk, or nullptr (no moment)
    unsigned char* seen;                      // (nchunk, ny, nx) "a channel contributed"
    int nstrips, nbands, zchunk, nchunk;
    float lim;                                // FLT_MAX under isfinite, +inf otherwise (NaN fails |v| <= lim either way)
    float sy, sx, inv_scale;                  // power-of-two scales of the fp16 taps; 1 / (sy * sx)
    alignas(8) float ky[R + 3];
    alignas(8) float kx[R + 3];
};

__device__ __forceinline__ void pk_fma_w(float2v& acc, const float* karr, int j, float2v x) {
    const float2v wp = *reinterpret_cast<const float2v*>(&karr[j & ~1]);
    if (j & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(wp), "v"(x));
    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(wp), "v"(x));
}
__device__ __forceinline__ void pk_mul_w(float2v& acc, const float* karr, int j, float2v x) {
    const float2v wp = *reinterpret_cast<const float2v*>(&karr[j & ~1]);
    if (j & 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(acc) : "s"(wp), "v"(x));
    else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "s"(wp), "v"(x));
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (chunk, strip, band) of this block: blocks are dealt round-robin to the 8 XCDs; the remap hands every XCD a
// contiguous range of work items with the band running fastest, so the bands that share halo rows of a plane run on
// one XCD (one L2) at about the same time
__device__ __forceinline__ void block_item(const SmArgs& A, int& chunk, int& strip, int& band) {
    const int64_t n = (int64_t)gridDim.x, b = blockIdx.x;
    const int64_t q = n / 8, r = n % 8, xcd = b % 8, i = b / 8;
    const int64_t w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    band = (int)(w % A.nbands);
    strip = (int)((w / A.nbands) % A.nstrips);
    chunk = (int)(w / ((int64_t)A.nbands * A.nstrips));
}

__device__ __forceinline__ half8 make_half8(u32x2 a, u32x2 b) {
    const u32x4 v = {a.x, a.y, b.x, b.y};
    return __builtin_bit_cast(half8, v);
}
// float32 x 4 -> fp16 hi (2 registers) and fp16 lo = the residual (2 registers)
__device__ __forceinline__ void split4(f32x4 q, u32x2& hi, u32x2& lo) {
    const half2v h0 = {(_Float16)q.x, (_Float16)q.y}, h1 = {(_Float16)q.z, (_Float16)q.w};
    const half2v l0 = {(_Float16)(q.x - (float)h0.x), (_Float16)(q.y - (float)h0.y)};
    const half2v l1 = {(_Float16)(q.z - (float)h1.x), (_Float16)(q.w - (float)h1.y)};
    hi = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
    lo = u32x2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
}

template <bool ARR, bool STORE, bool MOM>
__global__ __launch_bounds__(kThreads, 2) void spatial_sep_mfma_kernel(const SmArgs A) {
    __shared__ float2v ybuf[2][4][kPitch2];                   // the band's 16 finished y-pass rows as 8 row pairs (36 KB)
    __shared__ float den[kBand * kDenPitch];                  // (31 KB)
    __shared__ __attribute__((aligned(16))) unsigned char cbits[(kBitGroups + 1) * kCols]; // conv-validity bits, byte = 8 rows of one column (+ a spill row)
    __shared__ __attribute__((aligned(16))) unsigned char ibits[(kBitGroups + 1) * kCols]; // include bits (the ORIGINAL mask: what the moment sums over)
    __shared__ u32x2 lut[16];                                 // nibble -> four fp16 of 0.0 / 1.0
    __shared__ float taps[2][32];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    int chunk, strip, band;
    block_item(A, chunk, strip, band);
    const int64_t x0 = (int64_t)strip * kTxo;                 // first output column
    const int y0 = band * kBand;                              // first output row
    const int ny = (int)A.ny;
    const int64_t xin = x0 - H + 2 * t;                       // first of this lane's two input columns (even)
    const bool col_in = (xin >= 0) && (xin + 1 < A.nx);
    const int64_t xc = min(max(xin, (int64_t)0), A.nx - 2);
    const bool edge_cols = (x0 - H < 0) || (x0 - H + kCols > A.nx);
    const bool edge_rows = (y0 - H < 0) || (y0 - H + kInRows > ny);
    const int z_begin = chunk * A.zchunk, z_end = (int)min((int64_t)z_begin + A.zchunk, A.nz);

    if (t < 16) {
        const unsigned a = ((t & 1) ? 0x3C00u : 0u) | ((t & 2) ? 0x3C000000u : 0u);
        const unsigned b = ((t & 4) ? 0x3C00u : 0u) | ((t & 8) ? 0x3C000000u : 0u);
        lut[t] = u32x2{a, b};
    }
    if (t < 32) { taps[0][t] = t < R ? A.ky[t] * A.sy : 0.f; taps[1][t] = t < R ? A.kx[t] * A.sx : 0.f; }
    for (int i = t; i < (kBitGroups + 1) * kCols; i += kThreads) { cbits[i] = 0; ibits[i] = 0; }
    __syncthreads();

    // ---- Toeplitz operands of this lane (constant over the whole kernel)
    // first product  Q^T[in col][out row] = sum_k V^T[in col][k] Ty^T[k][out row], k = band-relative input row:
    //   B operand, lane (n = lane & 15: out row, g = lane >> 4), step s, slot e: k = 32 s + 8 g + e, weight ky[n + 28 - k]
    // second product den^T[out col][out row] = sum_kc Tx[out col][kc] Q^T[kc][out row], kc = input column relative to the tile:
    //   A operand, lane (m = lane & 15: out col, g), step s, slot e: kc = 32 s + (e < 4 ? 4 g + e : 16 + 4 g + e - 4)
    //   (= the input columns the accumulator registers of Q tiles 2 s and 2 s + 1 hold in this lane), weight kx[m + 28 - kc]
    const int ln = lane & 15, lg = lane >> 4;
    half8 By_hi[2], By_lo[2], Ax_hi[2], Ax_lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * s + 8 * lg + e;
            const int jy = ln + 2 * H - k;
            const float wy = (jy >= 0 && jy <= 2 * H) ? taps[0][jy] : 0.f;
            const _Float16 hy = (_Float16)wy;
            By_hi[s][e] = hy;
            By_lo[s][e] = (_Float16)(wy - (float)hy);
            const int kc = 32 * s + (e < 4 ? 4 * lg + e : 16 + 4 * lg + (e - 4));
            const int jx = ln + 2 * H - kc;
            const float wx = (jx >= 0 && jx <= 2 * H) ? taps[1][jx] : 0.f;
            const _Float16 hx = (_Float16)wx;
            Ax_hi[s][e] = hx;
            Ax_lo[s][e] = (_Float16)(wx - (float)hx);
        }
    }

    // x-pass tasks of this lane: task = t + 256 q -> (row pair pr8 of the band, run j of 8 columns); fixed over the channels,
    // so the moment sums of its 16 outputs stay in registers
    float2v msum[2][kRun];
    unsigned mseen[2] = {0u, 0u};
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < kRun; ++k) msum[q][k] = float2v{0.f, 0.f};
    const int c0 = 2 * t, c1 = 2 * t + 1;
    const int ph0 = c0 + (c0 >> 3), ph1 = c1 + (c1 >> 3);

    float2v v[2][8];
    unsigned mk[2][8];
    auto fetch = [&](int slot, int64_t z, int g) {            // rows 8 g .. 8 g + 7 of the band's 44 input rows
        const float* p = A.cube + z * A.plane_stride + xc;
        const uint8_t* pm = ARR ? A.marr + z * A.mplane_stride + xc : nullptr;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (8 * g + s < kInRows) {
                const int64_t ic = min(max(y0 - H + 8 * g + s, 0), ny - 1);
                v[slot][s] = __builtin_nontemporal_load(reinterpret_cast<const float2v*>(p + ic * A.row_stride));
                if (ARR) mk[slot][s] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(pm + ic * A.mrow_stride));
            }
        }
    };

    if (z_begin < z_end) fetch(0, z_begin, 0);
    for (int z = z_begin; z < z_end; ++z) {
        // ================= y pass: 44 input rows -> 16 rows, two columns per lane =================
        float2v acc[kBand];
        float2v held = float2v{0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (g + 1 < 6) fetch((g + 1) & 1, z, g + 1);      // the next group is in flight during this group's FMAs
            unsigned cb0 = 0, cb1 = 0, ib0 = 0, ib1 = 0;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const int s = 8 * g + s8;
                if (s < kInRows) {
                    const float2v d = v[g & 1][s8];
                    bool m0 = true, m1 = true;
                    if (ARR) { m0 = (mk[g & 1][s8] & 0xffu) != 0; m1 = (mk[g & 1][s8] & 0xff00u) != 0; }
                    // include = the cube's mask on its own voxel (array term and, under isfinite, |v| <= FLT_MAX);
                    // conv-valid = include and not NaN (astropy interpolates over NaN whatever the mask says)
                    bool i0 = m0 && (A.lim == INFINITY || __builtin_fabsf(d.x) <= A.lim);
                    bool i1 = m1 && (A.lim == INFINITY || __builtin_fabsf(d.y) <= A.lim);
                    bool ok0 = m0 && __builtin_fabsf(d.x) <= A.lim, ok1 = m1 && __builtin_fabsf(d.y) <= A.lim;
                    if (edge_cols || edge_rows) {            // out of bounds = a valid zero that is not a voxel
                        const bool in = col_in && (y0 - H + s >= 0) && (y0 - H + s < ny);
                        if (!in) { ok0 = ok1 = true; i0 = i1 = false; }
                    }
                    float2v dm = float2v{ok0 ? d.x : 0.f, ok1 ? d.y : 0.f};
                    if ((edge_cols || edge_rows) && !(col_in && (y0 - H + s >= 0) && (y0 - H + s < ny))) dm = float2v{0.f, 0.f};
                    cb0 |= (ok0 ? 1u : 0u) << s8; cb1 |= (ok1 ? 1u : 0u) << s8;
                    ib0 |= (i0 ? 1u : 0u) << s8; ib1 |= (i1 ? 1u : 0u) << s8;
#pragma unroll
                    for (int o = 0; o < kBand; ++o) {
                        const int a = s - o;              // tap distance: weight ky[2H - a]
                        if (a == 0) pk_mul_w(acc[o], A.ky, 2 * H, dm);
                        else if (a > 0 && a <= 2 * H) pk_fma_w(acc[o], A.ky, 2 * H - a, dm);
                    }
                    if (s >= 2 * H) {                        // output row o = s - 2H is complete
                        const int o = s - 2 * H;
                        const float2v done = acc[o];
                        if ((o & 1) == 0) held = done;
                        else {
                            ybuf[o >> 3][(o & 7) >> 1][ph0] = float2v{held.x, done.x};
                            ybuf[o >> 3][(o & 7) >> 1][ph1] = float2v{held.y, done.y};
                        }
                    }
                }
            }
            *reinterpret_cast<unsigned short*>(&cbits[g * kCols + c0]) = (unsigned short)(cb0 | (cb1 << 8));
            *reinterpret_cast<unsigned short*>(&ibits[g * kCols + c0]) = (unsigned short)(ib0 | (ib1 << 8));
        }
        lds_barrier();
        if (z + 1 < z_end) fetch(0, z + 1, 0);                // next channel's first rows: in flight during the rest

        // ================= matrix phase: den (16 rows x 480 columns) of this channel =================
        {
            // wave w owns output tiles 8 w .. 8 w + 7 (< 30); output tile c needs Q tiles c .. c + 3
            const int c_first = 8 * wave, c_last = min(8 * wave + 8, kTxo / 16);
            u32x2 qhi[4], qlo[4];
            auto q_tile = [&](int c, u32x2& hi, u32x2& lo) {
                f32x4 qa = {0.f, 0.f, 0.f, 0.f};
                if (c < kCols / 16) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const unsigned byte = cbits[(4 * s + lg) * kCols + 16 * c + ln];
                        const half8 va = make_half8(lut[byte & 15], lut[byte >> 4]);
                        qa = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, By_hi[s], qa, 0, 0, 0);
                        qa = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, By_lo[s], qa, 0, 0, 0);
                    }
                }
                split4(qa, hi, lo);
            };
            if (c_first < c_last) {
                q_tile(c_first, qhi[0], qlo[0]);
                q_tile(c_first + 1, qhi[1], qlo[1]);
                q_tile(c_first + 2, qhi[2], qlo[2]);
            }
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) {
                const int c = c_first + ci;
                if (c < c_last) {
                    q_tile(c + 3, qhi[(ci + 3) & 3], qlo[(ci + 3) & 3]);
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const half8 bh = make_half8(qhi[(ci + 2 * s) & 3], qhi[(ci + 2 * s + 1) & 3]);
                        const half8 bl = make_half8(qlo[(ci + 2 * s) & 3], qlo[(ci + 2 * s + 1) & 3]);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ax_hi[s], bh, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ax_hi[s], bl, d, 0, 0, 0);
                        d = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ax_lo[s], bh, d, 0, 0, 0);
                    }
                    // lane holds out row ln, out columns 16 c + 4 lg + 0..3
                    *reinterpret_cast<f32x4*>(&den[ln * kDenPitch + 16 * c + 4 * lg]) = d * A.inv_scale;
                }
            }
        }
        lds_barrier();

        // ================= x pass, division, moment sums =================
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int task = t + kThreads * q;
            if (task < 8 * (kTxo / kRun)) {
                const int pr8 = task / (kTxo / kRun), j = task - pr8 * (kTxo / kRun);
                const int oa = 2 * pr8, ob = oa + 1;         // band-relative output rows
                const float2v* row = ybuf[pr8 >> 2][pr8 & 3];
                float2v r[kRun];
#pragma unroll
                for (int k = 0; k < kRun; ++k) r[k] = float2v{0.f, 0.f};
                constexpr int kXB = 4, kNB = (kRun + 2 * H + kXB - 1) / kXB;
                const float2v* src = row + kRun * j + j;     // column c = 8 j + i sits at c + (c >> 3)
                float2v in[2][kXB];
#pragma unroll
                for (int i = 0; i < kXB; ++i) in[0][i] = src[i + (i >> 3)];
#pragma unroll
                for (int b = 0; b < kNB; ++b) {
                    if (b + 1 < kNB) {
#pragma unroll
                        for (int i4 = 0; i4 < kXB; ++i4) {
                            const int i = (b + 1) * kXB + i4;
                            if (i < kRun + 2 * H) in[(b + 1) & 1][i4] = src[i + (i >> 3)];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i4 = 0; i4 < kXB; ++i4) {
                        const int i = b * kXB + i4;
                        if (i < kRun + 2 * H) {
#pragma unroll
                            for (int k = 0; k < kRun; ++k) {
                                const int widx = k + 2 * H - i;
                                if (widx >= 0 && widx <= 2 * H) pk_fma_w(r[k], A.kx, widx, in[b & 1][i4]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x4 da0 = *reinterpret_cast<const f32x4*>(&den[oa * kDenPitch + kRun * j]);
                const f32x4 da1 = *reinterpret_cast<const f32x4*>(&den[oa * kDenPitch + kRun * j + 4]);
                const f32x4 db0 = *reinterpret_cast<const f32x4*>(&den[ob * kDenPitch + kRun * j]);
                const f32x4 db1 = *reinterpret_cast<const f32x4*>(&den[ob * kDenPitch + kRun * j + 4]);
                const float dna[kRun] = {da0.x, da0.y, da0.z, da0.w, da1.x, da1.y, da1.z, da1.w};
                const float dnb[kRun] = {db0.x, db0.y, db0.z, db0.w, db1.x, db1.y, db1.z, db1.w};
                float va[kRun], vb[kRun];
#pragma unroll
                for (int k = 0; k < kRun; ++k) {              // den = 0 (empty window): 0 * inf = NaN, what astropy returns there
                    va[k] = r[k].x * __builtin_amdgcn_rcpf(dna[k]);
                    vb[k] = r[k].y * __builtin_amdgcn_rcpf(dnb[k]);
                }
                const int ya = y0 + oa, yb = y0 + ob;
                const int64_t xo = x0 + kRun * j;
                if (STORE && xo < A.nx) {
                    float* pa = A.out + (int64_t)z * A.out_plane_stride + (int64_t)ya * A.out_row_stride + xo;
                    float* pb = pa + A.out_row_stride;
                    if (xo + kRun <= A.nx && (A.out_row_stride & 3) == 0 && ((uintptr_t)A.out & 15) == 0 && (A.out_plane_stride & 3) == 0) {
                        if (ya < ny) {
                            *reinterpret_cast<f32x4*>(pa) = f32x4{va[0], va[1], va[2], va[3]};
                            *reinterpret_cast<f32x4*>(pa + 4) = f32x4{va[4], va[5], va[6], va[7]};
                        }
                        if (yb < ny) {
                            *reinterpret_cast<f32x4*>(pb) = f32x4{vb[0], vb[1], vb[2], vb[3]};
                            *reinterpret_cast<f32x4*>(pb + 4) = f32x4{vb[4], vb[5], vb[6], vb[7]};
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < kRun; ++k) {
                            if (xo + k < A.nx) {
                                if (ya < ny) pa[k] = va[k];
                                if (yb < ny) pb[k] = vb[k];
                            }
                        }
                    }
                }
                if (MOM) {
                    // include bits of the 16 outputs: rows oa, ob have their centre samples at input rows oa + 14, ob + 14
                    // (one byte group: oa is even), columns 8 j + 14 .. 8 j + 21
                    const int sc = oa + H;
                    const unsigned char* ip = &ibits[(sc >> 3) * kCols + kRun * j + 8];
                    const unsigned long long lo8 = *reinterpret_cast<const unsigned long long*>(ip);
                    const unsigned long long hi8 = *reinterpret_cast<const unsigned long long*>(ip + 8);
                    const unsigned long long w = (lo8 >> 48) | (hi8 << 16);       // byte k = input column 8 j + 14 + k
                    const int bit = sc & 7;
#pragma unroll
                    for (int k = 0; k < kRun; ++k) {
                        const unsigned byte = (unsigned)(w >> (8 * k)) & 0xffu;
                        const bool ca = ((byte >> bit) & 1u) && (va[k] == va[k]);       // nansum: a NaN value is skipped
                        const bool cb = ((byte >> (bit + 1)) & 1u) && (vb[k] == vb[k]);
                        msum[q][k] += float2v{ca ? va[k] : 0.f, cb ? vb[k] : 0.f};
                        mseen[q] |= (ca ? 1u : 0u) << k;
                        mseen[q] |= (cb ? 1u : 0u) << (8 + k);
                    }
                }
            }
        }
        lds_barrier();
    }

    if (MOM) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int task = t + kThreads * q;
            if (task < 8 * (kTxo / kRun)) {
                const int pr8 = task / (kTxo / kRun), j = task - pr8 * (kTxo / kRun);
                const int ya = y0 + 2 * pr8, yb = ya + 1;
                const int64_t xo = x0 + kRun * j;
                float* pp = A.partial + (int64_t)chunk * A.ny * A.nx;
                unsigned char* ps = A.seen + (int64_t)chunk * A.ny * A.nx;
#pragma unroll
                for (int k = 0; k < kRun; ++k) {
                    if (xo + k < A.nx) {
                        if (ya < ny) { pp[(int64_t)ya * A.nx + xo + k] = msum[q][k].x; ps[(int64_t)ya * A.nx + xo + k] = (mseen[q] >> k) & 1u; }
                        if (yb < ny) { pp[(int64_t)yb * A.nx + xo + k] = msum[q][k].y; ps[(int64_t)yb * A.nx + xo + k] = (mseen[q] >> (8 + k)) & 1u; }
                    }
                }
            }
        }
    }
}

// moment 0 = dv * sum over the chunks (float64), NaN where no channel contributed (nansum_allbadtonan,
// dask_spectral_cube.py:54-59)
__global__ __launch_bounds__(256) void spatial_moment_finish_kernel(const float* partial, const unsigned char* seen, int nchunk,
                                                                     int64_t ny, int64_t nx, double dv, double* m0, int64_t m0_row_stride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ny * nx) return;
    double s = 0.0;
    unsigned any = 0;
    for (int c = 0; c < nchunk; ++c) {
        s += (double)partial[(int64_t)c * ny * nx + i];
        any |= seen[(int64_t)c * ny * nx + i];
    }
    const int64_t y = i / nx, x = i - y * nx;
    m0[y * m0_row_stride + x] = any ? dv * s : __longlong_as_double(0x7ff8000000000000LL);
}

inline int chunk_planes(int64_t nz, int64_t ny, int64_t nx) {
    // enough blocks to fill the chip several times over, chunks of at most 64 channels (float32 sums inside a chunk)
    const int64_t tiles = ((ny + kBand - 1) / kBand) * ((nx + kTxo - 1) / kTxo);
    int64_t want_chunks = std::max<int64_t>(1, (4096 + tiles - 1) / tiles);
    int64_t zc = std::max<int64_t>(1, (nz + want_chunks - 1) / want_chunks);
    zc = std::min<int64_t>(zc, 64);
    return (int)zc;
}

}  // namespace

size_t spc_ws_spatial_conv_mfma(int64_t nz, int64_t ny, int64_t nx) {
    const int64_t zc = chunk_planes(nz, ny, nx), nchunk = (nz + zc - 1) / zc;
    return spc_ws_round((size_t)nchunk * ny * nx * sizeof(float)) + spc_ws_round((size_t)nchunk * ny * nx) + 512;
}

extern "C" int spc_spatial_conv_sep_mfma_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                             const double* h_ky, int nky, const double* h_kx, int nkx, float* d_out,
                                             int64_t out_row_stride, int64_t out_plane_stride, double dv, double* d_m0,
                                             int64_t m0_row_stride, void* d_workspace, size_t workspace_bytes) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(h_ky && h_kx && nky > 0 && nkx > 0 && (nky & 1) && (nkx & 1), "kernels must have an odd, positive number of taps");
    SPC_REQUIRE(d_out || d_m0, "nothing to compute: d_out and d_m0 are both NULL");
    MaskDev md{};
    rc = spc_mask_to_dev(mask, cube, &md);
    if (rc) return rc;
#define SPC_UNSUPPORTED(...) do { spc_set_error(__VA_ARGS__); return SPC_ERR_UNSUPPORTED; } while (0)
    if (nky > R || nkx > R) SPC_UNSUPPORTED("spatial_conv_sep_mfma: at most %d taps per axis (got %d x %d)", R, nky, nkx);
    if (md.flags & ~(uint32_t)(SPC_MASK_ARRAY | SPC_MASK_FINITE)) SPC_UNSUPPORTED("spatial_conv_sep_mfma: mask terms other than the array and isfinite");
    if ((cube->nx & 1) || (cube->row_stride & 1) || (cube->plane_stride & 1) || (((uintptr_t)cube->d_data) & 7))
        SPC_UNSUPPORTED("spatial_conv_sep_mfma: nx, the strides and the base must allow 8-byte column pairs");
    if ((md.flags & SPC_MASK_ARRAY) && ((md.row_stride & 1) || (md.plane_stride & 1) || (((uintptr_t)md.arr) & 1)))
        SPC_UNSUPPORTED("spatial_conv_sep_mfma: odd mask strides");
    SmArgs A{};
    double sumy = 0.0, sumx = 0.0;
    for (int i = 0; i < R + 3; ++i) { A.ky[i] = 0.f; A.kx[i] = 0.f; }
    for (int i = 0; i < nky; ++i) { if (!(h_ky[i] >= 0.0)) SPC_UNSUPPORTED("spatial_conv_sep_mfma: negative or NaN tap"); A.ky[(R - nky) / 2 + i] = (float)h_ky[i]; sumy += h_ky[i]; }
    for (int i = 0; i < nkx; ++i) { if (!(h_kx[i] >= 0.0)) SPC_UNSUPPORTED("spatial_conv_sep_mfma: negative or NaN tap"); A.kx[(R - nkx) / 2 + i] = (float)h_kx[i]; sumx += h_kx[i]; }
    if (!(A.ky[H] > 0.f) || !(A.kx[H] > 0.f) || !(sumy > 0.0) || !(sumx > 0.0) || !std::isfinite(sumy) || !std::isfinite(sumx))
        SPC_UNSUPPORTED("spatial_conv_sep_mfma: the centre tap must be positive (astropy's filled-centre rule for empty windows)");
    // fp16 range: the sum of the scaled taps (the largest Q / the largest den contribution) stays below 2^15
    A.sy = (float)std::exp2(std::floor(std::log2(32768.0 / sumy)));
    A.sx = (float)std::exp2(std::floor(std::log2(32768.0 / sumx)));
    A.inv_scale = 1.0f / (A.sy * A.sx);
    SPC_DEVICE(device);
    A.cube = cube->d_data; A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.marr = (md.flags & SPC_MASK_ARRAY) ? md.arr : nullptr;
    A.mrow_stride = md.row_stride; A.mplane_stride = md.plane_stride;
    A.out = d_out;
    A.out_row_stride = out_row_stride ? out_row_stride : cube->nx;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : cube->ny * A.out_row_stride;
    A.lim = (md.flags & SPC_MASK_FINITE) ? 3.402823466e+38f : INFINITY;
    A.nstrips = (int)((cube->nx + kTxo - 1) / kTxo);
    A.nbands = (int)((cube->ny + kBand - 1) / kBand);
    A.zchunk = chunk_planes(cube->nz, cube->ny, cube->nx);
    A.nchunk = (int)((cube->nz + A.zchunk - 1) / A.zchunk);
    const int64_t nblocks = (int64_t)A.nstrips * A.nbands * A.nchunk;
    SPC_REQUIRE(nblocks < (1ll << 31), "too many blocks");
    hipStream_t st = (hipStream_t)stream;
    if (d_m0) {
        SpcWorkspace ws(d_workspace, workspace_bytes);
        SPC_WS_TAKE(d_partial, ws, float, (size_t)A.nchunk * cube->ny * cube->nx);
        SPC_WS_TAKE(d_seen, ws, unsigned char, (size_t)A.nchunk * cube->ny * cube->nx);
        A.partial = d_partial; A.seen = d_seen;
    }
    const bool arr = A.marr != nullptr;
    dim3 grid((unsigned)nblocks), block(kThreads);
#define SPC_SM_LAUNCH(ARR_, STORE_, MOM_) hipLaunchKernelGGL((spatial_sep_mfma_kernel<ARR_, STORE_, MOM_>), grid, block, 0, st, A)
    if (d_out && d_m0) { if (arr) SPC_SM_LAUNCH(true, true, true); else SPC_SM_LAUNCH(false, true, true); }
    else if (d_out) { if (arr) SPC_SM_LAUNCH(true, true, false); else SPC_SM_LAUNCH(false, true, false); }
    else { if (arr) SPC_SM_LAUNCH(true, false, true); else SPC_SM_LAUNCH(false, false, true); }
    SPC_LAUNCH_CHECK();
    if (d_m0) {
        const int64_t n = cube->ny * cube->nx;
        hipLaunchKernelGGL(spatial_moment_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A.partial, A.seen, A.nchunk,
                           cube->ny, cube->nx, dv, d_m0, m0_row_stride ? m0_row_stride : cube->nx);
        SPC_LAUNCH_CHECK();
    }
    return SPC_OK;
}
