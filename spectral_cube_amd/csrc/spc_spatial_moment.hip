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

// ISOSYM: kx == ky and symmetric taps (every Gaussian2DKernel with one stddev): both passes read the SAME scalar weights,
// addressed folded - k[min(j, 2H - j)], 15 distinct values = 8 SGPR pairs.  With two unfolded sets (60 SGPRs of weights)
// the kernel spilled ~700 scalar registers, and every spill is a VALU v_readlane / v_writelane.
// FIN: the mask keeps finite samples only (isfinite): "included by the mask" and "valid for the convolution" are then the
// same bit; without it a NaN under a true mask bit is interpolated over by the convolution but still summed by the moment
template <bool ARR, bool STORE, bool MOM, bool ISOSYM, bool FIN>
__global__ __launch_bounds__(kThreads, 2) void spatial_sep_mfma_kernel(const SmArgs A) {
    auto wj = [](int j) { return ISOSYM ? (j <= H ? j : 2 * H - j) : j; };
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
    // Loads go through one buffer descriptor per plane: lane byte offset in a VGPR, row byte offset in an SGPR that is
    // recomputed where it is used.  With plain pointers the compiler hoisted the 2 x 44 loop-invariant 64-bit row addresses
    // out of the channel loop and spilled ~600 scalar registers (every spill a VALU v_readlane / v_writelane).
    const int voff = (int)(xc * 4), moff = (int)xc;
    const unsigned rbytes = (unsigned)(A.row_stride * 4), mrbytes = (unsigned)A.mrow_stride;
    auto fetch = [&](int slot, int64_t z, int g) {            // rows 8 g .. 8 g + 7 of the band's 44 input rows
        const auto rs = spc_plane_srd(A.cube + z * A.plane_stride);
        const auto rm = spc_plane_srd(ARR ? (const void*)(A.marr + z * A.mplane_stride) : (const void*)A.cube);
        int ybase = y0 - H + 8 * g;
        asm volatile("" : "+s"(ybase));                       // opaque: the row offsets below are not loop invariants
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (8 * g + s < kInRows) {
                const int ic = min(max(ybase + s, 0), ny - 1);
                v[slot][s] = __builtin_bit_cast(float2v, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (int)((unsigned)ic * rbytes), 0));
                if (ARR) mk[slot][s] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rm, moff, (int)((unsigned)ic * mrbytes), 0);
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
            unsigned cb = 0, ib = 0;                          // bits 0-7: column 2t, bits 8-15: column 2t + 1
            int yrow0 = y0 - H + 8 * g;
            asm volatile("" : "+s"(yrow0));                   // (opaque: 44 row-in-bounds masks are not hoisted into SGPRs)
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const int s = 8 * g + s8;
                if (s < kInRows) {
                    const float2v d = v[g & 1][s8];
                    // branch-free: bitwise & on the compare results (&& / || became 88 divergent branches per channel)
                    bool ok0 = __builtin_fabsf(d.x) <= A.lim, ok1 = __builtin_fabsf(d.y) <= A.lim;     // false for NaN
                    bool i0 = true, i1 = true;
                    if (ARR) {
                        const bool m0 = (mk[g & 1][s8] & 0xffu) != 0, m1 = mk[g & 1][s8] > 0xffu;
                        ok0 = ok0 & m0; ok1 = ok1 & m1;
                        i0 = m0; i1 = m1;
                    }
                    if (edge_cols || edge_rows) {            // out of bounds = a valid zero that is not a voxel (block-uniform branch)
                        const bool row_in = (unsigned)(yrow0 + s8) < (unsigned)ny;          // uniform
                        const bool in = col_in & row_in;
                        ok0 = in ? ok0 : true; ok1 = in ? ok1 : true;
                        i0 = i0 & in; i1 = i1 & in;
                        const float2v dz = float2v{in ? d.x : 0.f, in ? d.y : 0.f};
                        const float2v dm_e = float2v{ok0 ? dz.x : 0.f, ok1 ? dz.y : 0.f};
                        cb |= (ok0 ? (1u << s8) : 0u) | (ok1 ? (256u << s8) : 0u);
                        if (!FIN) ib |= (i0 ? (1u << s8) : 0u) | (i1 ? (256u << s8) : 0u);
                        else ib |= ((ok0 & in) ? (1u << s8) : 0u) | ((ok1 & in) ? (256u << s8) : 0u);
#pragma unroll
                        for (int o = 0; o < kBand; ++o) {
                            const int a = s - o;
                            if (a == 0) pk_mul_w(acc[o], A.ky, wj(2 * H), dm_e);
                            else if (a > 0 && a <= 2 * H) pk_fma_w(acc[o], A.ky, wj(2 * H - a), dm_e);
                        }
                    } else {
                        const float2v dm = float2v{ok0 ? d.x : 0.f, ok1 ? d.y : 0.f};
                        cb |= (ok0 ? (1u << s8) : 0u) | (ok1 ? (256u << s8) : 0u);
                        if (!FIN) ib |= (i0 ? (1u << s8) : 0u) | (i1 ? (256u << s8) : 0u);
#pragma unroll
                        for (int o = 0; o < kBand; ++o) {
                            const int a = s - o;              // tap distance: weight ky[2H - a]
                            if (a == 0) pk_mul_w(acc[o], A.ky, wj(2 * H), dm);
                            else if (a > 0 && a <= 2 * H) pk_fma_w(acc[o], A.ky, wj(2 * H - a), dm);
                        }
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
            *reinterpret_cast<unsigned short*>(&cbits[g * kCols + c0]) = (unsigned short)cb;
            // the bits the moment sums over: in an edge block "valid zero" samples are not voxels; under FIN they are the
            // conv-valid bits otherwise
            if (MOM && (!FIN || edge_cols || edge_rows)) *reinterpret_cast<unsigned short*>(&ibits[g * kCols + c0]) = (unsigned short)ib;
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
                                if (widx >= 0 && widx <= 2 * H) pk_fma_w(r[k], ISOSYM ? A.ky : A.kx, wj(widx), in[b & 1][i4]);
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
                    const unsigned char* ip = ((!FIN || edge_cols || edge_rows) ? ibits : cbits) + (sc >> 3) * kCols + kRun * j + 8;
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

// =====================================================================================================================
// Second form (round 4, measured against the first: profiles/r04_masked_spatial_mfma.log): BOTH convolutions on the matrix
// pipe.  The first form keeps the numerator on the vector ALU and, counted by SQ_INSTS_VALU, issues as many vector
// instructions as the ring kernel it was to beat: what the matrix pipe takes off (29 of 58 FMA slots per voxel) the band
// structure puts back (classification of 44 input rows per 16 output rows, the feed of the matrix products, the epilogue).
// Here the vector ALU only classifies, converts and divides:
//
//   numerator    Y^T = DM^T . Ty^T, Out^T = Tx . Y^T   on v_mfma_f32_16x16x4_f32 (float32 in, float32 accumulate: the same
//                arithmetic as the vector FMA chain, at the same rate, on the OTHER pipe)
//   denominator  as above, fp16 hi + lo on v_mfma_f32_16x16x32_f16
//
// A wave owns 32 output rows x 128 output columns and walks a chunk of channels; the waves of a block do not talk to each
// other (no LDS staging, no barriers in the channel loop).  Per channel it visits the 10 input column tiles (16 columns)
// of its region: lane (m = lane & 15, g = lane >> 4) loads column m, rows 4 i + g (i = 0 .. 14) - exactly the A operand
// of the first product; the C layout of every first product (lane: output row n, input columns 4 g + r) is the B layout
// of the second, and numerator and denominator of an output tile end up in the same lane and register (output row n,
// output columns 4 g + r): the division and the moment sums need no exchange either.
struct Sm2Args {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    const uint8_t* marr;                      // uint8 mask array with the cube's strides, or nullptr
    float* out;
    int64_t out_row_stride, out_plane_stride;
    float* partial;
    unsigned char* seen;
    int nstrips, nbands, zchunk, nchunk;
    float lim, sy, sx, scale;                 // scale = sy * sx (den_true = D / scale)
    float ky[R + 3], kx[R + 3];
};

constexpr int k2Rows = 32, k2ColsW = 128, k2Tiles = k2ColsW / 16, k2InTiles = k2Tiles + 2, k2Steps = 15;

template <bool ARR, bool STORE, bool MOM, bool FIN>
__global__ __launch_bounds__(kThreads, 2) void spatial_mfma2_kernel(const Sm2Args A) {
    __shared__ u32x2 lut[16];
    __shared__ float taps[2][32];                             // scaled taps (denominator)
    __shared__ float rawt[2][32];                             // the taps as they are (numerator)
    __shared__ half8 dconst[4 * 2 * 64];                      // fp16 operands of the denominator products: [tyB hi, tyB lo, txA hi, txA lo][step][lane] (8 KB)
    __shared__ f32x4 msum_lds[MOM ? kThreads / 64 * 2 * k2Tiles * 64 : 1];     // (64 KB with the moment)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ln = lane & 15, lg = lane >> 4;
    // (chunk, strip, band): contiguous work items per XCD, the band running fastest
    int chunk, strip, band;
    {
        const int64_t n = (int64_t)gridDim.x, b = blockIdx.x;
        const int64_t q = n / 8, r = n % 8, xcd = b % 8, i = b / 8;
        const int64_t w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
        band = (int)(w % A.nbands);
        strip = (int)((w / A.nbands) % A.nstrips);
        chunk = (int)(w / ((int64_t)A.nbands * A.nstrips));
    }
    const int ny = (int)A.ny, nx = (int)A.nx;
    const int y0 = band * k2Rows;
    const int xw = strip * (4 * k2ColsW) + wave * k2ColsW;    // first output column of this wave
    if (t < 16) {
        const unsigned a = ((t & 1) ? 0x3C00u : 0u) | ((t & 2) ? 0x3C000000u : 0u);
        const unsigned b = ((t & 4) ? 0x3C00u : 0u) | ((t & 8) ? 0x3C000000u : 0u);
        lut[t] = u32x2{a, b};
    }
    if (t < 32) {
        taps[0][t] = t < R ? A.ky[t] * A.sy : 0.f; taps[1][t] = t < R ? A.kx[t] * A.sx : 0.f;
        rawt[0][t] = t < R ? A.ky[t] : 0.f; rawt[1][t] = t < R ? A.kx[t] : 0.f;
    }
    __syncthreads();
    const int z_begin = chunk * A.zchunk, z_end = (int)min((int64_t)z_begin + A.zchunk, A.nz);

    // ---- constant operands of this lane
    // first product B: step i' (0..10), k = lg: input row rho = 4 i' + lg (relative to the row tile), weight ky[n + 28 - rho]
    float tyB[11];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        const int j = ln + 2 * H - (4 * i + lg);
        tyB[i] = (j >= 0 && j <= 2 * H) ? rawt[0][j] : 0.f;
    }
    // second product A: step (tau, r), k = lg: input column kc = 16 tau + 4 lg + r relative to the output tile, weight kx[m + 28 - kc]
    float txA[12];
#pragma unroll
    for (int st = 0; st < 12; ++st) {
        const int j = ln + 2 * H - (16 * (st >> 2) + 4 * lg + (st & 3));
        txA[st] = (j >= 0 && j <= 2 * H) ? rawt[1][j] : 0.f;
    }
    // denominator, first product B (fp16 hi / lo): step s, slot e <-> i' = 8 s + e, rho = 4 i' + lg
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        half8 tyBh, tyBl, txAh, txAl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ip = 8 * s + e;
            const int jy = ln + 2 * H - (4 * ip + lg);
            const float wy = (ip < 11 && jy >= 0 && jy <= 2 * H) ? taps[0][jy] : 0.f;
            const _Float16 hy = (_Float16)wy;
            tyBh[e] = hy; tyBl[e] = (_Float16)(wy - (float)hy);
            // second product A: step 0: slots 0-3 <-> (tau 0, r = e), 4-7 <-> (tau 1, r = e - 4); step 1: slots 0-3 <-> (tau 2, r = e), 4-7 empty
            const int tau = 2 * s + (e >> 2), r = e & 3;
            const int jx = ln + 2 * H - (16 * tau + 4 * lg + r);
            const float wx = (tau < 3 && jx >= 0 && jx <= 2 * H) ? taps[1][jx] : 0.f;
            const _Float16 hx = (_Float16)wx;
            txAh[e] = hx; txAl[e] = (_Float16)(wx - (float)hx);
        }
        if (wave == 0) {                                      // (the operands depend on the lane, not on the wave)
            dconst[(0 * 2 + s) * 64 + lane] = tyBh; dconst[(1 * 2 + s) * 64 + lane] = tyBl;
            dconst[(2 * 2 + s) * 64 + lane] = txAh; dconst[(3 * 2 + s) * 64 + lane] = txAl;
        }
    }

    __syncthreads();
    if (xw >= nx) return;                                     // (no barrier below: a whole wave may leave)
    // ---- addresses: per-lane byte offsets of this lane's 15 input rows at input tile 0 (rows clamped into the plane)
    unsigned rowoff[k2Steps];
    unsigned rowin = 0;                                        // bit i: row 4 i + lg of the region lies inside the plane
    const int col0 = xw - H + ln;                              // this lane's column in input tile 0
#pragma unroll
    for (int i = 0; i < k2Steps; ++i) {
        const int yr = y0 - H + 4 * i + lg;
        rowin |= ((yr >= 0 && yr < ny) ? 1u : 0u) << i;
        rowoff[i] = (unsigned)(min(max(yr, 0), ny - 1) * (int)A.row_stride) * 4u;
    }
    const bool rows_inside = (y0 - H >= 0) && (y0 - H + 60 <= ny);   // uniform
    // epilogue positions: output row y0 + 16 b + ln, columns xw + 16 a + 4 lg + r
    unsigned eoff[2];
    bool erow[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int yo = y0 + 16 * b + ln;
        erow[b] = yo < ny;
        eoff[b] = (unsigned)(min(yo, ny - 1) * (int)A.row_stride + xw + 4 * lg) * 4u;
    }

    // moment sums of the wave's 2 x 8 output tiles: in LDS (16 KB per wave) so that the loop over the input tiles stays a
    // loop - with the sums in registers (64 of them, statically indexed) the tile loop had to be unrolled ten times and
    // the kernel spilled 350 vector and 300 scalar registers
    f32x4* macc = msum_lds + (wave * 2 * k2Tiles) * 64 + lane;
    unsigned mseen[2] = {0u, 0u};
    if (MOM) {
#pragma unroll
        for (int i = 0; i < 2 * k2Tiles; ++i) macc[i * 64] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int z = z_begin; z < z_end; ++z) {
        const auto rs = spc_plane_srd(A.cube + (int64_t)z * A.plane_stride);
        const auto rm = spc_plane_srd(ARR ? (const void*)(A.marr + (int64_t)z * A.plane_stride) : (const void*)A.cube);
        float raw[k2Steps];
        unsigned mkb[k2Steps];
        auto issue_loads = [&](int u) {
            const int cu = col0 + 16 * u;
            const int cc = min(max(cu, 0), nx - 1);            // (clamped: the value is overridden when the column is outside)
            const unsigned cb = (unsigned)cc * 4u;
#pragma unroll
            for (int i = 0; i < k2Steps; ++i) {
                raw[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(rowoff[i] + cb), 0, 0));
                if (ARR) mkb[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rm, (int)((rowoff[i] + cb) >> 2), 0, 0);
            }
        };
        f32x4 Y0[2], Y1[2], Y2[2];                             // rolling window of three input tiles, per row tile
        u32x2 Qh0[2], Qh1[2], Qh2[2], Ql0[2], Ql1[2], Ql2[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            Y0[b] = Y1[b] = Y2[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            Qh0[b] = Qh1[b] = Qh2[b] = Ql0[b] = Ql1[b] = Ql2[b] = u32x2{0u, 0u};
        }
        issue_loads(0);
#pragma unroll 1
        for (int u = 0; u < k2InTiles; ++u) {
            // ---- classify the 15 samples of input tile u
            float dm[k2Steps];
            unsigned vbits = 0;
            const int cu = col0 + 16 * u;
            const bool colin = (cu >= 0) && (cu < nx);
            const bool tile_inside = rows_inside && (xw - H + 16 * u >= 0) && (xw - H + 16 * u + 16 <= nx);   // uniform
            if (tile_inside) {
#pragma unroll
                for (int i = 0; i < k2Steps; ++i) {
                    bool ok = __builtin_fabsf(raw[i]) <= A.lim;
                    if (ARR) ok = ok & (mkb[i] != 0);
                    dm[i] = ok ? raw[i] : 0.f;
                    vbits |= (ok ? 1u : 0u) << i;
                }
            } else {                                           // samples outside the plane are valid zeros
#pragma unroll
                for (int i = 0; i < k2Steps; ++i) {
                    bool ok = __builtin_fabsf(raw[i]) <= A.lim;
                    if (ARR) ok = ok & (mkb[i] != 0);
                    const bool in = colin & (((rowin >> i) & 1u) != 0);
                    ok = in ? ok : true;
                    dm[i] = (ok & in) ? raw[i] : 0.f;
                    vbits |= (ok ? 1u : 0u) << i;
                }
            }
            if (u + 1 < k2InTiles) issue_loads(u + 1);          // in flight during this tile's matrix work
            // ---- first products: numerator Y (float32) and denominator Q (fp16 hi / lo) for both row tiles
#pragma unroll
            for (int b = 0; b < 2; ++b) { Y0[b] = Y1[b]; Y1[b] = Y2[b]; Qh0[b] = Qh1[b]; Qh1[b] = Qh2[b]; Ql0[b] = Ql1[b]; Ql1[b] = Ql2[b]; }
            {
                f32x4 ya = {0.f, 0.f, 0.f, 0.f}, yb = {0.f, 0.f, 0.f, 0.f};      // two independent chains
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    ya = __builtin_amdgcn_mfma_f32_16x16x4f32(dm[i], tyB[i], ya, 0, 0, 0);
                    yb = __builtin_amdgcn_mfma_f32_16x16x4f32(dm[4 + i], tyB[i], yb, 0, 0, 0);
                }
                Y2[0] = ya; Y2[1] = yb;
            }
            {
                // (the two row tiles' chains interleaved: a dependent matrix instruction waits longer than an independent one)
                f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f};
                const unsigned bits0 = vbits & 0x7ffu, bits1 = (vbits >> 4) & 0x7ffu;     // the 11 rows of each row tile
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned byte0 = (bits0 >> (8 * s)) & 0xffu, byte1 = (bits1 >> (8 * s)) & 0xffu;
                    const half8 va0 = make_half8(lut[byte0 & 15], lut[byte0 >> 4]), va1 = make_half8(lut[byte1 & 15], lut[byte1 >> 4]);
                    const half8 bh = dconst[(0 * 2 + s) * 64 + lane], bl = dconst[(1 * 2 + s) * 64 + lane];
                    q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(va0, bh, q0, 0, 0, 0);
                    q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(va1, bh, q1, 0, 0, 0);
                    q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(va0, bl, q0, 0, 0, 0);
                    q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(va1, bl, q1, 0, 0, 0);
                }
                split4(q0, Qh2[0], Ql2[0]);
                split4(q1, Qh2[1], Ql2[1]);
            }
            // ---- second products and the epilogue of output tile a = u - 2
            const int a = u - 2;
            if (a >= 0 && xw + 16 * a < nx) {                  // uniform (nx is a multiple of 16: checked on the host)
                f32x4 oo[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dd[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    oo[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(txA[r], Y0[0][r], oo[0], 0, 0, 0);
                    oo[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(txA[r], Y0[1][r], oo[1], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    oo[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(txA[4 + r], Y1[0][r], oo[0], 0, 0, 0);
                    oo[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(txA[4 + r], Y1[1][r], oo[1], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    oo[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(txA[8 + r], Y2[0][r], oo[0], 0, 0, 0);
                    oo[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(txA[8 + r], Y2[1][r], oo[1], 0, 0, 0);
                }
                {
                    const u32x2 zero2 = {0u, 0u};
                    const half8 ah0 = dconst[(2 * 2 + 0) * 64 + lane], ah1 = dconst[(2 * 2 + 1) * 64 + lane];
                    const half8 al0 = dconst[(3 * 2 + 0) * 64 + lane], al1 = dconst[(3 * 2 + 1) * 64 + lane];
                    const half8 bh00 = make_half8(Qh0[0], Qh1[0]), bl00 = make_half8(Ql0[0], Ql1[0]), bh10 = make_half8(Qh2[0], zero2), bl10 = make_half8(Ql2[0], zero2);
                    const half8 bh01 = make_half8(Qh0[1], Qh1[1]), bl01 = make_half8(Ql0[1], Ql1[1]), bh11 = make_half8(Qh2[1], zero2), bl11 = make_half8(Ql2[1], zero2);
                    dd[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh00, dd[0], 0, 0, 0);
                    dd[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh01, dd[1], 0, 0, 0);
                    dd[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl00, dd[0], 0, 0, 0);
                    dd[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl01, dd[1], 0, 0, 0);
                    dd[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh00, dd[0], 0, 0, 0);
                    dd[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh01, dd[1], 0, 0, 0);
                    dd[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh10, dd[0], 0, 0, 0);
                    dd[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh11, dd[1], 0, 0, 0);
                    dd[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl10, dd[0], 0, 0, 0);
                    dd[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl11, dd[1], 0, 0, 0);
                    dd[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh10, dd[0], 0, 0, 0);
                    dd[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh11, dd[1], 0, 0, 0);
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const f32x4 o = oo[b], d = dd[b];
                    // lane: output row y0 + 16 b + ln, output columns xw + 16 a + 4 lg + 0..3
                    f32x4 val;
                    val.x = o.x * (A.scale * __builtin_amdgcn_rcpf(d.x));     // den = 0 (empty window): 0 * inf = NaN
                    val.y = o.y * (A.scale * __builtin_amdgcn_rcpf(d.y));
                    val.z = o.z * (A.scale * __builtin_amdgcn_rcpf(d.z));
                    val.w = o.w * (A.scale * __builtin_amdgcn_rcpf(d.w));
                    if (STORE && erow[b]) {
                        float* po = A.out + (int64_t)z * A.out_plane_stride + (int64_t)(y0 + 16 * b + ln) * A.out_row_stride + xw + 16 * a + 4 * lg;
                        *reinterpret_cast<f32x4*>(po) = val;
                    }
                    if (MOM) {
                        // the ORIGINAL mask on the output voxels themselves (array term; under isfinite the sample as well)
                        bool i0 = erow[b], i1 = erow[b], i2 = erow[b], i3 = erow[b];
                        if (ARR) {
                            const unsigned m4 = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rm, (int)((eoff[b] >> 2) + 16 * a), 0, 0);
                            i0 = i0 & ((m4 & 0xffu) != 0); i1 = i1 & ((m4 & 0xff00u) != 0);
                            i2 = i2 & ((m4 & 0xff0000u) != 0); i3 = i3 & ((m4 & 0xff000000u) != 0);
                        }
                        if (FIN) {
                            const f32x4 c4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(eoff[b] + 64 * a), 0, 0));
                            i0 = i0 & (__builtin_fabsf(c4.x) <= A.lim); i1 = i1 & (__builtin_fabsf(c4.y) <= A.lim);
                            i2 = i2 & (__builtin_fabsf(c4.z) <= A.lim); i3 = i3 & (__builtin_fabsf(c4.w) <= A.lim);
                        }
                        i0 = i0 & (val.x == val.x); i1 = i1 & (val.y == val.y);      // nansum: a NaN value is skipped
                        i2 = i2 & (val.z == val.z); i3 = i3 & (val.w == val.w);
                        f32x4* slot = macc + (b * k2Tiles + a) * 64;
                        *slot = *slot + f32x4{i0 ? val.x : 0.f, i1 ? val.y : 0.f, i2 ? val.z : 0.f, i3 ? val.w : 0.f};
                        mseen[b] |= ((i0 ? 1u : 0u) | (i1 ? 2u : 0u) | (i2 ? 4u : 0u) | (i3 ? 8u : 0u)) << (4 * a);
                    }
                }
            }
        }
    }

    if (MOM) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (erow[b]) {
                const int64_t base = ((int64_t)chunk * A.ny + (y0 + 16 * b + ln)) * A.nx + xw + 4 * lg;
                for (int a = 0; a < k2Tiles; ++a) {
                    if (xw + 16 * a < nx) {
                        *reinterpret_cast<f32x4*>(A.partial + base + 16 * a) = macc[(b * k2Tiles + a) * 64];
                        const unsigned sb = (mseen[b] >> (4 * a)) & 15u;
                        *reinterpret_cast<unsigned*>(A.seen + base + 16 * a) = (sb & 1u) | ((sb & 2u) << 7) | ((sb & 4u) << 14) | ((sb & 8u) << 21);
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

inline bool use_form2(const spc_cube_f32* cube, const MaskDev& md) {
    const char* env = getenv("SPC_SPATIAL_MFMA_FORM");
    if (env && atoi(env) == 1) return false;
    if ((cube->nx & 15) || (cube->row_stride & 3) || (cube->plane_stride & 3) || (((uintptr_t)cube->d_data) & 15)) return false;
    if ((md.flags & SPC_MASK_ARRAY) && (md.row_stride != cube->row_stride || md.plane_stride != cube->plane_stride || (((uintptr_t)md.arr) & 3))) return false;
    return true;
}

inline int chunk_planes(int64_t nz, int64_t ny, int64_t nx) {
    // enough blocks to fill the chip several times over, chunks of at most 64 channels (float32 sums inside a chunk)
    const int64_t tiles = ((ny + k2Rows - 1) / k2Rows) * ((nx + 4 * k2ColsW - 1) / (4 * k2ColsW));
    int64_t want_chunks = std::max<int64_t>(1, (4096 + tiles - 1) / tiles);
    int64_t zc = std::max<int64_t>(1, (nz + want_chunks - 1) / want_chunks);
    zc = std::min<int64_t>(zc, 64);
    return (int)zc;
}

}  // namespace

// third form (round 5): every product on the fp16 matrix instruction - spc_spatial_split.hip
bool spc_spatial_split_takes(const spc_cube_f32* cube, const MaskDev& md);
size_t spc_ws_spatial_split(int64_t nz, int64_t ny, int64_t nx, int nsum);
constexpr int kSplitTaps = 65;              // (33 entries for kernels of up to 33 taps - three Toeplitz blocks - else 65: five)
int spc_spatial_split_launch(hipStream_t st, const spc_cube_f32* cube, const MaskDev& md, const float* ky, const float* kx, int ntaps,
                             float sy, float sx, float* d_out, int64_t out_row_stride, int64_t out_plane_stride,
                             int nsum, double dv, double m1_add, const double* d_cen, double* d_m0, double* d_m1, double* d_m2,
                             int64_t map_row_stride, void* d_workspace, size_t workspace_bytes);

size_t spc_ws_spatial_conv_mfma(int64_t nz, int64_t ny, int64_t nx, int64_t nsum) {
    const int64_t zc = chunk_planes(nz, ny, nx), nchunk = (nz + zc - 1) / zc;
    const size_t older = spc_ws_round((size_t)nchunk * ny * nx * sizeof(float)) + spc_ws_round((size_t)nchunk * ny * nx) + 512;
    return std::max(older, spc_ws_spatial_split(nz, ny, nx, nsum == 3 ? 3 : 1));
}

static int mfma_entry(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                      const double* h_ky, int nky, const double* h_kx, int nkx, float* d_out,
                      int64_t out_row_stride, int64_t out_plane_stride, double dv, double* d_m0,
                      int64_t m0_row_stride, void* d_workspace, size_t workspace_bytes,
                      const double* d_cen, double m1_add, double* d_m1, double* d_m2, bool split_only = false) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(h_ky && h_kx && nky > 0 && nkx > 0 && (nky & 1) && (nkx & 1), "kernels must have an odd, positive number of taps");
    SPC_REQUIRE(d_out || d_m0 || d_m1 || d_m2, "nothing to compute: d_out and the moment maps are all NULL");
    SPC_REQUIRE(!(d_m1 || d_m2) || d_cen, "moments 1 / 2 need the channel coordinates (d_cen)");
    MaskDev md{};
    rc = spc_mask_to_dev(mask, cube, &md);
    if (rc) return rc;
#define SPC_UNSUPPORTED(...) do { spc_set_error(__VA_ARGS__); return SPC_ERR_UNSUPPORTED; } while (0)
    const int form = [] { const char* e = getenv("SPC_SPATIAL_MFMA_FORM"); return e ? atoi(e) : 3; }();
    const int nsum = (d_m1 || d_m2) ? 3 : (d_m0 ? 1 : 0);
    // ---- the split form (round 5) first: up to 33 taps per axis (three 16-wide Toeplitz blocks cover offsets of -16 .. 16), or up
    // to 65 (five blocks: half the output columns per wave)
    if ((form == 3 || form == 0) && nky <= kSplitTaps && nkx <= kSplitTaps && !(md.flags & ~(uint32_t)(SPC_MASK_ARRAY | SPC_MASK_FINITE)) &&
        spc_spatial_split_takes(cube, md)) {
        const int ntaps = std::max(nky, nkx) <= 33 ? 33 : 65;
        float ky[kSplitTaps] = {}, kx[kSplitTaps] = {};
        double sumy = 0.0, sumx = 0.0;
        bool ok = true;
        for (int i = 0; i < nky; ++i) { ok = ok && (h_ky[i] >= 0.0); ky[(ntaps - nky) / 2 + i] = (float)h_ky[i]; sumy += h_ky[i]; }
        for (int i = 0; i < nkx; ++i) { ok = ok && (h_kx[i] >= 0.0); kx[(ntaps - nkx) / 2 + i] = (float)h_kx[i]; sumx += h_kx[i]; }
        ok = ok && (ky[ntaps / 2] > 0.f) && (kx[ntaps / 2] > 0.f) && (sumy > 0.0) && (sumx > 0.0) && std::isfinite(sumy) && std::isfinite(sumx);
        const int64_t ors = out_row_stride ? out_row_stride : cube->nx, ops = out_plane_stride ? out_plane_stride : cube->ny * ors;
        ok = ok && (!d_out || (((ors | ops) & 3) == 0 && (((uintptr_t)d_out) & 15) == 0 && cube->ny * ors * 4 < 0xfffffff0ll));   // (32-bit byte offsets inside an output plane)
        if (ok) {
            SPC_DEVICE(device);
            // fp16 range: the sum of the scaled taps (the largest x-pass denominator) stays below 2^15
            const float sy = (float)std::exp2(std::floor(std::log2(32768.0 / sumy))), sx = (float)std::exp2(std::floor(std::log2(32768.0 / sumx)));
            return spc_spatial_split_launch((hipStream_t)stream, cube, md, ky, kx, ntaps, sy, sx, d_out, ors, ops, nsum, dv, m1_add, d_cen, d_m0, d_m1, d_m2,
                                            m0_row_stride, d_workspace, workspace_bytes);
        }
    }
    if (split_only) SPC_UNSUPPORTED("spatial_conv_sep_mfma: the split form does not take this call (up to 33 non-negative taps with a positive centre, "
                                    "array / isfinite mask terms, nx and the strides multiples of 4, 16-byte bases)");
    if (nsum == 3) SPC_UNSUPPORTED("spatial_conv_sep_mfma: moments 1 / 2 need the split form (nx and the strides multiples of 4, 16-byte base)");
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
    if (cube->ny * cube->row_stride * 4 >= (1ll << 32) || ((md.flags & SPC_MASK_ARRAY) && cube->ny * md.row_stride >= (1ll << 32)))
        SPC_UNSUPPORTED("spatial_conv_sep_mfma: a plane must stay below 4 GiB (buffer addressing)");
    hipStream_t st = (hipStream_t)stream;
    if (d_m0) {
        SpcWorkspace ws(d_workspace, workspace_bytes);
        SPC_WS_TAKE(d_partial, ws, float, (size_t)A.nchunk * cube->ny * cube->nx);
        SPC_WS_TAKE(d_seen, ws, unsigned char, (size_t)A.nchunk * cube->ny * cube->nx);
        A.partial = d_partial; A.seen = d_seen;
    }
    const bool arr = A.marr != nullptr;
    if (use_form2(cube, md)) {
        Sm2Args B{};
        B.cube = A.cube; B.nz = A.nz; B.ny = A.ny; B.nx = A.nx; B.row_stride = A.row_stride; B.plane_stride = A.plane_stride;
        B.marr = A.marr; B.out = A.out; B.out_row_stride = A.out_row_stride; B.out_plane_stride = A.out_plane_stride;
        B.partial = A.partial; B.seen = A.seen; B.zchunk = A.zchunk; B.nchunk = A.nchunk;
        B.lim = A.lim; B.sy = A.sy; B.sx = A.sx; B.scale = A.sy * A.sx;
        for (int i = 0; i < R + 3; ++i) { B.ky[i] = A.ky[i]; B.kx[i] = A.kx[i]; }
        B.nstrips = (int)((cube->nx + 4 * k2ColsW - 1) / (4 * k2ColsW));
        B.nbands = (int)((cube->ny + k2Rows - 1) / k2Rows);
        const int64_t nb2 = (int64_t)B.nstrips * B.nbands * B.nchunk;
        const bool fin2 = (md.flags & SPC_MASK_FINITE) != 0;
        const bool storev = d_out && ((B.out_row_stride & 3) == 0) && ((B.out_plane_stride & 3) == 0) && ((((uintptr_t)d_out) & 15) == 0);
        if (!d_out || storev) {
            dim3 g2((unsigned)nb2), b2(kThreads);
#define SPC_SM2(ARR_, STORE_, MOM_) do { if (fin2) hipLaunchKernelGGL((spatial_mfma2_kernel<ARR_, STORE_, MOM_, true>), g2, b2, 0, st, B); \
                                         else hipLaunchKernelGGL((spatial_mfma2_kernel<ARR_, STORE_, MOM_, false>), g2, b2, 0, st, B); } while (0)
            if (d_out && d_m0) { if (arr) SPC_SM2(true, true, true); else SPC_SM2(false, true, true); }
            else if (d_out) { if (arr) SPC_SM2(true, true, false); else SPC_SM2(false, true, false); }
            else { if (arr) SPC_SM2(true, false, true); else SPC_SM2(false, false, true); }
            SPC_LAUNCH_CHECK();
            if (d_m0) {
                const int64_t n = cube->ny * cube->nx;
                hipLaunchKernelGGL(spatial_moment_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, B.partial, B.seen, B.nchunk,
                                   cube->ny, cube->nx, dv, d_m0, m0_row_stride ? m0_row_stride : cube->nx);
                SPC_LAUNCH_CHECK();
            }
            return SPC_OK;
        }
    }
    dim3 grid((unsigned)nblocks), block(kThreads);
    bool isosym = true;
    for (int i = 0; i < R; ++i) isosym = isosym && (A.ky[i] == A.kx[i]) && (A.ky[i] == A.ky[R - 1 - i]);
    const bool fin = (md.flags & SPC_MASK_FINITE) != 0;
#define SPC_SM_LAUNCH2(ARR_, STORE_, MOM_, IS_) do { if (fin) hipLaunchKernelGGL((spatial_sep_mfma_kernel<ARR_, STORE_, MOM_, IS_, true>), grid, block, 0, st, A); \
                                                     else hipLaunchKernelGGL((spatial_sep_mfma_kernel<ARR_, STORE_, MOM_, IS_, false>), grid, block, 0, st, A); } while (0)
#define SPC_SM_LAUNCH(ARR_, STORE_, MOM_) do { if (isosym) SPC_SM_LAUNCH2(ARR_, STORE_, MOM_, true); else SPC_SM_LAUNCH2(ARR_, STORE_, MOM_, false); } while (0)
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

extern "C" int spc_spatial_conv_sep_mfma_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                             const double* h_ky, int nky, const double* h_kx, int nkx, float* d_out,
                                             int64_t out_row_stride, int64_t out_plane_stride, double dv, double* d_m0,
                                             int64_t m0_row_stride, void* d_workspace, size_t workspace_bytes) {
    return mfma_entry(device, stream, cube, mask, h_ky, nky, h_kx, nkx, d_out, out_row_stride, out_plane_stride, dv, d_m0, m0_row_stride,
                      d_workspace, workspace_bytes, nullptr, 0.0, nullptr, nullptr);
}

// the same with moments 1 / 2 of the smoothed cube (ABI 6): m1 = S1 / S0 + m1_add, m2 = S2 / S0 - (S1 / S0)^2 with the
// channel coordinates d_cen (nz doubles, device) - spc_moments_f32's conventions
extern "C" int spc_spatial_conv_sep_mfma_moments_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                                     const double* h_ky, int nky, const double* h_kx, int nkx, float* d_out,
                                                     int64_t out_row_stride, int64_t out_plane_stride, const double* d_cen, double dv,
                                                     double m1_add, double* d_m0, double* d_m1, double* d_m2, int64_t map_row_stride,
                                                     void* d_workspace, size_t workspace_bytes) {
    return mfma_entry(device, stream, cube, mask, h_ky, nky, h_kx, nkx, d_out, out_row_stride, out_plane_stride, dv, d_m0, map_row_stride,
                      d_workspace, workspace_bytes, d_cen, m1_add, d_m1, d_m2);
}

// cube -> cube only, and only through the split form: what spc_spatial_conv_sep_f32 tries first for a mask ARRAY (the ring
// kernels carry numerator and denominator through the vector ALU: 57.7 ms at C4 against 50.8).  SPC_ERR_UNSUPPORTED: the caller
// goes on with the ring kernel.
int spc_spatial_conv_split_store(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask, const double* h_ky, int nky,
                                 const double* h_kx, int nkx, float* d_out, int64_t out_row_stride, int64_t out_plane_stride) {
    const char* e = getenv("SPC_SPATIAL_MFMA_FORM");
    if (e && atoi(e) != 3 && atoi(e) != 0) { spc_set_error("split form switched off"); return SPC_ERR_UNSUPPORTED; }
    return mfma_entry(device, stream, cube, mask, h_ky, nky, h_kx, nkx, d_out, out_row_stride, out_plane_stride, 0.0, nullptr, 0, nullptr, 0,
                      nullptr, 0.0, nullptr, nullptr, true);
}
