// Kernel templates of the separable spatial stencil; included by the per-ring
// translation units spc_spatial_conv_r*.hip.
#pragma once
#include "spc_common.h"
#include <algorithm>


namespace spc_spconv {


constexpr int kMaxTaps = 65;
constexpr int kThreads = 256;
constexpr int kRun = 8;                       // x outputs per lane in the x pass
typedef float float2v __attribute__((ext_vector_type(2)));

struct SpArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    float* out;
    int64_t out_row_stride, out_plane_stride;
    int64_t ychunk;                           // output rows per blockIdx.z slice
    int txo;                                  // output columns per strip
    // all-valid fast pass: one byte per (plane, 480-column strip) tile, 0 = done by the fast kernel
    unsigned char* status;
    int fast_nstrips;
    int xcd_swizzle;                          // 1: (strip, plane) from the XCD-aware block order (spc_xcd_order)
    float inv_ksum;                           // 1 / (sum(ky) * sum(kx))
    int centre_zero;                          // the kernel's centre tap is zero: only then can an EMPTY window (den = 0) sit on a
                                              // valid centre sample, which astropy returns; otherwise den = 0 means NaN (= 0 * 1/0)
                                              // and the look-up of the centre sample - global loads in a divergent branch, 59 -> 90 ms
                                              // at C4 with a signal mask whose all-invalid regions are full of empty windows - is skipped
    float pred_lim, pred_lo, pred_hi;         // canonical predicate: |v| <= lim && !(v <= lo) && !(v >= hi)
    alignas(8) float ky[kMaxTaps + 1];        // padded to RY, centred (read pairwise as 64-bit scalars)
    alignas(8) float kx[kMaxTaps + 1];        // padded to RX, centred
};

// Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with its own L2.  With (strip, plane)
// taken straight from blockIdx, the 4 - 5 strips of a plane - which share 28 halo columns with their neighbours and
// sit next to each other in DRAM - run on different XCDs.  This bijective remap (guide: nwg % 8 != 0 needs the
// remainder handled) hands every XCD a contiguous range of work items, i.e. whole planes.
__device__ __forceinline__ void spc_xcd_order(int swizzle, int& strip, int64_t& z) {
    strip = (int)blockIdx.x;
    z = blockIdx.y;
    if (swizzle) {
        const int64_t n = (int64_t)gridDim.x * gridDim.y, b = (int64_t)blockIdx.x + (int64_t)gridDim.x * blockIdx.y;
        const int64_t q = n / 8, r = n % 8, xcd = b % 8, i = b / 8;
        const int64_t w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
        strip = (int)(w % gridDim.x);
        z = w / gridDim.x;
    }
}

// packed FMA / MUL with ONE half of an SGPR pair broadcast to both lanes (see
// spc_spectral_conv_impl.h): R weights live in R SGPRs
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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope
// fence, i.e. s_waitcnt vmcnt(0): every revolution would drain its output stores (and the
// prefetched loads) before the next y pass may start.  Global memory is never shared inside a
// block here, so waiting for the LDS counter is enough.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Output stores.  Measured on MI355X, 512 x 2048^2 (all-valid kernel, same box): the short rings (<= 17 taps)
// are store-bound and like non-temporal stores with the sector-pairing below (9 taps: 4.08 vs 4.31 ms); the
// long rings are bound by the FMA / LDS pipeline and run 2 - 3 % faster with plain stores.
template <bool NT>
__device__ __forceinline__ void st4(f32x4 val, f32x4* ptr) {
    if (NT) __builtin_nontemporal_store(val, ptr);
    else *ptr = val;
}

// A lane of the x pass owns a run of 8 outputs = two 16-byte stores; issued as they are, each
// store instruction writes the first (second) HALF of every 32-byte sector.  For the short,
// memory-bound rings neighbouring lanes first swap one half (DPP quad_perm [1,0,3,2]) so that
// each instruction writes whole sectors: 2.66 -> 2.13 ms (9 taps), 2.44 -> 2.21 ms (17 taps) at
// 1024^3.  The long rings are not memory-bound and keep the 16 extra VALU ops out.
constexpr int kPairedStoreMaxR = 17;
__device__ __forceinline__ float dpp_swap_pair(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
// p = address of this lane's own run; the partner lane (t ^ 1) owns the adjacent run of the same row
__device__ __forceinline__ void store_run8_paired(float* p, int odd, f32x4 lo, f32x4 hi) {
    const f32x4 give = odd ? lo : hi;
    const f32x4 got = f32x4{dpp_swap_pair(give.x), dpp_swap_pair(give.y), dpp_swap_pair(give.z), dpp_swap_pair(give.w)};
    const f32x4 s1 = odd ? got : lo;                      // even: own [0,4) ; odd: partner's [4,8) just below
    const f32x4 s2 = odd ? hi : got;                      // even: partner's [0,4) just above ; odd: own [4,8)
    st4<true>(s1, reinterpret_cast<f32x4*>(p + (odd ? -4 : 0)));
    st4<true>(s2, reinterpret_cast<f32x4*>(p + (odd ? 4 : 8)));
}

constexpr int kFastCols = 512;                // input columns per block of the fast kernel (2 per lane)
constexpr int fast_txo(int R) { return ((kFastCols - 2 * (R / 2)) / 16) * 16; }
// A revolution (R input rows -> R finished y-pass rows) is cut into groups of at most kGroupRows rows;
// only one group's rows (x 2 for the double buffer) ever sit in LDS.
constexpr int kGroupRows = 8;
constexpr int fast_ngroups(int R) { return (R + kGroupRows - 1) / kGroupRows; }

// ---- all-valid fast kernel -----------------------------------------------------------
// Speculative first pass for planes without invalid samples: out = (Gy * Gx * d) / (sum ky *
// sum kx), astropy's NaN-free branch.  One block = one channel x a strip of 480 output
// columns, streaming down the rows.
//   y pass: lane <-> 2 adjacent input columns, register ring of R packed numerators, one
//           v_pk_fma_f32 per tap and column pair; finished rows go to LDS two at a time as
//           float2 (row a, row b) per column - exactly the operand the x pass multiplies;
//   x pass: a task = (row pair, run of 8 columns): the two rows ride in one v_pk_fma_f32 per tap
//           (same scalar weight for both), fed by ONE ds_read_b64 per input column (9-per-8 padded
//           float2 rows: a wave's 32-lane groups touch 32 distinct even banks).
//           14.5 FMA instructions per voxel in each direction instead of 29 + 29.
// Pipeline: the R rows of a revolution are produced in groups of 8 (round 1: 70 KB of LDS for all
// R rows, two barriers per revolution, 2 blocks per CU, VALU 57 % busy).  Group g goes to LDS buffer
// g & 1; ONE barrier later every lane runs its x-pass task over that group while the loads of the next
// group are in flight, then produces the next group into the other buffer: y(g) | barrier | x(g), y(g+1) |
// barrier | ...  A lane that passes barrier g+1 has finished x(g), so buffer g & 1 is free for y(g+2).
// LDS per block: 2 x 4 pair rows x 576 float2 = 36 KB -> 4 blocks (16 waves) per CU, in different
// phases: while one block waits at its barrier or for its loads the others issue FMAs.
// A block that meets an invalid sample marks its tile dirty and quits; the general kernel then
// redoes the dirty tiles.
// ISO: kx == ky (every Gaussian2DKernel with one stddev): both passes read the SAME scalar
// weights - with two weight sets the 2 x 30 SGPRs spill and every spill reload is a VALU
// v_readlane (measured: 755 per revolution next to 1044 FMAs).
// waves per SIMD the register budget is cut for: 4 (128 VGPRs) up to 29 taps, 3 beyond (a 33-slot ring of
// pairs alone is 66 registers)
constexpr int fast_waves(int R) { return R <= 29 ? 4 : (R <= 33 ? 3 : 2); }
template <int R, bool ISO>
__global__ __launch_bounds__(kThreads, fast_waves(R)) void spatial_sep_fast_kernel(const SpArgs A) {
    constexpr int H = R / 2;
    constexpr int kTxoF = fast_txo(R);
    constexpr int NG = fast_ngroups(R);
    // groups requested ahead of the one being consumed.  Two (needs an even group count) was measured too:
    // no gain at 3 waves per SIMD, spills at 4 - the kernel is not waiting for its loads.
    constexpr int PF = 1;
    constexpr int kPairRows = kGroupRows / 2;
    constexpr int kPitch2 = kFastCols + kFastCols / 8;    // float2 per LDS pair row, 9-per-8 padded
    __shared__ float2v ybuf[2][kPairRows][kPitch2];
    __shared__ int dirty;             // set by any lane that meets a non-finite sample (plain LDS word: the
                                      // barrier asm is a compiler memory barrier; `volatile` made it a FLAT sc0 sc1 access)

    const int t = threadIdx.x;
    int strip;
    int64_t z;
    spc_xcd_order(A.xcd_swizzle, strip, z);
    const int64_t x0 = (int64_t)strip * kTxoF;            // first output column of the strip
    const int64_t xin = x0 - H + 2 * t;                   // first of this lane's two input columns
    const bool col_in = (xin >= 0) && (xin + 1 < A.nx);
    // block-uniform: does any lane of this strip look outside the row?
    const bool edge_cols = (x0 - H < 0) || (x0 - H + kFastCols > A.nx);
    const int64_t xc = min(max(xin, (int64_t)0), A.nx - 2);
    const float* p = A.cube + z * A.plane_stride + xc;
    const int ny = (int)A.ny;
    // partial last strip: whole waves beyond the columns the x pass will read retire (a retired
    // wave no longer takes part in the barriers), and the x pass only walks the runs that exist
    const int nrun_eff = ((int)min((int64_t)kTxoF, A.nx - x0) + kRun - 1) / kRun;
    const int ncols = nrun_eff * kRun + 2 * H;
    if (2 * (t & ~63) >= ncols) return;
    if (t == 0) dirty = 0;
    const int nthr = min(kThreads, ((ncols + 127) / 128) * 64);
    const float inv_nrun = 1.0f / (float)nrun_eff;
    // lanes t, t^1 hold adjacent runs of one row pair iff the run count is even; whole runs only
    const bool pairable = ((nrun_eff & 1) == 0) && (x0 + (int64_t)nrun_eff * kRun <= A.nx);
    const int c0 = 2 * t, c1 = 2 * t + 1;
    const int ph0 = c0 + (c0 >> 3), ph1 = c1 + (c1 >> 3);

    float2v num[R];
#pragma unroll
    for (int m = 0; m < R; ++m) num[m] = float2v{0.f, 0.f};

    const int T = ny + 2 * H;
    // the inputs of ONE group live in registers; those of the next group are requested right after the
    // barrier that ends the current group's y pass, i.e. they are in flight during its x pass
    float2v v[PF][kGroupRows];
    auto fetch = [&](int slot, int first_row, int n) {
#pragma unroll
        for (int s = 0; s < kGroupRows; ++s) {
            if (s < n) {
                const int64_t ic = min(max(first_row + s, 0), ny - 1);
                v[slot][s] = __builtin_nontemporal_load(reinterpret_cast<const float2v*>(p + ic * A.row_stride));
            }
        }
    };
    // rows of group gi counted from the first group of the first revolution
    auto group_first = [&](int gi) { return -H + (gi / NG) * R + (gi % NG) * kGroupRows; };
    auto group_rows = [&](int gi) { const int g = gi % NG; return ((g + 1) * kGroupRows < R ? (g + 1) * kGroupRows : R) - g * kGroupRows; };
#pragma unroll
    for (int q = 0; q < PF; ++q) fetch(q, group_first(q), group_rows(q));
    lds_barrier();                                        // dirty = 0 is visible
    float2v chk = float2v{0.f, 0.f};
    for (int t0 = 0; t0 < T; t0 += R) {
        const int i0 = t0 - H;                            // first input row of this revolution
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            constexpr int kG = kGroupRows;
            const int s0 = g * kG, s1 = (g + 1) * kG < R ? (g + 1) * kG : R;     // static after unrolling
            // consecutive groups alternate between the two buffers - across revolutions too, so with an odd
            // group count the choice depends on the revolution (a block-uniform run-time offset)
            float2v (*buf)[kPitch2] = ybuf[(NG % 2 == 0) ? (g & 1) : (((t0 / R) * NG + g) & 1)];
            // Interior groups of interior strips (the common case) carry NO per-sample bookkeeping;
            // rows / columns outside the plane (clamped duplicate loads) become valid zeros only where
            // they can occur, under block-uniform branches.
            if (edge_cols || (i0 + s0 < 0) || (i0 + s1 > ny)) {
#pragma unroll
                for (int s = s0; s < s1; ++s) {
                    const bool in = col_in && (i0 + s >= 0) && (i0 + s < ny);
                    if (!in) v[g % PF][s - s0] = float2v{0.f, 0.f};       // out of bounds = valid zero
                }
            }
            // ---- y pass of the group.  The fast pass only runs for masks where exactly the non-finite
            // samples are invalid (none, or isfinite), and those propagate: chk += 0 * (finished row) is
            // NaN iff a NaN or Inf went into that row - one packed FMA per row instead of compares on
            // every sample (an Inf under "no mask" is merely handed to the general kernel).
            float2v held = float2v{0.f, 0.f};
#pragma unroll
            for (int s = s0; s < s1; ++s) {
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    const int a = (s - m + R) % R;
                    if (a == 0) pk_mul_w(num[m], A.ky, 2 * H - a, v[g % PF][s - s0]);
                    else pk_fma_w(num[m], A.ky, 2 * H - a, v[g % PF][s - s0]);
                }
                const float2v done = num[(s + 1) % R];   // row o = i0 + s - H
                chk = __builtin_elementwise_fma(done, float2v{0.f, 0.f}, chk);
                if (((s - s0) & 1) == 0 && s + 1 < s1) {
                    held = done;                          // first row of a pair: wait for its partner
                } else {
                    const bool single = ((s - s0) & 1) == 0;       // odd group: the last row has no partner
                    const int pr = (s - s0) >> 1;
                    buf[pr][ph0] = single ? float2v{done.x, 0.f} : float2v{held.x, done.x};
                    buf[pr][ph1] = single ? float2v{done.y, 0.f} : float2v{held.y, done.y};
                }
            }
            if (!(chk.x == chk.x) || !(chk.y == chk.y)) dirty = 1;
            lds_barrier();
            if (dirty) {                                  // block-uniform: hand the tile to the general kernel
                if (t == 0) spc_flag_set(A.status + z * A.fast_nstrips + strip);   // (rows already written are redone)
                return;
            }
            // ---- inputs of the group PF ahead, into the registers this group has just freed
            {
                const int gi = (t0 / R) * NG + g + PF;          // (t0 / R: revolutions done; a shift-free division by a constant)
                if (group_first(gi) < ny + H) fetch(g % PF, group_first(gi), group_rows(gi));
            }
            // ---- x pass over the group: task = (row pair, run of kRun output columns)
            const int npairs = (s1 - s0 + 1) / 2;
            for (int task = t; task < npairs * nrun_eff; task += nthr) {
                const int pr = (int)(((float)task + 0.5f) * inv_nrun);   // exact: task < 2^12, nrun <= 62
                const int j = task - pr * nrun_eff;
                const int oa = i0 + s0 + 2 * pr - H, ob = oa + 1;
                const bool wa = (oa >= 0) && (oa < ny), wb = (s0 + 2 * pr + 1 < s1) && (ob >= 0) && (ob < ny);
                if (!wa && !wb) continue;
                const float2v* row = buf[pr];
                float2v r[kRun];
#pragma unroll
                for (int k = 0; k < kRun; ++k) r[k] = float2v{0.f, 0.f};
                // The kRun + 2H input columns come in batches of kXB: batch b + 1 is requested before the
                // FMAs of batch b issue, and scheduling fences keep it that way - left alone the compiler
                // hoists all 36 reads of a task to its top and pays for them with 72 VGPRs (spills at the
                // 128-register budget of four blocks per CU).
                constexpr int kXB = 4, kNB = (kRun + 2 * H + kXB - 1) / kXB;
                const float2v* src = row + kRun * j + j;              // column c = 8 j + i sits at c + (c >> 3)
                float2v in[2][kXB];
#pragma unroll
                for (int q = 0; q < kXB; ++q) in[0][q] = src[q + (q >> 3)];
#pragma unroll
                for (int b = 0; b < kNB; ++b) {
                    if (b + 1 < kNB) {
#pragma unroll
                        for (int q = 0; q < kXB; ++q) {
                            const int i = (b + 1) * kXB + q;
                            if (i < kRun + 2 * H) in[(b + 1) & 1][q] = src[i + (i >> 3)];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < kXB; ++q) {
                        const int i = b * kXB + q;
                        if (i < kRun + 2 * H) {
#pragma unroll
                            for (int k = 0; k < kRun; ++k) {
                                const int widx = k + 2 * H - i;
                                if (widx >= 0 && widx <= 2 * H) pk_fma_w(r[k], ISO ? A.ky : A.kx, widx, in[b & 1][q]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const int64_t xo = x0 + kRun * j;
                float* da = A.out + z * A.out_plane_stride + (int64_t)oa * A.out_row_stride + xo;
                float* db = da + A.out_row_stride;
                if (R <= kPairedStoreMaxR && pairable) {
                    if (wa) store_run8_paired(da, t & 1, f32x4{r[0].x, r[1].x, r[2].x, r[3].x} * A.inv_ksum,
                                              f32x4{r[4].x, r[5].x, r[6].x, r[7].x} * A.inv_ksum);
                    if (wb) store_run8_paired(db, t & 1, f32x4{r[0].y, r[1].y, r[2].y, r[3].y} * A.inv_ksum,
                                              f32x4{r[4].y, r[5].y, r[6].y, r[7].y} * A.inv_ksum);
                } else if (xo + kRun <= A.nx) {
                    if (wa) {
                        st4<(R <= kPairedStoreMaxR)>(f32x4{r[0].x, r[1].x, r[2].x, r[3].x} * A.inv_ksum, reinterpret_cast<f32x4*>(da));
                        st4<(R <= kPairedStoreMaxR)>(f32x4{r[4].x, r[5].x, r[6].x, r[7].x} * A.inv_ksum, reinterpret_cast<f32x4*>(da + 4));
                    }
                    if (wb) {
                        st4<(R <= kPairedStoreMaxR)>(f32x4{r[0].y, r[1].y, r[2].y, r[3].y} * A.inv_ksum, reinterpret_cast<f32x4*>(db));
                        st4<(R <= kPairedStoreMaxR)>(f32x4{r[4].y, r[5].y, r[6].y, r[7].y} * A.inv_ksum, reinterpret_cast<f32x4*>(db + 4));
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < kRun; ++k) {
                        if (xo + k < A.nx) {
                            if (wa) da[k] = r[k].x * A.inv_ksum;
                            if (wb) db[k] = r[k].y * A.inv_ksum;
                        }
                    }
                }
            }
        }
    }
}


__device__ __forceinline__ int lds_phys(int t) { return t + (t >> 3); }   // 9-per-8 padding

// ---- general (masked) kernel -----------------------------------------------------------
// Same march as the fast kernel, but every sample carries (value, validity) packed in one
// float2 so that ONE v_pk_fma_f32 per tap advances numerator and denominator together:
//   y pass: lane <-> 1 input column, register ring of R packed (num, den); finished rows go to
//           LDS as float2;
//   x pass: task = (row, run of 8 columns), packed (num, den) again, then num / den with
//           astropy's empty-window rule (the filled centre sample).
// Interior revolutions of interior strips carry no bounds bookkeeping (block-uniform branch);
// the mask predicate is canonicalised on the host into |v| <= lim, !(v <= lo), !(v >= hi)
// (absent bounds are NaN, which makes the negated compares true) - PRED=false kernels skip the
// two threshold compares.  ISO: kx == ky share their scalar registers.
template <int R, bool ARR, bool PRED>
__device__ __forceinline__ float2v classify(float v, unsigned char mk, float lim, float lo, float hi) {
    bool ok = __builtin_fabsf(v) <= lim;                   // false for NaN (and Inf under isfinite)
    if (PRED) ok = ok && !(v <= lo) && !(v >= hi);
    if (ARR) ok = ok && (mk != 0);
    return ok ? float2v{v, 1.f} : float2v{0.f, 0.f};
}

// ABL (timing-only ablations, built with -DSPC_ABLATE and selected by SPC_SPATIAL_ABLATE; results are wrong by design):
// 1 no output stores | 2 no mask loads | 4 no data loads | 8 no y-pass FMAs | 16 no x-pass FMAs
template <int R, bool ARR, bool PRED, bool ISO, int ABL = 0>
__global__ __launch_bounds__(kThreads, R == 29 ? 3 : 1) void spatial_sep_kernel(const SpArgs A) {
    constexpr int H = R / 2;
    // float2 per LDS row.  Pitch = 28 (mod 32): a 32-lane group of the x pass spans parts of
    // two rows (28 runs per row); with this pitch the second row's runs continue the first
    // row's 18-dword bank progression instead of landing on the same banks (the previous
    // pitch of 290 made every x-pass ds_read_b64 2-way conflicted - SQ_LDS_BANK_CONFLICT
    // was twice SQ_ACTIVE_INST_LDS)
    // 29 taps: the revolution is cut into groups of kGen = 8 rows: y pass of a group (its finished rows parked in LDS),
    // barrier, x pass of the group, barrier; the next revolution's rows are requested one by one as the y pass frees
    // their registers.  20 KB of LDS instead of 73 KB (more than two blocks per CU), 3 waves per SIMD: 1024^3 with a uint8
    // mask 4.73 -> 4.45 ms, with a threshold mask 4.58 -> 4.05 ms.  The other rings keep the whole revolution in one group:
    // 9 / 17 taps fill the rounds of 256 lanes better that way (9 taps 2.36 ms against 2.94 ms grouped), 33 / 65 taps need
    // more registers than 3 waves per SIMD leave (65 taps 12.3 ms against 18.6 ms grouped).
    constexpr bool kGrouped = (R == 29);
    constexpr int kGen = kGrouped ? 8 : R;
    constexpr int kPitch = (R <= 29) ? 316 : 290;       // 33 taps: two blocks per CU only with the short pitch; 65 taps: only fits with it
    static_assert(kPitch >= kThreads + kThreads / 8, "LDS row too short");
    __shared__ float2v yres[kGen * kPitch];

    const int t = threadIdx.x;
    const int64_t z = blockIdx.y;
    const int64_t x0 = (int64_t)blockIdx.x * A.txo;       // first output column of the strip
    if (A.status) {   // both fast tiles this strip overlaps were finished by the all-valid kernel
        const int ft = fast_txo(R);
        const int f0 = (int)(x0 / ft), f1 = (int)(min(x0 + A.txo, A.nx) - 1) / ft;
        if (spc_flag_get(A.status + z * A.fast_nstrips + f0) == 0 && spc_flag_get(A.status + z * A.fast_nstrips + f1) == 0) return;
    }
    constexpr int kTxo = ((kThreads - 2 * H) / kRun) * kRun;    // output columns per strip (== A.txo)
    // partial last strip: see the fast kernel
    const int nrun_eff = ((int)min((int64_t)kTxo, A.nx - x0) + kRun - 1) / kRun;
    const int ncols = nrun_eff * kRun + 2 * H;
    if ((t & ~63) >= ncols) return;
    const int nthr = min(kThreads, (ncols + 63) & ~63);
    const float inv_nrun = 1.0f / (float)nrun_eff;
    const bool pairable = ((nrun_eff & 1) == 0) && (x0 + (int64_t)nrun_eff * kRun <= A.nx) &&
                          (((uintptr_t)(A.out + z * A.out_plane_stride + x0) & 15) == 0) && (A.out_row_stride % 4 == 0);
    const int ny = (int)A.ny;
    const int yb = (int)(blockIdx.z * A.ychunk);
    const int ye = (int)min((int64_t)ny, (int64_t)yb + A.ychunk);
    const int64_t xin = x0 - H + t;                       // input column of this lane (y pass)
    const bool col_in = (xin >= 0) && (xin < A.nx);
    const bool edge_cols = (x0 - H < 0) || (x0 - H + kThreads > A.nx);     // block-uniform
    const int64_t xc = min(max(xin, (int64_t)0), A.nx - 1);
    const float* p = A.cube + z * A.plane_stride + xc;
    const uint8_t* pm = ARR ? A.mask.arr + z * A.mask.plane_stride + xc : nullptr;
    const float lim = A.pred_lim, lo = A.pred_lo, hi = A.pred_hi;
    float2v acc[R];
#pragma unroll
    for (int m = 0; m < R; ++m) acc[m] = float2v{0.f, 0.f};

    float v[R];
    unsigned char mk[R];
    auto load_rows = [&](int i0) {
        if ((i0 < 0) || (i0 + R > ny)) {                  // block-uniform: clamp only at the plane's ends
#pragma unroll
            for (int s = 0; s < R; ++s) {
                const int64_t ic = min(max(i0 + s, 0), ny - 1);
                v[s] = (ABL & 4) ? (float)(t + s) : p[ic * A.row_stride];
                if (ARR) mk[s] = (ABL & 2) ? (unsigned char)1 : pm[ic * A.mask.row_stride];
            }
        } else {
            const float* q = p + (int64_t)i0 * A.row_stride;
            const uint8_t* qm = ARR ? pm + (int64_t)i0 * A.mask.row_stride : nullptr;
#pragma unroll
            for (int s = 0; s < R; ++s) {
                v[s] = (ABL & 4) ? (float)(t + s) : q[(int64_t)s * A.row_stride];
                if (ARR) mk[s] = (ABL & 2) ? (unsigned char)1 : qm[(int64_t)s * A.mask.row_stride];
            }
        }
    };

    // row s of the revolution starting at i0 (the rolling prefetch of the next revolution: v[s] is free once step s has used it)
    auto load_row = [&](int i0, int s) {
        const int64_t ic = min(max(i0 + s, 0), ny - 1);
        v[s] = (ABL & 4) ? (float)(t + s + i0) : p[ic * A.row_stride];
        if (ARR) mk[s] = (ABL & 2) ? (unsigned char)1 : pm[ic * A.mask.row_stride];
    };
    const int T = (ye - yb) + 2 * H;
    load_rows(yb - H);
    for (int t0 = 0; t0 < T; t0 += R) {
        const int i0 = yb - H + t0;                       // first input row of this revolution
        const bool edge = edge_cols || (i0 < 0) || (i0 + R > ny);
        // ---- x pass over `grows` parked rows, the first of which is row slot gs of this revolution
        auto xpass = [&](const int gs, const int grows) {
            // ---- x pass over the rows of this group
            for (int task = t; task < grows * nrun_eff; task += nthr) {
                const int sl = (int)(((float)task + 0.5f) * inv_nrun);    // exact: task < 2^10, nrun <= 31
                const int j = task - sl * nrun_eff;
                const int o = i0 + gs + sl - H;       // output row
                if (o < yb || o >= ye) continue;
                float2v r[kRun];
    #pragma unroll
                for (int k = 0; k < kRun; ++k) r[k] = float2v{0.f, 0.f};
                const float2v* src = yres + sl * kPitch;
    #pragma unroll
                for (int i = 0; i < kRun + 2 * H; ++i) {
                    const float2v in = src[lds_phys(kRun * j + i)];
    #pragma unroll
                    for (int k = 0; k < kRun; ++k) {
                        const int widx = k + 2 * H - i;       // kx index for output k, input i
                        if (ABL & 16) { if (widx == H) r[k] = in; }
                        else if (widx >= 0 && widx <= 2 * H) pk_fma_w(r[k], ISO ? A.ky : A.kx, widx, in);
                    }
                }
                const int64_t xo = x0 + kRun * j;
                float* dst = A.out + z * A.out_plane_stride + (int64_t)o * A.out_row_stride + xo;
                float res[kRun];
    #pragma unroll
                for (int k = 0; k < kRun; ++k) {
                    if (r[k].y != 0.f || !A.centre_zero) {
                        res[k] = r[k].x * __builtin_amdgcn_rcpf(r[k].y);      // (den = 0: 0 * inf = NaN)
                    } else {
                        // empty window -> (filled) centre sample, like astropy
                        res[k] = NAN;
                        if (xo + k < A.nx) {
                            const float c = A.cube[z * A.plane_stride + (int64_t)o * A.row_stride + xo + k];
                            bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, c);
                            if (ARR) inc = inc && A.mask.arr[z * A.mask.plane_stride + (int64_t)o * A.mask.row_stride + xo + k] != 0;
                            if (inc) res[k] = c;
                        }
                    }
                }
                if ((ABL & 1) && !(res[0] == 12345.678f && res[5] == 3.25f && res[7] == res[2])) {
                    // ablation: no stores (the impossible condition keeps the arithmetic alive)
                } else if (R <= kPairedStoreMaxR && pairable) {
                    store_run8_paired(dst, t & 1, f32x4{res[0], res[1], res[2], res[3]}, f32x4{res[4], res[5], res[6], res[7]});
                } else if (xo + kRun <= A.nx && ((((uintptr_t)dst) & 15) == 0)) {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{res[0], res[1], res[2], res[3]};
                    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{res[4], res[5], res[6], res[7]};
                } else {
    #pragma unroll
                    for (int k = 0; k < kRun; ++k)
                        if (xo + k < A.nx) dst[k] = res[k];
                }
            }
        };
        // ---- y pass (a group of rows at a time where the ring is grouped)
#pragma unroll
        for (int s = 0; s < R; ++s) {
            float2v x2 = classify<R, ARR, PRED>(v[s], ARR ? mk[s] : (unsigned char)1, lim, lo, hi);
            if (edge) {                                   // out of bounds = valid zero
                const bool in = col_in && (i0 + s >= 0) && (i0 + s < ny);
                if (!in) x2 = float2v{0.f, 1.f};
            }
#pragma unroll
            for (int m = 0; m < R; ++m) {
                const int a = (s - m + R) % R;
                if (ABL & 8) { if (a == 0) acc[m] = x2; }
                else if (a == 0) pk_mul_w(acc[m], A.ky, 2 * H - a, x2);
                else pk_fma_w(acc[m], A.ky, 2 * H - a, x2);
            }
            // row o = i0 + s - H is complete: park it in the group's LDS slot
            yres[(s % kGen) * kPitch + lds_phys(t)] = acc[(s + 1) % R];
            if (kGrouped && t0 + R < T) load_row(i0 + R, s);   // in flight during the rest of this revolution
            if (kGrouped && (s % kGen == kGen - 1 || s == R - 1)) {
                const int gs = (s / kGen) * kGen;         // first row slot of the group
                lds_barrier();
                xpass(gs, s - gs + 1);
                lds_barrier();
            }
        }
        if (!kGrouped) {                                  // one group: the whole revolution
            lds_barrier();
            if (t0 + R < T) load_rows(i0 + R);            // every row has been used: in flight during the x pass
            xpass(0, R);
            lds_barrier();
        }
    }
}


// ---- general (masked) kernel, grouped pipeline (round 3; dispatched for the 29-tap ring) ----------------------
// The same march as spatial_sep_kernel - packed (num, den) rings along y, finished rows parked in LDS, runs of 8
// along x - rebuilt around what the timing ablations of round 3 showed (profiles/r03_masked_spatial_ablation_*.log,
// 512 x 2048^2 with a uint8 mask: 9.0 ms, of which 6.9 ms remain with NO memory traffic at all and 2.1 ms with no
// FMAs and no memory traffic): the kernel is bound by VALU issue, and what memory adds on top is not latency but the
// instructions that feed it.  Hence:
//   * loads and stores go through ONE buffer descriptor per plane with a SCALAR row offset (no 64-bit address
//     arithmetic per access: ~4 VALU instructions per voxel);
//   * only ONE group of 8 rows is staged ahead (requested right after the barrier that ends a group's y pass, in
//     flight during its x pass) instead of a whole revolution: 16 staging registers instead of 58 -> 128 VGPRs,
//     FOUR blocks (16 waves) per CU, each with 20 KB of LDS;
//   * the x pass divides without looking at the denominator (0 * (1 / 0) is the NaN an empty window needs for a kernel
//     with a non-zero centre tap); the filled-centre rule of astropy for an empty window whose centre sample is valid
//     (kernels with a zero centre tap) is a wave-uniform rare branch.
// BT = threads per block = input columns per strip: 512 (two blocks of 8 waves per CU) wastes half as much as 256 on
// the strip's halo columns in the y pass (28 of 512 instead of 28 of 256) and on idle lanes in the x pass (8 rows x 60
// runs = 480 tasks for 512 lanes instead of 224 for 256).
constexpr int grouped_txo(int R, int BT) { return ((BT - 2 * (R / 2)) / kRun) * kRun; }
constexpr int grouped_pitch(int BT) { return BT == 256 ? 316 : 604; }     // float2 per LDS row, = 28 (mod 32): see spatial_sep_kernel
// SYM (with ISO): the taps are symmetric (every Gaussian) and addressed folded, k[min(j, 2H - j)]: 15 distinct weights
// = 8 SGPR pairs instead of 15 - unfolded the kernel spilled 57 SGPRs (104 v_readlane + 57 v_writelane per revolution).
#ifndef SPC_GROUPED_PF
#define SPC_GROUPED_PF 1          // groups of rows staged ahead (experiment: 2 needs 16 more registers -> 3 blocks per CU)
#endif
template <int R, bool ARR, bool PRED, bool ISO, int BT, bool SYM, int ABL = 0>
__global__ __launch_bounds__(BT, SPC_GROUPED_PF == 1 ? 1024 / BT : 768 / BT) void spatial_sep_grouped_kernel(const SpArgs A) {
    constexpr int H = R / 2;
    static_assert(!SYM || ISO, "folded weights are for kx == ky");
    auto wj = [](int j) { return SYM ? (j <= H ? j : 2 * H - j) : j; };
    constexpr int kG = 8;                                 // rows per group
    constexpr int NG = (R + kG - 1) / kG;
    constexpr int kPitch = grouped_pitch(BT);
    static_assert(kPitch >= BT + BT / 8, "LDS row too short");
    // (one buffer, two barriers per group: with two buffers - one barrier - the block needs 40 KB and only three fit
    // a CU: 9.1 ms against 8.2 ms at 512 x 2048^2)
    __shared__ float2v yres[kG * kPitch];

    const int t = threadIdx.x;
    int strip;
    int64_t z;
    spc_xcd_order(A.xcd_swizzle, strip, z);
    constexpr int kTxo = grouped_txo(R, BT);              // output columns per strip
    const int64_t x0 = (int64_t)strip * kTxo;             // first output column of the strip
    if (A.status) {   // both fast tiles this strip overlaps were finished by the all-valid kernel
        const int ft = fast_txo(R);
        const int f0 = (int)(x0 / ft), f1 = (int)(min(x0 + kTxo, A.nx) - 1) / ft;
        if (spc_flag_get(A.status + z * A.fast_nstrips + f0) == 0 && spc_flag_get(A.status + z * A.fast_nstrips + f1) == 0) return;
    }
    const int nrun_eff = ((int)min((int64_t)kTxo, A.nx - x0) + kRun - 1) / kRun;
    const int ncols = nrun_eff * kRun + 2 * H;
    if ((t & ~63) >= ncols) return;
    const int nthr = min(BT, (ncols + 63) & ~63);
    const float inv_nrun = 1.0f / (float)nrun_eff;
    const int ny = (int)A.ny;
    const int yb = (int)(blockIdx.z * A.ychunk);
    const int ye = (int)min((int64_t)ny, (int64_t)yb + A.ychunk);
    const int64_t xin = x0 - H + t;                       // input column of this lane (y pass)
    const bool col_in = (xin >= 0) && (xin < A.nx);
    const bool edge_cols = (x0 - H < 0) || (x0 - H + BT > A.nx);     // block-uniform
    const int64_t xc = min(max(xin, (int64_t)0), A.nx - 1);
    const float lim = A.pred_lim, lo = A.pred_lo, hi = A.pred_hi;
    // one descriptor per plane; accesses = descriptor + lane byte offset (VGPR) + row byte offset (SGPR)
    const auto rs = spc_plane_srd(A.cube + z * A.plane_stride);
    const auto rm = spc_plane_srd(ARR ? (const void*)(A.mask.arr + z * A.mask.plane_stride) : (const void*)A.cube);
    const auto ro = spc_plane_srd(A.out + z * A.out_plane_stride);
    const int voff = (int)(xc * 4), moff = (int)xc;
    const unsigned rbytes = (unsigned)(A.row_stride * 4), mrbytes = ARR ? (unsigned)A.mask.row_stride : 0u;
    const unsigned orbytes = (unsigned)(A.out_row_stride * 4);
    const bool out_vec = (A.out_row_stride % 4 == 0) && ((((uintptr_t)(A.out + z * A.out_plane_stride)) & 15) == 0) && (x0 % 4 == 0);

    float2v acc[R];
#pragma unroll
    for (int m = 0; m < R; ++m) acc[m] = float2v{0.f, 0.f};
    constexpr int PF = SPC_GROUPED_PF;
    static_assert(PF == 1 || NG % PF == 0, "the staging slot of a group must be static");
    float v[PF][kG];
    unsigned mk[PF][kG];
    const int T = (ye - yb) + 2 * H;
    // rows of group gi counted from the first group of the first revolution
    auto group_first = [&](int gi) { return yb - H + (gi / NG) * R + (gi % NG) * kG; };
    auto group_rows = [&](int gi) { const int g = gi % NG; return ((g + 1) * kG < R ? (g + 1) * kG : R) - g * kG; };
    auto fetch = [&](int slot, int first_row, int n) {
#pragma unroll
        for (int s = 0; s < kG; ++s) {
            if (s < n) {
                const int ic = min(max(first_row + s, 0), ny - 1);                    // uniform
                v[slot][s] = (ABL & 4) ? (float)(t + s) : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (int)((unsigned)ic * rbytes), 0));
                if (ARR) mk[slot][s] = (ABL & 2) ? 1u : (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rm, moff, (int)((unsigned)ic * mrbytes), 0);
            }
        }
    };
#pragma unroll
    for (int q = 0; q < PF; ++q) fetch(q, group_first(q), group_rows(q));
    for (int t0 = 0; t0 < T; t0 += R) {
        const int i0 = yb - H + t0;                       // first input row of this revolution
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int s0 = g * kG, s1 = (g + 1) * kG < R ? (g + 1) * kG : R;          // static after unrolling
            const bool edge = edge_cols || (i0 + s0 < 0) || (i0 + s1 > ny);           // block-uniform
            // ---- y pass of the group
#pragma unroll
            for (int s = s0; s < s1; ++s) {
                float2v x2 = classify<R, ARR, PRED>(v[g % PF][s - s0], ARR ? (unsigned char)mk[g % PF][s - s0] : (unsigned char)1, lim, lo, hi);
                if (edge) {                                   // out of bounds = valid zero
                    const bool in = col_in && (i0 + s >= 0) && (i0 + s < ny);
                    if (!in) x2 = float2v{0.f, 1.f};
                }
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    const int a = (s - m + R) % R;
                    if (ABL & 8) { if (a == 0) acc[m] = x2; }
                    else if (a == 0) pk_mul_w(acc[m], A.ky, wj(2 * H - a), x2);
                    else pk_fma_w(acc[m], A.ky, wj(2 * H - a), x2);
                }
                // row o = i0 + s - H is complete: park it in the group's LDS slot
                yres[(s - s0) * kPitch + lds_phys(t)] = acc[(s + 1) % R];
            }
            lds_barrier();
            // ---- rows of the NEXT group, into the registers this group has just used: in flight during the x pass
            {
                const int gi = (t0 / R) * NG + g + PF;
                if (group_first(gi) < ye + H) fetch(g % PF, group_first(gi), group_rows(gi));
            }
            // ---- x pass over the rows of this group
            const int grows = s1 - s0;
            for (int task = t; task < grows * nrun_eff; task += nthr) {
                const int sl = (int)(((float)task + 0.5f) * inv_nrun);    // exact: task < 2^10, nrun <= 60
                const int j = task - sl * nrun_eff;
                const int o = i0 + s0 + sl - H;           // output row
                if (o < yb || o >= ye) continue;
                float2v r[kRun];
#pragma unroll
                for (int k = 0; k < kRun; ++k) r[k] = float2v{0.f, 0.f};
                const float2v* src = yres + sl * kPitch;
#pragma unroll
                for (int i = 0; i < kRun + 2 * H; ++i) {
                    const float2v in = src[lds_phys(kRun * j + i)];
#pragma unroll
                    for (int k = 0; k < kRun; ++k) {
                        const int widx = k + 2 * H - i;       // kx index for output k, input i
                        if (ABL & 16) { if (widx == H) r[k] = in; }
                        else if (widx >= 0 && widx <= 2 * H) pk_fma_w(r[k], ISO ? A.ky : A.kx, wj(widx), in);
                    }
                }
                const int64_t xo = x0 + kRun * j;
                float res[kRun];
                bool empty = false;
#pragma unroll
                for (int k = 0; k < kRun; ++k) {
                    res[k] = r[k].x * __builtin_amdgcn_rcpf(r[k].y);      // den = 0 (empty window): 0 * inf = NaN
                    empty = empty || (r[k].y == 0.f);
                }
                if (A.centre_zero && __any(empty)) {
                    // astropy returns the (filled) centre sample for an empty window: NaN when that sample is excluded
                    // (always, for a kernel whose centre tap is not zero), the sample itself otherwise
#pragma unroll
                    for (int k = 0; k < kRun; ++k) {
                        if (r[k].y == 0.f) {
                            res[k] = NAN;
                            if (xo + k < A.nx) {
                                const float c = A.cube[z * A.plane_stride + (int64_t)o * A.row_stride + xo + k];
                                bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, c);
                                if (ARR) inc = inc && A.mask.arr[z * A.mask.plane_stride + (int64_t)o * A.mask.row_stride + xo + k] != 0;
                                if (inc) res[k] = c;
                            }
                        }
                    }
                }
                const int ooff = (int)((unsigned)o * orbytes + (unsigned)(xo * 4));   // (< 4 GiB per plane: checked on the host)
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                if ((ABL & 1) && !(res[0] == 12345.678f && res[5] == 3.25f && res[7] == res[2])) {
                    // ablation: no stores (the impossible condition keeps the arithmetic alive)
                } else if (out_vec && xo + kRun <= A.nx) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{res[0], res[1], res[2], res[3]}), ro, ooff, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{res[4], res[5], res[6], res[7]}), ro, ooff + 16, 0, 0);
                } else {
#pragma unroll
                    for (int k = 0; k < kRun; ++k)
                        if (xo + k < A.nx) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res[k]), ro, ooff + 4 * k, 0, 0);
                }
            }
            lds_barrier();
        }
    }
}


inline int pick_ring(int ntaps) {
    const int rings[] = {9, 17, 29, 33, 49, 65};
    for (int r : rings) if (ntaps <= r) return r;
    return 0;
}

inline void pad_taps(float* dst, const double* k, int ntaps, int R) {
    for (int i = 0; i < kMaxTaps; ++i) dst[i] = 0.f;
    const int pad = (R - ntaps) / 2;
    for (int i = 0; i < ntaps; ++i) dst[pad + i] = (float)k[i];
}

inline bool is_sym(const float* k, int R) {
    for (int i = 0; i < R / 2; ++i) if (k[i] != k[R - 1 - i]) return false;
    return true;
}

// the general (ring) kernel's three forms for one mask kind.  Its own function so that the fully unrolled 49- and 65-tap
// rings - minutes of compile time each - can be instantiated in translation units of their own (csrc/spc_spatial_conv_r65m.hip:
// the mask-array forms), which `make -j` builds side by side.
template <int R, bool ARR>
int launch_sep_general(const SpArgs& A, hipStream_t st, dim3 grid, bool iso, bool pred) {
    dim3 block(kThreads);
    if (iso && !pred) hipLaunchKernelGGL((spatial_sep_kernel<R, ARR, false, true>), grid, block, 0, st, A);
    else if (iso) hipLaunchKernelGGL((spatial_sep_kernel<R, ARR, true, true>), grid, block, 0, st, A);
    else hipLaunchKernelGGL((spatial_sep_kernel<R, ARR, true, false>), grid, block, 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}
// the all-valid pass (one kernel: 4 minutes of compile time at 65 taps - spc_spatial_conv_r65f.hip)
template <int R>
int launch_sep_fast(const SpArgs& A, hipStream_t st, bool iso) {
    dim3 fgrid((unsigned)A.fast_nstrips, (unsigned)A.nz, 1), block(kThreads);
    if (iso) hipLaunchKernelGGL((spatial_sep_fast_kernel<R, true>), fgrid, block, 0, st, A);
    else if constexpr (R <= 33) hipLaunchKernelGGL((spatial_sep_fast_kernel<R, false>), fgrid, block, 0, st, A);
    // (49 / 65 taps: the host only asks for this pass when kx == ky - two weight sets do not fit the SGPR file)
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}
#ifdef SPC_SPLIT_MASKED
extern template int launch_sep_general<SPC_SPLIT_MASKED, true>(const SpArgs&, hipStream_t, dim3, bool, bool);
extern template int launch_sep_fast<SPC_SPLIT_MASKED>(const SpArgs&, hipStream_t, bool);
#endif

template <int R>
int launch_sep(const SpArgs& A, hipStream_t st, dim3 grid, bool arr) {
    bool iso = true;
    for (int i = 0; i < R; ++i) iso = iso && (A.ky[i] == A.kx[i]);
    if constexpr (R <= 33 || R == 49 || R == 65) {
        if (A.status) {      // speculative all-valid pass (the general kernel below redoes dirty tiles)
            const int rc = launch_sep_fast<R>(A, st, iso);
            if (rc != SPC_OK) return rc;
        }
    }
    const bool pred = (A.mask.flags & (SPC_MASK_GT | SPC_MASK_GE | SPC_MASK_LT | SPC_MASK_LE)) != 0;
    // non-isotropic kernels only come in the PRED flavour (a superset: absent bounds are NaN)
#ifdef SPC_ABLATE
    if constexpr (R == 29) {
        const char* ab = getenv("SPC_SPATIAL_ABLATE");
        const int abl = ab ? atoi(ab) : 0;
        if (abl && iso && !pred && arr) {
            switch (abl) {
                case 1: hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, true, false, true, 256, true, 1>), dim3((unsigned)((A.nx + grouped_txo(R, 256) - 1) / grouped_txo(R, 256)), grid.y, grid.z), dim3(256), 0, st, A); break;
                case 2: hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, true, false, true, 256, true, 2>), dim3((unsigned)((A.nx + grouped_txo(R, 256) - 1) / grouped_txo(R, 256)), grid.y, grid.z), dim3(256), 0, st, A); break;
                case 3: hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, true, false, true, 256, true, 3>), dim3((unsigned)((A.nx + grouped_txo(R, 256) - 1) / grouped_txo(R, 256)), grid.y, grid.z), dim3(256), 0, st, A); break;
                case 6: hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, true, false, true, 256, true, 6>), dim3((unsigned)((A.nx + grouped_txo(R, 256) - 1) / grouped_txo(R, 256)), grid.y, grid.z), dim3(256), 0, st, A); break;
                case 7: hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, true, false, true, 256, true, 7>), dim3((unsigned)((A.nx + grouped_txo(R, 256) - 1) / grouped_txo(R, 256)), grid.y, grid.z), dim3(256), 0, st, A); break;
                case 8: hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, true, false, true, 256, true, 8>), dim3((unsigned)((A.nx + grouped_txo(R, 256) - 1) / grouped_txo(R, 256)), grid.y, grid.z), dim3(256), 0, st, A); break;
                case 16: hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, true, false, true, 256, true, 16>), dim3((unsigned)((A.nx + grouped_txo(R, 256) - 1) / grouped_txo(R, 256)), grid.y, grid.z), dim3(256), 0, st, A); break;
                case 24: hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, true, false, true, 256, true, 24>), dim3((unsigned)((A.nx + grouped_txo(R, 256) - 1) / grouped_txo(R, 256)), grid.y, grid.z), dim3(256), 0, st, A); break;
                case 31: hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, true, false, true, 256, true, 31>), dim3((unsigned)((A.nx + grouped_txo(R, 256) - 1) / grouped_txo(R, 256)), grid.y, grid.z), dim3(256), 0, st, A); break;
                default: spc_set_error("no such ablation"); return SPC_ERR_UNSUPPORTED;
            }
            SPC_LAUNCH_CHECK();
            return SPC_OK;
        }
    }
#endif
    if constexpr (R == 29) {
        // grouped pipeline; planes of 2 GiB and more keep the round-2 kernel
        const char* gk = getenv("SPC_SPATIAL_GROUPED");
        // 0: round-2 kernel, 256 / 512: threads per block.  Measured at 512 x 2048^2 with a uint8 mask (same box): round-2
        // kernel 9.01 ms, 256 threads 8.29 ms, 512 threads 8.76 ms (less halo and fewer idle x-pass lanes, but barriers
        // over 8 waves and two blocks per CU)
        const int want = gk ? atoi(gk) : 256;
        // (byte offsets inside a plane stay below 2 GiB, like the spectral kernels: larger planes keep 64-bit addressing)
        const bool fits = (uint64_t)A.plane_stride * 4 < (1ull << 31) && (uint64_t)A.out_plane_stride * 4 < (1ull << 31) &&
                          (!arr || (uint64_t)A.mask.plane_stride < (1ull << 31));
        if (want && fits) {
            const int bt = (want == 256 || A.nx < 384) ? 256 : 512;
            const int txo = bt == 256 ? grouped_txo(R, 256) : grouped_txo(R, 512);
            dim3 ggrid((unsigned)((A.nx + txo - 1) / txo), grid.y, grid.z), gblock(bt);
#define SPC_GROUPED(ARRV, PREDV, ISOV, SYMV)                                                                                              \
            do { if (bt == 256) hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, ARRV, PREDV, ISOV, 256, SYMV>), ggrid, gblock, 0, st, A);  \
                 else hipLaunchKernelGGL((spatial_sep_grouped_kernel<R, ARRV, PREDV, ISOV, 512, SYMV>), ggrid, gblock, 0, st, A); } while (0)
            const bool symk = iso && is_sym(A.ky, R);
            if (symk && !pred) { if (arr) SPC_GROUPED(true, false, true, true); else SPC_GROUPED(false, false, true, true); }
            else if (symk) { if (arr) SPC_GROUPED(true, true, true, true); else SPC_GROUPED(false, true, true, true); }
            else if (iso) { if (arr) SPC_GROUPED(true, true, true, false); else SPC_GROUPED(false, true, true, false); }
            else { if (arr) SPC_GROUPED(true, true, false, false); else SPC_GROUPED(false, true, false, false); }
#undef SPC_GROUPED
            SPC_LAUNCH_CHECK();
            return SPC_OK;
        }
    }
    return arr ? launch_sep_general<R, true>(A, st, grid, iso, pred) : launch_sep_general<R, false>(A, st, grid, iso, pred);
}


}  // namespace spc_spconv
