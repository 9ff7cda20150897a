// Third form of the masked separable spatial_smooth (+ moment 0 / 1 / 2 of the smoothed cube), round 5: BOTH convolutions
// on the fp16 matrix instruction (reference: dask_spectral_cube.py:962-993 - the per-channel astropy convolution with
// boundary='fill', nan_treatment='interpolate' - then the nansum of :1083-1104 under the ORIGINAL mask).
//
//   out[z, y, x] = sum_ij ky[i] kx[j] d m / sum_ij ky[i] kx[j] m            (m: validity, 0 / 1; outside the plane: d = 0, m = 1)
//
// The second form (spc_spatial_moment.hip) ran the numerator on v_mfma_f32_16x16x4_f32, which issues at the VECTOR rate
// (32 cycles per 1024 multiply-adds) and fetched x3 the cube (wave-private 32 x 128 regions with a 28-row / 28-column halo,
// loaded four bytes per lane).  Here every product is a v_mfma_f32_16x16x32_f16 (16 cycles per 8192 multiply-adds):
//
//   data     d m  = hi + lo   two fp16 under ONE power-of-two scale per step (the running maximum of the wave's channel:
//                             fp16 needs the range, the products keep 2^-22 of every sample's own magnitude)
//   taps     k    = hi + lo   two fp16 under a power-of-two scale that keeps the far Gaussian tail a normal number
//   x pass   Z    = D Tx      hi.hi + lo.hi + hi.lo (numerator), v.hi + v.lo (denominator; v = the 0 / 1 validity, exact)
//   y pass   O^T  = Z^T Ty^T  Z (float32 accumulators) split hi + lo again: three products each
//
// with banded Toeplitz blocks of the taps as the constant operands (from LDS).  A wave owns 16 NRT output rows x 64
// output columns and walks a chunk of channels alone (no barriers in the channel loop); per channel it marches down its
// NRT + 2 input row tiles of 16 rows x 96 columns - six 16-column UNITS, lane (m, g) = row m, columns 4 g .. 4 g + 3 of
// each unit: one 16-byte load per unit and lane, 64 contiguous bytes per row and instruction.  The x pass leaves
// Z[row 4 g + r][column m] in lane (m, g) - which IS the A operand of the transposed y pass - and the y pass leaves
// out[row m][columns 4 g + r] there, i.e. the lane that loaded exactly those four samples and mask bytes one step
// earlier: the original mask of the moment needs no exchange, and the smoothed cube is stored 16 bytes per lane.
// The y pass runs in scatter form: the new Z tile and the previous one make one K = 32 operand (their order alternates
// with the step, the Toeplitz constants come in both orders), two output row tiles are pending per column tile.
#include "spc_common.h"
#include <algorithm>
#include <cmath>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// NB = number of 16-wide Toeplitz blocks a window spans: 3 (up to 33 taps: offsets of -16 .. 16 columns / rows) or 5 (up to 65
// taps).  Either way a wave loads six 16-column units per step: 4 + 2 halo units for NB = 3 (64 output columns), 2 + 4 for NB = 5
// (32 output columns); the y pass keeps NB - 1 output row tiles pending per column tile.
constexpr int kThreads = 256, kWaves = 4;
constexpr int kUnits = 6;                     // 16-column input units per wave and step
constexpr int kMaxTaps = 65;
template <int NB> struct Geo {
    static constexpr int HB = (NB - 1) / 2;   // halo units (= halo row tiles) on each side
    static constexpr int H = 16 * HB, R = 2 * H + 1;
    static constexpr int CT = kUnits - 2 * HB;           // output column tiles per wave
    static constexpr int OC = 16 * CT;
    static constexpr int NQ = (NB + 1) / 2;   // K = 32 operands (pairs of blocks) per window
    static constexpr int NP = NB - 1;         // pending output row tiles per column tile = period of the step's static pattern
    static constexpr int SETS = 4 * NQ;       // constant operand sets: x pass 2 tile parities x NQ, y pass 2 slot orders x NQ
};

// constant operands in LDS: [set][hi / lo][lane]; a set = the two Toeplitz blocks T_b of the K = 32 operand's halves (or zeros)
// x pass, column tile n of parity par, operand q = unit pair (n - par) / 2 + q (only the pairs (0,1) (2,3) (4,5) are ever formed):
//         blocks (2 q - par, 2 q + 1 - par), zeros where that leaves 0 .. NB - 1          set = par NQ + q
//         NB = 3:  even  [T0; T1] [T2; 0]   odd  [0; T0] [T1; T2]          NB = 5:  even  [T0; T1] [T2; T3] [T4; 0]   odd  [0; T0] [T1; T2] [T3; T4]
// y pass (scatter), operand (previous, new) Z tile, product m: blocks (2 m, 2 m + 1) into the row tile 2 m + 1 steps back, the last
//         one (0, T_{NB-1}) into the row tile NB - 1 steps back, which is then complete; and the same with the halves exchanged
//         for the steps whose operand reads (new, previous)                              set = 2 NQ + order NQ + m
template <int NB>
__device__ __forceinline__ int set_block(int set, int half) {          // Toeplitz block index b or -1 (zeros)
    using G = Geo<NB>;
    if (set < 2 * G::NQ) {
        const int par = set / G::NQ, q = set % G::NQ;
        const int b = 2 * q - par + half;
        return (b >= 0 && b < NB) ? b : -1;
    }
    const int order = (set - 2 * G::NQ) / G::NQ, m = (set - 2 * G::NQ) % G::NQ;
    const bool full = 2 * m + 1 <= NB - 1;
    const int bp = full ? 2 * m : -1, bc = full ? 2 * m + 1 : NB - 1;
    return (half == order) ? bp : bc;                     // order 0: (previous, new); order 1: (new, previous)
}

struct Sm3Args {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    const uint8_t* marr;                      // uint8 mask array or nullptr
    int64_t mrow_stride, mplane_stride;
    float* out;                               // smoothed cube or nullptr
    int64_t out_row_stride, out_plane_stride;
    float* partial;                           // (nsum, nchunk, ny, nx) float32 sums of a chunk, or nullptr
    const double* cen;                        // channel coordinates about the reference (moments 1 / 2): nz doubles, device
    const float* dlt;                         // (three sums) nchunk * zchunk floats: channel coordinate about the chunk's middle channel, 0 beyond nz
    int nstrips, nbands, zchunk, nchunk;
    int nrt;                                  // output row tiles per band (the NRT = 0 instantiations: set by the launch; else = NRT)
    float lim;                                // FLT_MAX under isfinite, +inf otherwise (NaN fails |v| <= lim either way)
    float sy, sx;                             // power-of-two scales of the fp16 taps
    int mirror, sync;                         // odd bands march upwards / one rendezvous of the block's waves per channel
    int wide_mask;                            // the mask rows of a step in two requests per lane + a transpose over lanes (SPC_SPLIT_WIDE_MASK=0: six dword loads)
    float ky[kMaxTaps + 3], kx[kMaxTaps + 3];
};

__device__ __forceinline__ half8 as_half8(u32x4 v) { return __builtin_bit_cast(half8, v); }

// two float32 (times the uniform power-of-two scale s) -> packed fp16 hi and packed fp16 residual.
// hi = fp16(a s) (v_cvt_pk_f16_f32: both halves in one instruction), lo = fp16(a s - hi): the difference is EXACT in float32
// (a s is exact - s is a power of two - and the residual of a rounding to 11 bits fits 24), so v_fma_mix_f32 + one packed
// conversion give the bits v_fma_mixlo / mixhi_f16 gave - those write half a register each and issue at 8.8 cycles where
// v_fma_mix_f32, v_pk_mul_f32 and v_cvt_pk_f16_f32 take ~5 (profiles/r05_micro_mfma_valu_rates.txt): 25 instead of 35 cycles
// per pair (round 6, third session).
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_scaled(float as, float bs, unsigned& hi, unsigned& lo) {
    const half2v h2 = {(_Float16)as, (_Float16)bs};
    const unsigned h = __builtin_bit_cast(unsigned, h2);
    // (in place: the kernel sits at 256 registers, and two more temporaries per pair were answered with scratch)
    asm("v_fma_mix_f32 %0, %0, 1.0, -%1 op_sel_hi:[0,0,1]" : "+v"(as) : "v"(h));
    asm("v_fma_mix_f32 %0, %0, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(bs) : "v"(h));
    const half2v l2 = {(_Float16)as, (_Float16)bs};
    hi = h; lo = __builtin_bit_cast(unsigned, l2);
}
__device__ __forceinline__ void split_pair(float a, float b, float s, unsigned& hi, unsigned& lo) { split_scaled(a * s, b * s, hi, lo); }

// the same for values that come out of a matrix instruction: the FIRST reads of the accumulators are instructions the
// compiler sees (it pads the MFMA -> VALU wait states; it pads nothing in front of inline asm: read too early, an
// accumulator lacks its last products - the lo terms - or holds garbage) - here the multiplications by the scale
__device__ __forceinline__ void split_pair_acc(float a, float b, float s, unsigned& hi, unsigned& lo) { split_scaled(a * s, b * s, hi, lo); }

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));     // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));     // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));     // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));     // row_shr:8: lane 15 of a row holds the row's maximum
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));     // row_bcast:15 into rows 1 and 3
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));     // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// 2^k for a wave-uniform k (scalar arithmetic; below the normal range: 0)
__device__ __forceinline__ float exp2i(int k) { return k < -126 ? 0.f : __builtin_bit_cast(float, (unsigned)(k + 127) << 23); }
__device__ __forceinline__ u32x4 scale_h8(u32x4 v, _Float16 r) { return __builtin_bit_cast(u32x4, __builtin_bit_cast(half8, v) * r); }

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// NSUM: 0 (no moment), 1 (sum of the smoothed values), 3 (+ sum c v, sum c^2 v)
// INC: which output voxels the moment sums over - 0: all (no mask), 1: those valid for the convolution (mask byte and / or
// finite sample: the mask holds isfinite), 2: the mask BYTE alone (a NaN under a true byte is interpolated over and summed)
// NRT: output row tiles per band; 0 (cube -> cube only: no sums in LDS) = A.nrt, a launch-time count - the wave marches down a
// band of ANY height, the whole column if the launch says so: the y halo (2 HB of NRT + 2 HB steps) all but disappears
//
// NSUM = 1: as round 5 - four waves side by side in x, every wave its own sums of a band of NRT = 4 row tiles (4 x 4 KB).
// NSUM = 3 (round 6): the block is EIGHT waves on the SAME 16 NRT rows x 16 CT columns of the map, wave w walking the channels
// z_begin + w, + 8, ... of the chunk, all in the same step at the same time.  A wave leaves the values of a completed row
// tile (the excluded ones as -0.0) in an exchange buffer in LDS; after one barrier per step two / four of the waves add the
// eight contributions, in wave order, to the band's ONE set of sums.  One set per block instead of one per wave: bands of
// 6 row tiles (16 for kernels of 35 - 65 taps) in 152 KB of LDS at one block per CU, where a wave-private set left 1: 8 steps
// per 6 output row tiles instead of 18.  The order of the additions is fixed: the sums are reproducible bit for bit.
// (The same form for ONE sum - bands of 16 row tiles, 18 steps instead of 24 - was built and measured: 40.5 ms at C4 against
//  39.6 ms of the wave-private form.  Without its barrier 36.6 ms: eight waves in the same step at the same time cost what the
//  shorter march wins - profiles/r06_split_shared_sums.txt.)
// Three sums: S0 = sum v, S1' = sum v d, S2' = sum v d^2 with d = the channel coordinate about the CHUNK's middle channel
// (float32: |d| is at most 32 channel widths); the finish kernel shifts the chunks' sums to the map's mean in float64
// (round-5 advisor: sums about the reference channel in float32 lost moment 2 of a narrow line far from it to cancellation).
template <int NB, int NRT, bool ARR, int INC, bool STORE, int NSUM, bool WIDE = false>
__global__ __launch_bounds__(NSUM == 3 ? 512 : kThreads, NSUM == 3 ? 1 : 2) void spatial_split_kernel(const Sm3Args A) {
    static_assert(ARR || !WIDE, "WIDE is about the mask array's loads");
    using G = Geo<NB>;
    constexpr int kCT = G::CT, kOC = G::OC, HB = G::HB, H = G::H, R = G::R, NQ = G::NQ, NP = G::NP, kSets = G::SETS;
    static_assert(NRT > 0 || NSUM == 0, "the moment sums of a band live in LDS: their row tile count is static");
    constexpr bool SH = NSUM == 3;            // channel-parallel waves, shared sums
    constexpr int kW = SH ? 8 : kWaves;       // waves per block
    constexpr int ZSTEP = SH ? kW : 1;        // channels between two channels of one wave
    __shared__ half8 cB[kSets * 2 * 64];                                              // 16 / 24 KB
    __shared__ f32x4 msum[NSUM ? (SH ? 1 : kWaves) * NSUM * NRT * kCT * 64 : 1];      // NSUM x NRT x CT KB per BLOCK (three sums) / per wave
    __shared__ f32x4 xch[SH ? 2 * kW * kCT * 64 : 1];                                 // exchange: 2 buffers x 8 waves x CT KB
    const int nrt = NRT > 0 ? NRT : A.nrt;    // output row tiles of this band
    const int NST = nrt + 2 * HB;             // input row tiles (steps) per channel

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);      // (provably uniform: scalar branches, scalar addresses)
    const int lm = lane & 15, lg = lane >> 4;
    // (chunk, strip, band): blocks are dealt round-robin to the 8 XCDs; every XCD gets a contiguous range of work items with
    // the band running fastest, so the bands that share halo rows of a plane run on one L2 at about the same time
    int chunk, strip, band;
    {
        const int64_t n = (int64_t)gridDim.x, b = blockIdx.x;
        const int64_t q = n / 8, r = n % 8, xcd = b % 8, i = b / 8;
        const int64_t w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
        // (strips fastest instead - the blocks of an XCD side by side in x, sharing 32 of their 96 columns - measured within 2.5 %
        //  of this order for all three forms: not kept)
        band = (int)(w % A.nbands);
        strip = (int)((w / A.nbands) % A.nstrips);
        chunk = (int)(w / ((int64_t)A.nbands * A.nstrips));
    }
    const int ny = (int)A.ny, nx = (int)A.nx;
    const int y0 = band * (16 * nrt);
    // Odd bands march UPWARDS (their rows are taken in mirrored order, the y taps reversed): a band's last two input row
    // tiles are the first two of the band below it, and with both marching down the two reads of those 32 rows lie four
    // steps apart - longer than the L2 keeps them (fetch x2.1 of the cube measured).  Mirrored, neighbouring bands - which
    // run at the same time on one XCD - touch their shared rows in the same step.
    const bool mir = !SH && A.mirror && (band & 1);       // (not with sums: bands of 16 / 6 row tiles share 2 of 18 / 8 steps' rows, and the code path costs registers)
    const int ymir = y0 + 16 * nrt - 1;        // local row q of a mirrored band is the plane's row ymir - q
    auto plane_row = [&](int q) { return mir ? ymir - q : y0 + q; };
    const int xw = SH ? strip * kOC : strip * (kWaves * kOC) + wave * kOC;       // first output column of this wave

    // ---- constant operands: the waves share the (set, hi / lo) pairs, every thread builds them for its lane
    static_assert((2 * kSets) % kW == 0, "the waves share the constant operands evenly");
    for (int q = wave * (2 * kSets / kW); q < (wave + 1) * (2 * kSets / kW); ++q) {
        const int set = q >> 1, lo = q & 1;
        const bool isx = set < 2 * NQ;
        half8 op;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int b = set_block<NB>(set, e >> 2);
            const int idx = 2 * H + lm - 16 * b - (4 * lg + (e & 3));       // tap index of (output lm, input 16 (b - HB) + 4 lg + (e & 3))
            float w = 0.f;
            if (b >= 0 && idx >= 0 && idx < R) w = isx ? A.kx[idx] * A.sx : A.ky[mir ? R - 1 - idx : idx] * A.sy;
            const _Float16 h = (_Float16)w;
            op[e] = lo ? (_Float16)(w - (float)h) : h;
        }
        cB[q * 64 + lane] = op;
    }
    if (NSUM) {
        for (int i = t; i < (SH ? 1 : kWaves) * NSUM * NRT * kCT * 64; i += 64 * kW) msum[i] = f32x4{-0.f, -0.f, -0.f, -0.f};       // -0.0: "nothing added yet" (see the epilogue)
    }
    __syncthreads();
    if (xw >= nx) return;                                      // (NSUM = 0: no barrier below, a whole wave may leave; NSUM > 0: the block leaves)
    const int z_begin = chunk * A.zchunk, z_end = (int)min((int64_t)z_begin + A.zchunk, A.nz);
    if (z_begin >= z_end) return;

    // ---- addressing: per-lane column byte offsets of the six units (clamped into the row), the row of step j is scalar + lane
    const unsigned rbytes = (unsigned)(A.row_stride * 4), mrbytes = (unsigned)A.mrow_stride;
    unsigned coff[kUnits];
    unsigned colin = 0;                                        // bit u: this lane's four columns of unit u lie inside the plane
#pragma unroll
    for (int u = 0; u < kUnits; ++u) {
        const int c = xw - 16 * HB + 16 * u + 4 * lg;
        colin |= ((c >= 0 && c + 3 < nx) ? 1u : 0u) << u;
        coff[u] = (unsigned)min(max(c, 0), nx - 4);
    }
    const bool cols_inside = (xw - 16 * HB >= 0) && (xw + kOC + 16 * HB <= nx);      // uniform
    // (WIDE) byte columns of this lane's two mask requests: unit lg whole, and 8 bytes of unit 4 + (lg & 1) (half lg >> 1)
    const unsigned wcolA = (unsigned)min(max(xw - 16 * HB + 16 * lg, 0), nx - 16);
    const unsigned wcolB = (unsigned)min(max(xw - 16 * HB + 16 * (4 + (lg & 1)), 0), nx - 16) + 8u * (unsigned)(lg >> 1);

    f32x4 raw[kUnits];
    unsigned mk[kUnits];
    auto issue_loads = [&](int z, int j) {
        const auto rs = spc_plane_srd(A.cube + (int64_t)z * A.plane_stride);
        const auto rm = spc_plane_srd(ARR ? (const void*)(A.marr + (int64_t)z * A.mplane_stride) : (const void*)A.cube);
        const int row = min(max(plane_row(-16 * HB + 16 * j + lm), 0), ny - 1);
        const unsigned ro = (unsigned)row * rbytes, mo = (unsigned)row * mrbytes;
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            raw[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ro + coff[u] * 4u), 0, 0));
            if (ARR && !WIDE) mk[u] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rm, (int)(mo + coff[u]), 0, 0);
        }
        if (ARR && WIDE) {
            // The 96 mask bytes of a row in TWO requests per lane instead of six (round 6, third session): a mask dword per unit and
            // lane is 16 rows x 16 bytes per instruction - six instructions touch 96 half lines for 1.5 KB, and with the mask loads
            // taken out the three forms ran 11 - 23 % faster, nearly what the four times larger data loads cost
            // (profiles/r06_split_mask_loads.txt).  Lane (m, g) asks for the 16 bytes of unit g of its row and for 8 bytes of
            // units 4 / 5 (piece = g with its two bits exchanged); mask_transpose() below hands every lane ITS dword of every unit.
            // WIDE is launched for nx % 16 == 0: a unit then lies inside the plane or outside it as a whole - a unit outside is
            // read from a clamped place and replaced by the step's fix-up like before.
            const unsigned q = (((unsigned)lg & 1u) << 1) | ((unsigned)lg >> 1);
            const u32x4 a = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, (int)(mo + wcolA), 0, 0));
            const u32x2 b = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rm, (int)(mo + wcolB), 0, 0));
            (void)q;
            mk[0] = a.x; mk[1] = a.y; mk[2] = a.z; mk[3] = a.w; mk[4] = b.x; mk[5] = b.y;
        }
    };
    // lane (m, g) holds the four dwords of unit g (mk[0..3]) and two of units 4 / 5: a 4 x 4 transpose over the lanes m, m + 16,
    // m + 32, m + 48 - two v_permlane32_swap, two v_permlane16_swap - and one more swap leave mk[u] = its own four bytes of unit u
    auto mask_transpose = [&]() {
        auto s32 = [](unsigned& x, unsigned& y) { const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); x = r[0]; y = r[1]; };
        auto s16 = [](unsigned& x, unsigned& y) { const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false); x = r[0]; y = r[1]; };
        s32(mk[0], mk[2]); s32(mk[1], mk[3]);
        s16(mk[0], mk[1]); s16(mk[2], mk[3]);
        s16(mk[4], mk[5]);
    };

    // ---- state
    u32x4 za[kCT][4];                         // y-pass A operands per column tile: numerator hi, lo, denominator hi, lo; (.x,.y) = slot 0, (.z,.w) = slot 1
    f32x4 Pn[NP][kCT], Pd[NP][kCT];           // pending output row tiles (numerator, denominator), slot = row tile mod NP
    unsigned incsave[NP][kCT];                // one byte per output column: "included by the ORIGINAL mask", for the centre units of a step (slot = step mod NP)
#pragma unroll
    for (int n = 0; n < kCT; ++n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) za[n][q] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < NP; ++q) { Pn[q][n] = Pd[q][n] = f32x4{0.f, 0.f, 0.f, 0.f}; incsave[q][n] = 0u; }
    }
    f32x4* const macc = msum + (SH ? (size_t)0 : (size_t)wave * NSUM * NRT * kCT * 64) + lane;     // (one sum) this wave's sums
    const bool tile_cols_inside = cols_inside;
    int E = -128;                              // 2^E bounds every sample of the channel seen so far (the pending numerators carry 2^-E)
    int zcur = z_begin, zround_next = z_begin;
    bool alive = true;                         // (NSUM > 0) this wave has a channel in the current round of eight
    int zround = z_begin;                      // first channel of the current round of eight

    // the block's sums: after ONE barrier the contributions of the eight waves to the row tile completed in step jp are added, in
    // wave order, by the waves of one group (CT waves, a different group every step).  Writes and additions alternate in every
    // wave; the buffer is the parity of the row tile - of the step, a compile-time constant of every copy of the step's code
    // (NRT is even: the parity starts over with every channel): a buffer is written again two writes later, behind the barrier
    // of the write in between - which every wave reaches after ITS reads of this one
    static_assert(!SH || (NRT % 2 == 0 && (NB - 1) % 2 == 0 && NP % 2 == 0), "the exchange buffer of a row tile is the parity of its step");
    auto reduce_tile = [&](auto buf, const int jp) {
        constexpr int bw = decltype(buf)::value;
        // (three sums) the eight channel coordinates of the round: scalar loads, in flight across the barrier
        float dls[kW];
#pragma unroll
        for (int w = 0; w < kW; ++w) dls[w] = NSUM == 3 ? A.dlt[zround + w] : 0.f;
        // (LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier; __syncthreads() would also drain vmcnt - the loads of the next
        //  step or channel that are in flight, and the stores of the smoothed cube)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int i = jp - (NB - 1);
        constexpr int kGroups = kW / kCT;
        if ((wave / kCT) == (i % kGroups)) {        // uniform
            const int n = wave % kCT;
            f32x4* const sums = msum + (size_t)((i * kCT + n) * NSUM) * 64 + lane;
            const f32x4* const src = xch + (size_t)(bw * kW * kCT + n) * 64 + lane;
            f32x4 s0 = sums[0], s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (NSUM == 3) { s1 = sums[64]; s2 = sums[128]; }
#pragma unroll
            for (int w = 0; w < kW; w += 2) {            // (two contributions in flight: all eight at once spill)
                const f32x4 a = src[(size_t)w * kCT * 64], b = src[(size_t)(w + 1) * kCT * 64];
                s0 = s0 + a;
                s0 = s0 + b;
                if (NSUM == 3) {
                    const float da = dls[w], db = dls[w + 1];
                    const f32x4 ad = a * da, bd = b * db;
                    s1 = s1 + ad; s2 = s2 + ad * da;
                    s1 = s1 + bd; s2 = s2 + bd * db;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            sums[0] = s0;
            if (NSUM == 3) { sums[64] = s1; sums[128] = s2; }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // one step = one input row tile (16 rows x 96 columns) of the channel; PAR = parity of the step (static: it names the
    // operand slot the new Z tile goes to and the slots of the pending row tiles)
    auto step = [&](auto par, const int j) {
        constexpr int PARN = decltype(par)::value;           // j mod NP
        constexpr int PAR = PARN & 1;                         // slot of the new Z tile in the y-pass operands
        const int z = zcur;
        // ================= classification of the step's samples: d = valid ? sample : 0, validity as fp16 0 / 1
        const int rowj = plane_row(-16 * HB + 16 * j + lm);
        const bool row_in = (rowj >= 0) && (rowj < ny);
        const int ra = plane_row(-16 * HB + 16 * j), rb = plane_row(-16 * HB + 16 * j + 15);
        const bool tile_inside = tile_cols_inside && (min(ra, rb) >= 0) && (max(ra, rb) < ny);            // uniform
        float d[kUnits][4];
        u32x2 vh[kUnits];
        float mx = 0.f;
        if (WIDE) mask_transpose();
        if (!tile_inside) {                      // samples outside the plane are VALID ZEROS (boundary='fill', fill_value=0): in place,
#pragma unroll                                   // one uniform branch (per unit inside the loop below it cost the interior five moves per unit)
            for (int u = 0; u < kUnits; ++u) {
                const bool in = row_in & (((colin >> u) & 1u) != 0);
                raw[u].x = in ? raw[u].x : 0.f; raw[u].y = in ? raw[u].y : 0.f; raw[u].z = in ? raw[u].z : 0.f; raw[u].w = in ? raw[u].w : 0.f;
                if (ARR) mk[u] = in ? mk[u] : 0x01010101u;
            }
        }
        // Two classifications.  FAST (a mask array whose bytes are all 0 / 1 in this step, no NaN among the included samples,
        // and - where the mask holds isfinite - no infinity): the validity of a sample is bit 0 of its mask byte, sign-extended
        // to a word (v_bfe_i32), ANDed onto the sample and onto the fp16 1.0 pattern - 4 vector instructions per sample and no
        // lane masks, against 5.6 and two lane masks per sample of the general form below (compare, byte test, two selects).
        // Whether it applies is decided per wave and step: the OR of the six mask dwords, one v_cmp_u per PAIR of masked
        // samples, and the step's maximum (an infinity shows there).
        bool use_fast = false;
        if (ARR) {
            const unsigned mor = mk[0] | mk[1] | mk[2] | mk[3] | mk[4] | mk[5];
            use_fast = !__any((mor & 0xfefefefeu) != 0u);
        }
        unsigned mb = 0u;
        {
            if (use_fast) {
                unsigned long long nanmask = 0ull;
#pragma unroll
                for (int u = 0; u < kUnits; ++u) {
                    const f32x4 r = raw[u];
                    const unsigned m = mk[u];
                    const unsigned w0 = (unsigned)__builtin_amdgcn_sbfe((int)m, 0, 1), w1 = (unsigned)__builtin_amdgcn_sbfe((int)m, 8, 1);
                    const unsigned w2 = (unsigned)__builtin_amdgcn_sbfe((int)m, 16, 1), w3 = (unsigned)__builtin_amdgcn_sbfe((int)m, 24, 1);
                    // (copies first: __builtin_bit_cast of a vector ELEMENT reads element 0 whatever the element - clang 19 / ROCm 7.2)
                    const float rx = r.x, ry = r.y, rz = r.z, rw = r.w;
                    d[u][0] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rx) & w0);
                    d[u][1] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, ry) & w1);
                    d[u][2] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rz) & w2);
                    d[u][3] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, rw) & w3);
                    vh[u].x = (w0 & 0x3C00u) | (w1 & 0x3C000000u);
                    vh[u].y = (w2 & 0x3C00u) | (w3 & 0x3C000000u);
                    nanmask |= __builtin_amdgcn_ballot_w64(__builtin_isunordered(d[u][0], d[u][1])) | __builtin_amdgcn_ballot_w64(__builtin_isunordered(d[u][2], d[u][3]));
                    mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(d[u][0])), __builtin_fabsf(d[u][1]));
                    mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(d[u][2])), __builtin_fabsf(d[u][3]));
                }
                __builtin_amdgcn_sched_barrier(0);
                mb = wave_max_u32(__builtin_bit_cast(unsigned, mx));
                // a NaN under a true byte, or an infinity the mask's isfinite term must reject: the general form decides
                if (nanmask != 0ull || (INC == 1 && mb >= 0x7f800000u)) use_fast = false;
            }
            if (!use_fast) {
                mx = 0.f;
#pragma unroll
                for (int u = 0; u < kUnits; ++u) {
                    const f32x4 r = raw[u];
                    const unsigned m = ARR ? mk[u] : 0x01010101u;
                    bool o0 = __builtin_fabsf(r.x) <= A.lim, o1 = __builtin_fabsf(r.y) <= A.lim;
                    bool o2 = __builtin_fabsf(r.z) <= A.lim, o3 = __builtin_fabsf(r.w) <= A.lim;
                    if (ARR) { o0 = o0 & ((m & 0xffu) != 0); o1 = o1 & ((m & 0xff00u) != 0); o2 = o2 & ((m & 0xff0000u) != 0); o3 = o3 & ((m & 0xff000000u) != 0); }
                    d[u][0] = o0 ? r.x : 0.f; d[u][1] = o1 ? r.y : 0.f; d[u][2] = o2 ? r.z : 0.f; d[u][3] = o3 ? r.w : 0.f;
                    vh[u].x = (o0 ? 0x3C00u : 0u) | (o1 ? 0x3C000000u : 0u);
                    vh[u].y = (o2 ? 0x3C00u : 0u) | (o3 ? 0x3C000000u : 0u);
                    mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(d[u][0])), __builtin_fabsf(d[u][1]));       // (v_max3_f32 with |.| modifiers)
                    mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(d[u][2])), __builtin_fabsf(d[u][3]));
                }
                __builtin_amdgcn_sched_barrier(0);       // (the validity words are built HERE: sunk below the scale they keep 24 lane masks in SGPRs, and those spill)
                mb = wave_max_u32(__builtin_bit_cast(unsigned, mx));
            }
        }
        // what the moment's include test needs HB steps later: unit u = n + HB holds output column tile n of row tile j - HB
        if (NSUM) {
#pragma unroll
            for (int u = HB; u < HB + kCT; ++u) {
                if (INC == 1) incsave[PARN][u - HB] = __builtin_amdgcn_perm(vh[u].y, vh[u].x, 0x07050301u);   // byte 1 of every fp16: 0x3C / 0
                else if (INC == 2) incsave[PARN][u - HB] = mk[u];   // the array term alone: a NaN under a true byte is interpolated over AND summed
            }
        }
        // ================= the step's scale
        int e = (int)(mb >> 23) - 126;                                                 // max < 2^e
        e = min(max(e, -100), 127);
        if (e > E) {
            if (j > 0) {
                const float r = exp2i(E - e);
#pragma unroll
                for (int n = 0; n < kCT; ++n) {
#pragma unroll
                    for (int q = 0; q < NP; ++q) Pn[q][n] = Pn[q][n] * r;
                    // (the previous Z tile in the y-pass operand carries the old scale too; fp16 times a power of two)
                    za[n][0] = scale_h8(za[n][0], (_Float16)r); za[n][1] = scale_h8(za[n][1], (_Float16)r);
                }
            }
            E = e;
        }
        const float s = exp2i(15 - e);                                                 // samples -> below 2^15
        const float fz = exp2i(e - E - 15);                                            // x-pass numerators -> below 2^15, common scale 2^-E
        // ================= fp16 operands of the x pass: three unit pairs, hi / lo / validity
        u32x4 hiP[3], loP[3], vhP[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            unsigned h[4], l[4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int u = 2 * p + q;
                split_pair(d[u][0], d[u][1], s, h[2 * q], l[2 * q]);
                split_pair(d[u][2], d[u][3], s, h[2 * q + 1], l[2 * q + 1]);
            }
            hiP[p] = u32x4{h[0], h[1], h[2], h[3]};
            loP[p] = u32x4{l[0], l[1], l[2], l[3]};
            vhP[p] = u32x4{vh[2 * p].x, vh[2 * p].y, vh[2 * p + 1].x, vh[2 * p + 1].y};
        }
        __builtin_amdgcn_sched_barrier(0);
        // ================= x pass and split of its result, per column tile
        // (column tiles in the order even ones, odd ones: tiles of one parity share their constant operands - read from LDS once
        //  per parity: read per tile, every tile's first matrix instruction waited out an LDS round trip)
        half8 ch[NQ], cl[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { ch[q] = half8{}; cl[q] = half8{}; }
#pragma unroll
        for (int nn = 0; nn < kCT; ++nn) {
            constexpr int kEven = (kCT + 1) / 2;
            const int n = nn < kEven ? 2 * nn : 2 * (nn - kEven) + 1;          // 0, 2, 1, 3  /  0, 1
            const int par = n & 1, p0 = (n - par) / 2;
            if (nn == 0 || nn == kEven) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) { ch[q] = cB[((par * NQ + q) * 2 + 0) * 64 + lane]; cl[q] = cB[((par * NQ + q) * 2 + 1) * 64 + lane]; }
            }
            f32x4 zn = {0.f, 0.f, 0.f, 0.f}, zd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int p = p0 + q;                      // unit pair (2 p, 2 p + 1)
                zn = MFMA(as_half8(hiP[p]), ch[q], zn);
                zd = MFMA(as_half8(vhP[p]), ch[q], zd);
                zn = MFMA(as_half8(loP[p]), ch[q], zn);
                zd = MFMA(as_half8(vhP[p]), cl[q], zd);
                zn = MFMA(as_half8(hiP[p]), cl[q], zn);
            }
            // lane (m = output column, g): zn / zd [r] = row 4 g + r of input row tile j
            unsigned nh0, nl0, nh1, nl1, dh0, dl0, dh1, dl1;
            split_pair_acc(zn.x, zn.y, fz, nh0, nl0);
            split_pair_acc(zn.z, zn.w, fz, nh1, nl1);
            split_pair_acc(zd.x, zd.y, 1.0f, dh0, dl0);
            split_pair_acc(zd.z, zd.w, 1.0f, dh1, dl1);
            if (PAR == 0) { za[n][0].x = nh0; za[n][0].y = nh1; za[n][1].x = nl0; za[n][1].y = nl1; za[n][2].x = dh0; za[n][2].y = dh1; za[n][3].x = dl0; za[n][3].y = dl1; }
            else          { za[n][0].z = nh0; za[n][0].w = nh1; za[n][1].z = nl0; za[n][1].w = nl1; za[n][2].z = dh0; za[n][2].w = dh1; za[n][3].z = dl0; za[n][3].w = dl1; }
            __builtin_amdgcn_sched_barrier(0);        // (one column tile at a time: interleaved, the tiles' accumulators and constants do not fit 256 registers)
        }
        // ================= the block's sums (NSUM > 0): the row tile completed in the PREVIOUS step.  Here - the x-pass operands
        // are dead, the next loads not yet issued - the additions have the registers they need
        if (SH && j >= NB) reduce_tile(std::integral_constant<int, PAR ^ 1>{}, j - 1);
        const float escale = exp2i(E);
        // ================= the next step's loads fly during the y pass and the epilogues (the x-pass operands are dead by now:
        // issued before the x pass, the 30 registers of the loads in flight pushed the kernel over 256 and into scratch)
        if (j + 1 < NST) issue_loads(z, j + 1);
        else if (zround_next < z_end) issue_loads(SH ? min(zround_next + wave, z_end - 1) : zround_next, 0);
        // ================= y pass (scatter) and the epilogue of the completed row tile, per column tile
        // operand order: (slot 0, slot 1) = (previous, new) on odd steps, (new, previous) on even ones; the constant operands of
        // the y pass are the same for every column tile: read once per step.  Product m adds blocks (2 m, 2 m + 1) to the row tile
        // TB(m) = 2 m + 1 steps back (the last product: block NB - 1 alone, NB - 1 steps back - that row tile is then complete)
        constexpr int order = PAR ? 0 : 1;
        auto TB = [](int m) { return 2 * m + 1 <= NB - 1 ? 2 * m + 1 : NB - 1; };
        half8 bh[NQ], bl[NQ];
#pragma unroll
        for (int m = 0; m < NQ; ++m) {
            bh[m] = half8{}; bl[m] = half8{};
            const int i = j - TB(m);
            if (i >= 0 && i < nrt) { bh[m] = cB[((2 * NQ + order * NQ + m) * 2 + 0) * 64 + lane]; bl[m] = cB[((2 * NQ + order * NQ + m) * 2 + 1) * 64 + lane]; }
        }
#pragma unroll
        for (int n = 0; n < kCT; ++n) {
            if (j >= 1) {
                const half8 anh = as_half8(za[n][0]), anl = as_half8(za[n][1]), adh = as_half8(za[n][2]), adl = as_half8(za[n][3]);
#pragma unroll
                for (int m = 0; m < NQ - 1; ++m) {          // all but the completing product
                    const int i = j - TB(m);
                    constexpr int dummy = 0; (void)dummy;
                    if (i >= 0 && i < nrt) {
                        const int slot = ((PARN - TB(m)) % NP + NP) % NP;                 // row tile i mod NP (static)
                        f32x4 pn = m == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : Pn[slot][n], pd = m == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : Pd[slot][n];
                        pd = MFMA(adh, bh[m], pd);
                        pn = MFMA(anh, bh[m], pn);
                        pd = MFMA(adl, bh[m], pd);
                        pn = MFMA(anl, bh[m], pn);
                        pd = MFMA(adh, bl[m], pd);
                        pn = MFMA(anh, bl[m], pn);
                        Pn[slot][n] = pn; Pd[slot][n] = pd;
                    }
                }
                if (j >= NB - 1) {                  // last block of output row tile j - (NB - 1), then its epilogue
                    constexpr int ml = NQ - 1;
                    constexpr int slot = ((PARN - (NB - 1)) % NP + NP) % NP;
                    const int i = j - (NB - 1);
                    f32x4 pn = Pn[slot][n], pd = Pd[slot][n];
                    pd = MFMA(adh, bh[ml], pd);                     // (the denominator's chain ends first: its reciprocals run under the numerator's last product)
                    pn = MFMA(anh, bh[ml], pn);
                    pd = MFMA(adl, bh[ml], pd);
                    pn = MFMA(anl, bh[ml], pn);
                    pd = MFMA(adh, bl[ml], pd);
                    pn = MFMA(anh, bl[ml], pn);
                    // lane (m = output row, g): pn / pd [r] = output column 4 g + r of column tile n, row tile i
                    // den = 0 (empty window): 0 * inf = NaN.  The + 0.0 rides in the FMA and turns a -0.0 result into +0.0: the moment
                    // sums start at -0.0 and excluded voxels add -0.0, so a sum that is still -0.0 at the end says "no channel
                    // contributed" (nansum_allbadtonan) without a separate record of who was seen
                    f32x4 val;
                    val.x = __builtin_fmaf(pn.x, escale * __builtin_amdgcn_rcpf(pd.x), 0.f);
                    val.y = __builtin_fmaf(pn.y, escale * __builtin_amdgcn_rcpf(pd.y), 0.f);
                    val.z = __builtin_fmaf(pn.z, escale * __builtin_amdgcn_rcpf(pd.z), 0.f);
                    val.w = __builtin_fmaf(pn.w, escale * __builtin_amdgcn_rcpf(pd.w), 0.f);
                    const int yo = plane_row(16 * i + lm), xo = xw + 16 * n + 4 * lg;
                    const bool inside = (yo < ny) & (xo < nx);
                    if (STORE && inside) {
                        // one descriptor per plane, a 32-bit byte offset per lane
                        const auto ro = spc_plane_srd(A.out + (int64_t)z * A.out_plane_stride);
                        const unsigned off = ((unsigned)yo * (unsigned)A.out_row_stride + (unsigned)xo) * 4u;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, val), ro, (int)off, 0, 0);
                    }
                    if (NSUM) {
                        unsigned w = INC ? incsave[((PARN - HB) % NP + NP) % NP][n] : 0xffffffffu;   // saved by step j - HB (the centre row tile of i)
                        if (INC) asm volatile("" : "+v"(w));           // (tested HERE: hoisted to the top of the step, the 16 lane masks of the four tiles spill)
                        bool i0 = (w & 0xffu) != 0, i1 = (w & 0xff00u) != 0, i2 = (w & 0xff0000u) != 0, i3 = (w & 0xff000000u) != 0;
                        const bool live = inside & (!SH || alive);     // (a wave without a channel of its own in the chunk's last round repeats the last channel and adds nothing)
                        i0 = i0 & live; i1 = i1 & live; i2 = i2 & live; i3 = i3 & live;
                        if (INC != 1) {             // nansum: a NaN value is skipped (INC 1: an included voxel is a valid centre sample, its window is not empty)
                            i0 = i0 & (val.x == val.x); i1 = i1 & (val.y == val.y); i2 = i2 & (val.z == val.z); i3 = i3 & (val.w == val.w);
                        }
                        const f32x4 add = {i0 ? val.x : -0.f, i1 ? val.y : -0.f, i2 ? val.z : -0.f, i3 ? val.w : -0.f};
                        if (SH) xch[(size_t)((PAR * kW + wave) * kCT + n) * 64 + lane] = add;      // this wave's slot of the step's exchange buffer
                        else { f32x4* const slot = macc + (size_t)(i * kCT + n) * 64; *slot = *slot + add; }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    issue_loads(SH ? min(z_begin + wave, z_end - 1) : z_begin, 0);
    for (int zr = z_begin; zr < z_end; zr += ZSTEP) {
        // (NSUM > 0) a wave whose channel of this round lies beyond the chunk repeats the chunk's last channel - the same steps, the
        // same barriers - and adds nothing
        alive = !SH || zr + wave < z_end;
        zcur = SH ? min(zr + wave, z_end - 1) : zr;
        zround_next = zr + ZSTEP;
        // (NSUM = 0: one rendezvous per channel keeps the block's four waves - neighbours in x, 32 shared columns each - in the
        //  same channel; waves that left above do not count)
        if (!SH && A.sync) __builtin_amdgcn_s_barrier();
        E = -128;
        zround = zr;
#pragma unroll 1
        for (int jj = 0; jj < NST; jj += NP) {
            step(std::integral_constant<int, 0>{}, jj);
            if (jj + 1 < NST) step(std::integral_constant<int, 1>{}, jj + 1);
            if (NP > 2) {
                if (jj + 2 < NST) step(std::integral_constant<int, 2 % NP>{}, jj + 2);
                if (jj + 3 < NST) step(std::integral_constant<int, 3 % NP>{}, jj + 3);
            }
        }
        if (SH) reduce_tile(std::integral_constant<int, 1>{}, NST - 1);      // the round's last row tile (NRT - 1: odd)
    }

    if (NSUM == 1) {
        const int64_t plane = (int64_t)A.ny * A.nx;
#pragma unroll
        for (int tt = 0; tt < NRT * kCT; ++tt) {
            const int i = tt / kCT, n = tt % kCT;
            const int yo = plane_row(16 * i + lm), xo = xw + 16 * n + 4 * lg;
            if (yo < ny && xo < nx)
                *reinterpret_cast<f32x4*>(A.partial + (int64_t)chunk * plane + (int64_t)yo * A.nx + xo) = macc[(size_t)(i * kCT + n) * 64];
        }
    }
    if (SH) {
        __syncthreads();                                   // the last step's additions
        const int64_t plane = (int64_t)A.ny * A.nx;
        for (int tt = wave; tt < NRT * kCT; tt += kW) {    // the band's tiles, dealt to the waves
            const int i = tt / kCT, n = tt % kCT;
            const int yo = plane_row(16 * i + lm), xo = xw + 16 * n + 4 * lg;
            if (yo < ny && xo < nx) {
                const int64_t at = (int64_t)chunk * plane + (int64_t)yo * A.nx + xo;
#pragma unroll
                for (int q = 0; q < NSUM; ++q)
                    *reinterpret_cast<f32x4*>(A.partial + (int64_t)q * A.nchunk * plane + at) = msum[(size_t)((i * kCT + n) * NSUM + q) * 64 + lane];
            }
        }
    }
}

// moments of the smoothed cube from the chunk sums (float64 across chunks): m0 = dv S0 (NaN where no channel contributed:
// nansum_allbadtonan, dask_spectral_cube.py:54-59), m1 = S1 / S0 + m1_add, m2 = sum v (c - m1)^2 / S0   (:1083-1104).
// A chunk's sums S1', S2' are about ITS middle channel c_k (float32: |c - c_k| is at most 32 channel widths); here, in float64,
//   S1 = sum_k S1'_k + c_k S0_k,      m2 S0 = sum_k S2'_k + 2 (c_k - mu) S1'_k + (c_k - mu)^2 S0_k
// - the second moment about the mean, chunk by chunk: what float32 rounding of a chunk's sums costs is bounded by the chunk's
// width, not by the distance of the line from the reference channel.
__global__ __launch_bounds__(256) void split_finish_kernel(const float* partial, const double* cen, int nsum, int nchunk, int zchunk, int64_t nz,
                                                            int64_t ny, int64_t nx, double dv, double m1_add,
                                                            double* m0, double* m1, double* m2, int64_t map_row_stride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = ny * nx;
    if (i >= plane) return;
    double s0 = 0.0, s1 = 0.0;
    unsigned any = 0;
    for (int c = 0; c < nchunk; ++c) {
        const float p0 = partial[(int64_t)c * plane + i];
        any |= (__float_as_uint(p0) != 0x80000000u) ? 1u : 0u;      // still -0.0: nothing was added in this chunk
        s0 += (double)p0;
        if (nsum == 3) {
            const double ck = cen[min((int64_t)c * zchunk + zchunk / 2, nz - 1)];
            s1 += (double)partial[((int64_t)nchunk + c) * plane + i] + ck * (double)p0;
        }
    }
    const int64_t y = i / nx, x = i - y * nx;
    const int64_t o = y * map_row_stride + x;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    if (m0) m0[o] = any ? dv * s0 : nan;
    if (m1 || m2) {
        const double mu = s1 / s0;                             // 0 / 0 = NaN for rays without a contribution, like the reference
        if (m1) m1[o] = mu + m1_add;
        if (m2) {
            double s2 = 0.0;
            for (int c = 0; c < nchunk; ++c) {
                const double d = cen[min((int64_t)c * zchunk + zchunk / 2, nz - 1)] - mu;
                s2 += (double)partial[((int64_t)2 * nchunk + c) * plane + i] + d * (2.0 * (double)partial[((int64_t)nchunk + c) * plane + i]
                                                                                   + d * (double)partial[(int64_t)c * plane + i]);
            }
            m2[o] = s2 / s0;
        }
    }
}

// (three sums) channel coordinate about the middle channel of the channel's chunk, as float32; zeros beyond nz
__global__ __launch_bounds__(256) void split_delta_kernel(const double* cen, float* dlt, int64_t nz, int zchunk, int64_t n) {
    const int64_t z = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (z >= n) return;
    const int64_t zm = min((z / zchunk) * zchunk + zchunk / 2, nz - 1);
    dlt[z] = z < nz ? (float)(cen[z] - cen[zm]) : 0.f;
}

inline int split_chunk_planes(int64_t nz, int64_t tiles, int nsum) {
    // enough blocks to fill the chip several times over, chunks of at most 64 channels (float32 sums inside a chunk); with sums a
    // block walks eight channels at a time: whole rounds of eight
    int64_t want_chunks = std::max<int64_t>(1, ((nsum == 3 ? 1024 : 2048) + tiles - 1) / tiles);
    int64_t zc = std::max<int64_t>(1, (nz + want_chunks - 1) / want_chunks);
    if (nsum == 3) zc = (zc + 7) / 8 * 8;
    return (int)std::min<int64_t>(zc, 64);
}

// output row tiles per band of the forms with sums
//   one sum: what the LDS holds next to the constant operands at two blocks per CU (80 KB each), every wave its own sums
//     NB = 3 (16 KB of constants): 4 x 4 KB per wave (64 rows)        NB = 5 (24 KB): 7 x 2 KB per wave (112 rows)
//   three sums: what the LDS of a CU (160 KB, one block of eight waves) holds next to the constant operands and the two
//   exchange buffers (2 x 8 waves x CT KB), one set of sums per block
//     NB = 3 (16 KB + 64 KB): 6 x 12 KB (96 rows: 152 KB)             NB = 5 (24 KB + 32 KB): 16 x 6 KB (256 rows: 152 KB)
constexpr int kNRT1 = 4, kNRT3 = 6, kNRT1w = 7, kNRT3w = 16;

struct SplitGeo { int nrt, oc, waves_x; };     // waves_x: waves of a block side by side in x (4; the three-sum form: 1)
inline SplitGeo split_geo(int nb, int nsum) {
    const int wx = nsum == 3 ? 1 : kWaves;
    return nb == 3 ? SplitGeo{nsum == 3 ? kNRT3 : kNRT1, Geo<3>::OC, wx} : SplitGeo{nsum == 3 ? kNRT3w : kNRT1w, Geo<5>::OC, wx};
}

}  // namespace

bool spc_spatial_split_takes(const spc_cube_f32* cube, const MaskDev& md) {
    if ((cube->nx & 3) || (cube->row_stride & 3) || (cube->plane_stride & 3) || (((uintptr_t)cube->d_data) & 15)) return false;
    if ((md.flags & SPC_MASK_ARRAY) && ((md.row_stride & 3) || (md.plane_stride & 3) || (((uintptr_t)md.arr) & 3))) return false;
    if (cube->ny * cube->row_stride * 4 >= (1ll << 32)) return false;
    return true;
}

size_t spc_ws_spatial_split(int64_t nz, int64_t ny, int64_t nx, int nsum) {
    size_t need = 0;
    for (int nb : {3, 5}) {
        const SplitGeo g = split_geo(nb, nsum);
        const int64_t tiles = ((ny + 16 * g.nrt - 1) / (16 * g.nrt)) * ((nx + g.waves_x * g.oc - 1) / (g.waves_x * g.oc));
        const int64_t zc = split_chunk_planes(nz, tiles, nsum), nchunk = (nz + zc - 1) / zc;
        need = std::max(need, spc_ws_round((size_t)nsum * nchunk * ny * nx * sizeof(float)) + spc_ws_round((size_t)nchunk * zc * sizeof(float)) + 512);
    }
    return need;
}

// launches the split kernel (+ the finish kernel); the caller has validated the kernel taps (non-negative, positive centre,
// centred in arrays of `ntaps` = 33 or 65 entries) and the mask terms (array / isfinite only).  nsum = 0 / 1 / 3.
int spc_spatial_split_launch(hipStream_t st, const spc_cube_f32* cube, const MaskDev& md, const float* ky, const float* kx, int ntaps,
                             float sy, float sx, float* d_out, int64_t out_row_stride, int64_t out_plane_stride,
                             int nsum, double dv, double m1_add, const double* d_cen, double* d_m0, double* d_m1, double* d_m2,
                             int64_t map_row_stride, void* d_workspace, size_t workspace_bytes) {
    Sm3Args A{};
    A.cube = cube->d_data; A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.marr = (md.flags & SPC_MASK_ARRAY) ? md.arr : nullptr;
    A.mrow_stride = md.row_stride; A.mplane_stride = md.plane_stride;
    A.out = d_out; A.out_row_stride = out_row_stride; A.out_plane_stride = out_plane_stride;
    A.cen = d_cen;
    A.lim = (md.flags & SPC_MASK_FINITE) ? 3.402823466e+38f : INFINITY;
    A.sy = sy; A.sx = sx;
    // measured at 256 x 2048^2 + uint8 mask (profiles/r05_split_fetch_ab.txt): fetch / algorithmic x1.97 as built first, x1.52 with
    // mirrored odd bands, x1.43 with the rendezvous per channel as well; the time does not move (the kernel is bound by issue).
    // The three-sum form (16-row regions, three times the steps per channel) loses 8 % to the rendezvous: not there.
    { const char* e = getenv("SPC_SPLIT_MIRROR"); A.mirror = e ? atoi(e) : 1; e = getenv("SPC_SPLIT_SYNC"); A.sync = e ? atoi(e) : 1;
      e = getenv("SPC_SPLIT_WIDE_MASK"); A.wide_mask = e ? atoi(e) : 1; }
    SPC_REQUIRE(ntaps == Geo<3>::R || ntaps == Geo<5>::R, "internal: the split form takes taps padded to 33 or 65 entries");
    const int nb = ntaps == Geo<3>::R ? 3 : 5;
    for (int i = 0; i < kMaxTaps + 3; ++i) { A.ky[i] = i < ntaps ? ky[i] : 0.f; A.kx[i] = i < ntaps ? kx[i] : 0.f; }
    const SplitGeo geo = split_geo(nb, nsum);
    int nrt = geo.nrt;
    if (nsum == 0) {
        // cube -> cube: nothing of a band lives in LDS, so a wave marches down the WHOLE column (round 6): NRT + 2 HB steps for NRT
        // output row tiles - 130 for 128 at 2048 rows, where the 64-row bands took 192 (SPC_SPLIT_BAND_TILES: row tiles per band)
        static const int env_tiles = [] { const char* e = getenv("SPC_SPLIT_BAND_TILES"); return e ? atoi(e) : 0; }();
        const int64_t all = (cube->ny + 15) / 16;
        nrt = (int)(env_tiles > 0 ? std::min<int64_t>(env_tiles, all) : all);
    }
    A.nrt = nrt;
    A.nstrips = (int)((cube->nx + geo.waves_x * geo.oc - 1) / (geo.waves_x * geo.oc));
    A.nbands = (int)((cube->ny + 16 * nrt - 1) / (16 * nrt));
    A.zchunk = split_chunk_planes(cube->nz, (int64_t)A.nstrips * A.nbands, nsum);
    A.nchunk = (int)((cube->nz + A.zchunk - 1) / A.zchunk);
    const int64_t nblocks = (int64_t)A.nstrips * A.nbands * A.nchunk;
    SPC_REQUIRE(nblocks < (1ll << 31), "too many blocks");
    if (nsum) {
        SpcWorkspace ws(d_workspace, workspace_bytes);
        SPC_WS_TAKE(d_partial, ws, float, (size_t)nsum * A.nchunk * cube->ny * cube->nx);
        SPC_WS_TAKE(d_dlt, ws, float, (size_t)A.nchunk * A.zchunk);
        A.partial = d_partial; A.dlt = d_dlt;
        if (nsum == 3) {
            const int64_t n = (int64_t)A.nchunk * A.zchunk;
            hipLaunchKernelGGL(split_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_cen, d_dlt, cube->nz, A.zchunk, n);
            SPC_LAUNCH_CHECK();
        }
    }
    const bool arr = A.marr != nullptr, fin = (md.flags & SPC_MASK_FINITE) != 0, store = d_out != nullptr;
    dim3 grid((unsigned)nblocks), block(nsum == 3 ? 512 : kThreads);
    // WIDE: the mask rows of a step in two requests per lane (see issue_loads) - for planes whose width is a multiple of 16 columns
    // and mask rows on 4-byte boundaries (spc_spatial_split_takes); SPC_SPLIT_WIDE_MASK=0: six dword loads per lane as before
    const bool wide = arr && A.wide_mask && (cube->nx % 16 == 0) && cube->nx >= 16;
#define SPC_S3(NB_, NRT_, ARR_, INC_, STORE_, NSUM_, WIDE_) hipLaunchKernelGGL((spatial_split_kernel<NB_, NRT_, ARR_, INC_, STORE_, NSUM_, WIDE_>), grid, block, 0, st, A)
#define SPC_S3_ARR(NB_, NRT_, INC_, STORE_, NSUM_) do { if (wide) SPC_S3(NB_, NRT_, true, INC_, STORE_, NSUM_, true); else SPC_S3(NB_, NRT_, true, INC_, STORE_, NSUM_, false); } while (0)
#define SPC_S3_MASK(NB_, NRT_, STORE_, NSUM_) do { if (arr && fin) SPC_S3_ARR(NB_, NRT_, 1, STORE_, NSUM_); else if (arr) SPC_S3_ARR(NB_, NRT_, 2, STORE_, NSUM_); \
                                                   else if (fin) SPC_S3(NB_, NRT_, false, 1, STORE_, NSUM_, false); else SPC_S3(NB_, NRT_, false, 0, STORE_, NSUM_, false); } while (0)
    if (nb == 3) {
        if (nsum == 3) { if (store) SPC_S3_MASK(3, kNRT3, true, 3); else SPC_S3_MASK(3, kNRT3, false, 3); }
        else if (nsum == 1) { if (store) SPC_S3_MASK(3, kNRT1, true, 1); else SPC_S3_MASK(3, kNRT1, false, 1); }
        else SPC_S3_MASK(3, 0, true, 0);
    } else {
        if (nsum == 3) { if (store) SPC_S3_MASK(5, kNRT3w, true, 3); else SPC_S3_MASK(5, kNRT3w, false, 3); }
        else if (nsum == 1) { if (store) SPC_S3_MASK(5, kNRT1w, true, 1); else SPC_S3_MASK(5, kNRT1w, false, 1); }
        else SPC_S3_MASK(5, 0, true, 0);
    }
    SPC_LAUNCH_CHECK();
    if (nsum) {
        const int64_t n = cube->ny * cube->nx;
        hipLaunchKernelGGL(split_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A.partial, A.cen, nsum, A.nchunk, A.zchunk,
                           cube->nz, cube->ny, cube->nx, dv, m1_add, d_m0, d_m1, d_m2, map_row_stride ? map_row_stride : cube->nx);
        SPC_LAUNCH_CHECK();
    }
    return SPC_OK;
}
