// Order statistics along the spectral axis: median / percentile / mad_std per spaxel
// (SURVEY.md section 8f rank 4).  Replaces da.nanmedian / np.nanpercentile / astropy
// stats.mad_std applied per ray by DaskSpectralCubeMixin.median / percentile / mad_std
// (spectral_cube/dask_spectral_cube.py:657-731), which sort or partition every ray on the host.
//
// No sort: the k-th smallest of a ray is found by descending the bits of an order-preserving
// integer key (sign-flipped IEEE bits), most significant first.  One pass over z counts, for the
// current prefix, the samples by their next TWO key bits; 16 passes pin the key down exactly.  Two
// ranks are followed at once (the two order statistics a percentile interpolates between), a
// lane owns 4 adjacent spaxels (16-byte coalesced loads), all state lives in registers.  Cost:
// 17 streaming reads of the cube - no scratch memory, any nz.  For mad_std the same kernel runs
// on |x - centre| with the per-spaxel median as centre.
#include "spc_common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

#ifndef SPC_SEL_INFLIGHT
#define SPC_SEL_INFLIGHT 8
#endif
#ifndef SPC_SEL_FIRST_VOTE
#define SPC_SEL_FIRST_VOTE 1
#endif
#ifndef SPC_SEL_INFLIGHT_WIDE
#define SPC_SEL_INFLIGHT_WIDE 32
#endif

namespace {

struct SelArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    double q;                 // percentile in [0, 100]
    const float* center;      // NULL, or (ny, nx) map: select on |x - center|
    float scale;              // result multiplier (mad_std: 1.4826...)
    float* out;               // (ny, nx)
    // register-resident kernel only: step between adjacent spaxels of a row (1 for rays along z or y; the row stride of
    // the cube when the rays run along x) and whether consecutive LANES should walk along the ray (rays along x: the
    // samples of a ray are the contiguous ones)
    int64_t x_stride, m_x_stride;
    int along_ray;
    uint32_t key_min, key_span;            // sel_key_range of the mask's predicate terms (filled by the launcher)
    int ablate;                            // timing experiments only (SPC_SELECT_ABLATE): 1 = no descent, 2.. = passes
    int xcd_group;                         // tiles of a row kept on one XCD (sel_tile_of_block)
};

// Block -> tile order that keeps `g` neighbouring tiles of a row on ONE XCD (blocks go round the 8 XCDs by index): a tile
// of 16 spaxels is half a 128-byte line per plane, and the block holding the other half should find it in the same L2.
__device__ __forceinline__ int64_t sel_tile_of_block(int64_t bid, int64_t nblocks, int g) {
    if (g <= 1) return bid;
    const int64_t span = 8 * (int64_t)g, whole = nblocks / span * span;
    if (bid >= whole) return bid;
    const int64_t base = bid / span * span, r = bid - base;
    return base + (r % 8) * g + r / 8;
}

__device__ __forceinline__ uint32_t fkey(float v) {
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

typedef float f32x4s __attribute__((ext_vector_type(4)));

template <int VEC, bool ARR>
__global__ __launch_bounds__(256) void select_axis0_kernel(const SelArgs A) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t gpr = (A.nx + VEC - 1) / VEC;
    if (g >= A.ny * gpr) return;
    const int64_t y = g / gpr, x = (g - y * gpr) * VEC;
    const float* p = A.cube + y * A.row_stride + x;
    const uint8_t* pm = ARR ? A.mask.arr + y * A.mask.row_stride + x : nullptr;
    float cen[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) cen[c] = (A.center && x + c < A.nx) ? A.center[y * A.nx + x + c] : 0.f;
    const bool use_cen = A.center != nullptr;

    // one plane's samples of this lane: selection value and validity (0 / 1)
    struct Smp { float v[VEC]; int ok[VEC]; };
    auto load = [&](int64_t z) -> Smp {
        Smp r;
        float raw[VEC];
        unsigned mk[VEC];
        if (VEC == 4) {
            const f32x4s q4 = *reinterpret_cast<const f32x4s*>(p + z * A.plane_stride);
#pragma unroll
            for (int c = 0; c < VEC; ++c) raw[c] = q4[c];
            if (ARR) {
                const uint32_t m = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pm + z * A.mask.plane_stride));
#pragma unroll
                for (int c = 0; c < VEC; ++c) mk[c] = (m >> (8 * c)) & 0xffu;
            }
        } else {
            raw[0] = p[z * A.plane_stride];
            if (ARR) mk[0] = pm[z * A.mask.plane_stride];
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            bool ok = spc_pred_valid(A.mask, raw[c]);
            if (ARR) ok = ok && (mk[c] != 0);
            const float v = use_cen ? fabsf(raw[c] - cen[c]) : raw[c];
            r.v[c] = v;
            r.ok[c] = (ok && (v == v)) ? 1 : 0;
        }
        return r;
    };

    // pass 0: valid counts -> the two ranks of numpy's 'linear' percentile
    int n[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) n[c] = 0;
#pragma unroll 2
    for (int64_t z = 0; z < A.nz; ++z) {
        const Smp sm = load(z);
#pragma unroll
        for (int c = 0; c < VEC; ++c) n[c] += sm.ok[c];
    }
    int klo[VEC], khi[VEC];
    double frac[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        const double pos = A.q / 100.0 * (double)(n[c] > 0 ? n[c] - 1 : 0);
        const double fl = floor(pos);
        klo[c] = (int)fl;
        khi[c] = min((int)ceil(pos), max(n[c] - 1, 0));
        frac[c] = pos - fl;
    }
    uint32_t plo[VEC], phi[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) { plo[c] = 0u; phi[c] = 0u; }
    // two key bits per pass (16 passes): for each followed rank, count the prefix-matching samples
    // whose digit is 0, 1 or 2 (3 is the rest)
#pragma unroll 1
    for (int b = 30; b >= 0; b -= 2) {
        int clo[VEC][3], chi[VEC][3];
#pragma unroll
        for (int c = 0; c < VEC; ++c)
#pragma unroll
            for (int j = 0; j < 3; ++j) { clo[c][j] = 0; chi[c][j] = 0; }
#pragma unroll 2
        for (int64_t z = 0; z < A.nz; ++z) {
            const Smp sm = load(z);
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const uint32_t k = fkey(sm.v[c]);
                const uint32_t dg = (k >> b) & 3u;
                // (b + 2 == 32 in the first pass: no prefix yet; a 64-bit shift keeps that well defined)
                const int mlo = ((((uint64_t)(k ^ plo[c])) >> (b + 2)) == 0u ? 1 : 0) & sm.ok[c];
                const int mhi = ((((uint64_t)(k ^ phi[c])) >> (b + 2)) == 0u ? 1 : 0) & sm.ok[c];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int is = (dg == (uint32_t)j) ? 1 : 0;
                    clo[c][j] += mlo & is;
                    chi[c][j] += mhi & is;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            uint32_t d = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) if (klo[c] >= clo[c][j] && d == (uint32_t)j) { klo[c] -= clo[c][j]; d = j + 1; }
            plo[c] |= d << b;
            d = 0;
#pragma unroll
            for (int j = 0; j < 3; ++j) if (khi[c] >= chi[c][j] && d == (uint32_t)j) { khi[c] -= chi[c][j]; d = j + 1; }
            phi[c] |= d << b;
        }
    }
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        if (x + c >= A.nx) break;
        float res = NAN;
        if (n[c] > 0) {
            const double a = (double)funkey(plo[c]), bb = (double)funkey(phi[c]);
            // numpy's linear interpolation between the two order statistics (a + (b - a) * t; the
            // median of an even count is their mean)
            const double t = frac[c];
            res = (float)(((A.q == 50.0 && a == bb) ? a : (t == 0.5 ? 0.5 * (a + bb) : a + (bb - a) * t)) * (double)A.scale);   // (median of equal infinities)
        }
        A.out[y * A.nx + x + c] = res;
    }
}

// ---- four key bits per pass with per-spaxel histograms in LDS (8 reads of the cube instead of 17) ----
// Same descent, radix 16: a lane owns 4 adjacent spaxels and, for each of the two followed ranks, a
// 16-bin histogram per spaxel in LDS (two 16-bit counters per word: nz <= 65535; layout
// [word][rank][spaxel] so that a wave's updates fall into consecutive banks).  The bins are private to
// the lane, so the updates are fire-and-forget ds_add and no barrier is ever needed.  The first pass
// doubles as the valid count.  While both ranks still share their prefix (always, for an odd count)
// only one histogram is maintained.
constexpr int kSelSpaxels = 1024;                      // 256 lanes x 4

template <bool ARR>
__global__ __launch_bounds__(256) void select16_axis0_kernel(const SelArgs A) {
    __shared__ unsigned int hist[8][2][kSelSpaxels];     // 64 KB
    const int t = threadIdx.x;
    const int64_t g = (int64_t)blockIdx.x * 256 + t;
    const int64_t gpr = A.nx / 4;
    const bool live = g < A.ny * gpr;
    const int64_t gg = live ? g : 0;
    const int64_t y = gg / gpr, x = (gg - y * gpr) * 4;
    const float* p = A.cube + y * A.row_stride + x;
    const uint8_t* pm = ARR ? A.mask.arr + y * A.mask.row_stride + x : nullptr;
    const bool use_cen = A.center != nullptr;
    float cen[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) cen[c] = use_cen ? A.center[y * A.nx + x + c] : 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w)
#pragma unroll
        for (int c = 0; c < 4; ++c) { hist[w][0][4 * t + c] = 0u; hist[w][1][4 * t + c] = 0u; }

    int n[4] = {0, 0, 0, 0}, klo[4] = {0, 0, 0, 0}, khi[4] = {0, 0, 0, 0};
    double frac[4] = {0, 0, 0, 0};
    uint32_t plo[4] = {0, 0, 0, 0}, phi[4] = {0, 0, 0, 0};
    bool same[4] = {true, true, true, true};
#pragma unroll 1
    for (int b = 28; b >= 0; b -= 4) {
#pragma unroll 2
        for (int64_t z = 0; z < A.nz; ++z) {
            const f32x4s q4 = *reinterpret_cast<const f32x4s*>(p + z * A.plane_stride);
            uint32_t m = 0x01010101u;
            if (ARR) m = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pm + z * A.mask.plane_stride));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float raw = q4[c];
                bool ok = spc_pred_valid(A.mask, raw) & (((m >> (8 * c)) & 0xffu) != 0);
                const float v = use_cen ? fabsf(raw - cen[c]) : raw;
                ok = ok && (v == v);
                const uint32_t k = fkey(v);
                const uint32_t dg = (k >> b) & 15u;
                const uint32_t inc = 1u << (16 * (dg & 1u));
                const bool mlo = ok && ((((uint64_t)(k ^ plo[c])) >> (b + 4)) == 0u);
                if (mlo) atomicAdd(&hist[dg >> 1][0][4 * t + c], inc);
                if (!same[c]) {
                    const bool mhi = ok && ((((uint64_t)(k ^ phi[c])) >> (b + 4)) == 0u);
                    if (mhi) atomicAdd(&hist[dg >> 1][1][4 * t + c], inc);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned cnt[2][16];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const unsigned a0 = hist[w][0][4 * t + c];
                const unsigned a1 = same[c] ? a0 : hist[w][1][4 * t + c];
                cnt[0][2 * w] = a0 & 0xffffu; cnt[0][2 * w + 1] = a0 >> 16;
                cnt[1][2 * w] = a1 & 0xffffu; cnt[1][2 * w + 1] = a1 >> 16;
                hist[w][0][4 * t + c] = 0u; hist[w][1][4 * t + c] = 0u;
            }
            if (b == 28) {                                   // first pass: every valid sample was counted
                int tot = 0;
#pragma unroll
                for (int d = 0; d < 16; ++d) tot += (int)cnt[0][d];
                n[c] = tot;
                const double pos = A.q / 100.0 * (double)(tot > 0 ? tot - 1 : 0);
                const double fl = floor(pos);
                klo[c] = (int)fl;
                khi[c] = min((int)ceil(pos), max(tot - 1, 0));
                frac[c] = pos - fl;
            }
            uint32_t dlo = 15, dhi = 15;
            bool flo = false, fhi = false;
#pragma unroll
            for (int d = 0; d < 15; ++d) {
                if (!flo) { if (klo[c] < (int)cnt[0][d]) { dlo = d; flo = true; } else klo[c] -= (int)cnt[0][d]; }
                if (!fhi) { if (khi[c] < (int)cnt[1][d]) { dhi = d; fhi = true; } else khi[c] -= (int)cnt[1][d]; }
            }
            plo[c] |= dlo << b;
            phi[c] |= dhi << b;
            same[c] = same[c] && (dlo == dhi);
        }
    }
    if (!live) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float res = NAN;
        if (n[c] > 0) {
            const double a = (double)funkey(plo[c]), bb = (double)funkey(phi[c]);
            const double tt = frac[c];
            res = (float)(((A.q == 50.0 && a == bb) ? a : (tt == 0.5 ? 0.5 * (a + bb) : a + (bb - a) * tt)) * (double)A.scale);
        }
        A.out[y * A.nx + x + c] = res;
    }
}


// ---- rays resident in REGISTERS: ONE read of the cube (round 2) ------------------------------------------
// The descents above stream the cube once per digit (8 or 17 reads): 18.7 ms for a median at 1024^3, 3.6 % of the
// roofline of a single read.  Here a block owns TS adjacent spaxels (TS x 4 bytes contiguous per plane: 32- to
// 128-byte segments) and reads their rays ONCE: 256 / TS lanes share a ray, a lane turns its KPL = ceil(nz / lanes)
// <= 64 samples (z = j + lanes * i) into order-preserving keys (excluded / NaN samples become the largest key,
// which no valid float maps to) and keeps them in registers.  A digit pass of the radix-16 descent is then 7 VALU
// instructions per key: sixteen 4-bit counters in ONE 64-bit register (the digit picks the field: one shift, one
// add, no dynamic indexing, no atomics), emptied into 8-bit counters every 15 keys; the per-ray totals meet in a
// 16-counter LDS histogram (three rotating buffers: one barrier per digit) and every lane of the ray walks the
// sixteen totals to the digit.  LDS only carries the histograms (a few KB): the ~105 VGPRs set the occupancy
// (4 waves per SIMD).  The descent stops early: once four digits (16 bits) are known, the samples that still match
// are few (their number is the count of the selected bin); if no ray of the block has more than kCandMax of them,
// they are gathered into LDS and ranked directly (each lane of the ray ranks its share against all candidates)
// instead of four more passes over all keys.  Rays with many equal or near-equal samples (quantised data) take all
// eight passes.  The upper order statistic of an interpolated percentile is either the same value (ties) or the
// smallest key above the first: one more sweep over the registers.
// (A first version kept the keys in LDS - 64 KB per block, 2 waves per SIMD, an LDS read per key and pass: 6.25 ms
// where this one takes 2.5 - 3.2 ms.)
constexpr int kCandMax = 32;

// What a descent ranks: the resident keys themselves, or - for the MAD - the keys of |x - centre| formed on the fly
// from the resident keys (float32 subtraction like numpy's; the excluded key stays the largest)
struct KeyIdentity {
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return k; }
};
struct KeyAbsDev {
    float center;
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const {
        const float v = fabsf(funkey(k) - center);
        return (k != 0xffffffffu && v == v) ? fkey(v) : 0xffffffffu;
    }
};

template <int KPL, bool FIRST, class XF>
__device__ __forceinline__ void count_keys(const uint32_t (&key)[KPL], uint32_t prefix, int b, const XF& xf,
                                           unsigned long long& accE, unsigned long long& accO) {
    constexpr unsigned long long kNib = 0x0f0f0f0f0f0f0f0full;
    unsigned long long a = 0ull;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        const uint32_t ki = xf(key[i]);
        if (FIRST) {
            a += 1ull << ((ki >> 28) * 4u);
        } else {
            const uint32_t tt = (ki ^ prefix) >> b;              // < 16 exactly for the samples inside the prefix
            const unsigned long long one = (tt < 16u) ? 1ull : 0ull;
            a += one << ((tt * 4u) & 63u);
        }
        if (i % 15 == 14 || i == KPL - 1) {                      // a 4-bit field holds 15
            accE += a & kNib;
            accO += (a >> 4) & kNib;
            a = 0ull;
            __builtin_amdgcn_sched_barrier(0);                   // (keeps the scheduler from expanding all keys at once)
        }
    }
}

// ---- the one read of the cube: TS rays into register keys ------------------------------------------------
// Lane (r, j) takes samples z = j + L k (L = 256 / TS lanes per ray, k < KPL) of ray r; an excluded / NaN / out-of-range
// sample becomes the largest key.
//  * Validity is ONE unsigned range test on the key: NaN samples, isfinite and the threshold comparisons of the mask all
//    select an interval of the order-preserving key (sel_key_range on the host), so (key - kmin) < span replaces round 2's
//    five float compares and their scalar ANDs (35 - 40 issue slots per key with the addressing below - as much as the
//    four digit passes of the selection that follows).  A lane whose column lies beyond the image gets span 0.
//  * DESC: a key slot is addressed through a buffer descriptor whose base is advanced by L planes per slot in SCALAR
//    arithmetic; a lane's offset (its column and its first plane) is one 32-bit register shared by all its loads.  Slots
//    that can run past the last plane (only the upper half of the slots can, by the launcher's choice of KPL) send the
//    lanes beyond it to offset 0 of the last slot that exists - always inside the cube - and drop the sample by the
//    validity test.  The launcher picks DESC when L planes and TS columns stay below 2 GiB of byte offset
//    (sel_desc_fits); larger planes keep 64-bit per-lane addresses (clamped to the last plane).
// cen: select on |x - cen| when CEN (mad_std).
static inline bool spc_env_on(const char* name) { const char* e = getenv(name); return e ? atoi(e) != 0 : true; }
static inline bool spc_env_set(const char* name) { const char* e = getenv(name); return e ? atoi(e) != 0 : false; }
static inline bool sel_desc_fits(int ts, int64_t plane_stride, int64_t x_stride, int bt = 256) {
    const int64_t L = bt / ts;
    return (L * plane_stride + ts * x_stride) * 4 < (1ll << 31);
}
static inline uint32_t sel_host_key(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// mask predicate (isfinite, > >= < <= thresholds; NaN never valid) -> the keys it includes: [*kmin, *kmin + *span)
static inline void sel_key_range(uint32_t flags, float thr_lo, float thr_hi, uint32_t* kmin, uint32_t* span) {
    float lim, lo, hi;
    spc_canonical_pred(flags, thr_lo, thr_hi, &lim, &lo, &hi);              // |v| <= lim && !(v <= lo) && !(v >= hi)
    *kmin = 0u;
    *span = 0u;
    if (!(lim >= 0.f)) return;
    uint64_t a = sel_host_key(-lim), b = sel_host_key(lim);
    if (lo == lo) a = std::max<uint64_t>(a, (uint64_t)sel_host_key(lo == 0.f ? 0.f : lo) + 1u);      // v > lo; both zeros: > +0
    if (hi == hi) {
        const uint64_t h = sel_host_key(hi == 0.f ? -0.f : hi);                                     // v < hi; both zeros: < -0
        if (h == 0u) return;
        b = std::min<uint64_t>(b, h - 1u);
    }
    if (b < a) return;
    *kmin = (uint32_t)a;
    *span = (uint32_t)(b - a + 1u);
}

__device__ __forceinline__ uint32_t fkey_bits(uint32_t u) {
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}

template <int TS, int KPL, bool ARR, bool DESC, bool CEN, int BT>
__device__ __forceinline__ int sel_load_keys_impl(const float* cube, int64_t plane_stride, int64_t x_stride, int64_t tile_off,
                                                  const uint8_t* marr, int64_t m_plane_stride, int64_t m_x_stride, int64_t m_tile_off,
                                                  int rc, int j, int nz, bool col_in, uint32_t kmin, uint32_t span1,
                                                  float cen, uint32_t (&key)[KPL]) {
    constexpr int L = BT / TS;
    // loads in flight per lane: the raw samples land in the key registers themselves (converted in place), so a whole
    // group costs no registers beyond the mask bytes.  Round 2 / 3a kept 8 in flight (2 KB per wave, 40 KB per CU): the
    // load phase ran at 2.1 TB/s - latency-bound - and made up 2.0 of the kernel's 2.9 ms at 1024^3.
    // 512-thread blocks (round 4, tools/bench_select_bt.py at 1024^3): 16 / 32 / 64 in flight - no mask 1.95 / 1.87 / 1.93 ms against 2.00,
    // uint8 mask 2.18 / 2.33 ms against 2.24 (its bytes wait in registers of their own); the 256-thread table loses with more than 8.
    constexpr int kWant = BT == 512 ? (ARR ? 16 : SPC_SEL_INFLIGHT_WIDE) : SPC_SEL_INFLIGHT;
    constexpr int U = KPL < kWant ? KPL : kWant;
    constexpr int kChk0 = (KPL == 16) ? 0 : KPL / 2;             // first slot that can run past the last plane
    int mine = 0;
    const uint32_t span_l = col_in ? span1 : 0u;
    const int last = (nz - 1) / L;                               // the last slot that holds a sample (uniform)
    // which of the slots kChk0 .. KPL - 1 hold a sample of THIS lane: a bit each (sign-extended bit extracts below - no
    // compare, no scalar mask per slot)
    const int nk = min(max((nz - j + L - 1) / L - kChk0, 0), KPL - kChk0);
    const unsigned long long vbits = nk >= 64 ? ~0ull : ((1ull << nk) - 1ull);
    const uint32_t vb0 = (uint32_t)vbits, vb1 = (uint32_t)(vbits >> 32);
    const unsigned voff = (unsigned)(((int64_t)rc * x_stride + (int64_t)j * plane_stride) * 4);
    const unsigned moff = ARR ? (unsigned)((int64_t)rc * m_x_stride + (int64_t)j * m_plane_stride) : 0u;
    const uint64_t step = (uint64_t)(L * plane_stride) * 4u, mstep = ARR ? (uint64_t)(L * m_plane_stride) : 0u;
    const char* kb = reinterpret_cast<const char*>(cube + tile_off);          // slot base (DESC), advanced in scalar arithmetic
    const char* mb = ARR ? reinterpret_cast<const char*>(marr + m_tile_off) : nullptr;
    const float* p = cube + tile_off + (int64_t)rc * x_stride;                 // (64-bit form)
    const uint8_t* pm = ARR ? marr + m_tile_off + (int64_t)rc * m_x_stride : nullptr;
#pragma unroll
    for (int i0 = 0; i0 < KPL; i0 += U) {
        unsigned mk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u;
            const bool chk = i >= kChk0;
            // all ones when the slot holds a sample of this lane
            const uint32_t vmu = chk ? (uint32_t)__builtin_amdgcn_sbfe((int)((i - kChk0) < 32 ? vb0 : vb1), (i - kChk0) & 31, 1) : 0xffffffffu;
            if (DESC) {
                if (i > 0) {
                    const bool adv = !chk || (i <= last);
                    kb += adv ? step : 0u;
                    if (ARR) mb += adv ? mstep : 0u;
                }
                const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, (int)0xffffffffu, 0x00020000);
                key[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(voff & vmu), 0, /*nt*/ 2);
                if (ARR) {
                    const auto rm = __builtin_amdgcn_make_buffer_rsrc((void*)mb, 0, (int)0xffffffffu, 0x00020000);
                    mk[u] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rm, (int)(moff & vmu), 0, 2);
                } else {
                    mk[u] = 1u;
                }
            } else {
                const int z = chk ? min(j + L * i, nz - 1) : j + L * i;
                key[i] = __float_as_uint(__builtin_nontemporal_load(p + (int64_t)z * plane_stride));
                mk[u] = ARR ? pm[(int64_t)z * m_plane_stride] : 1u;
            }
        }
        __builtin_amdgcn_sched_barrier(0);                       // (all loads of the group first)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u;
            const uint32_t raw = key[i];
            uint32_t k = fkey_bits(raw);
            // (beyond the last plane: the excluded key fails the range test)
            if (i >= kChk0) k |= ~(uint32_t)__builtin_amdgcn_sbfe((int)((i - kChk0) < 32 ? vb0 : vb1), (i - kChk0) & 31, 1);
            bool ok = (k - kmin) < span_l;
            if (ARR) ok = ok && (mk[u] != 0u);
            if (CEN) {
                // (an infinite raw under "no isfinite" passes and |inf - cen| stays a valid key unless cen is the same
                //  infinity: that NaN sorts above +inf)
                k = __float_as_uint(fabsf(__uint_as_float(raw) - cen)) | 0x80000000u;
                ok = ok && (k <= 0xff800000u);
            }
            key[i] = ok ? k : 0xffffffffu;
            mine += ok ? 1 : 0;
        }
        __builtin_amdgcn_sched_barrier(0);                       // U loads in flight, not KPL
    }
    return mine;
}

template <int TS, int KPL, bool ARR, bool DESC, int BT = 256>
__device__ __forceinline__ int sel_load_keys(const float* cube, int64_t plane_stride, int64_t x_stride, int64_t tile_off,
                                             const uint8_t* marr, int64_t m_plane_stride, int64_t m_x_stride, int64_t m_tile_off,
                                             int rc, int j, int nz, bool col_in, uint32_t kmin, uint32_t span1,
                                             bool use_cen, float cen, uint32_t (&key)[KPL]) {
    // (block-uniform branch; the memory clobbers keep the compiler from hoisting the - identical - loads of both arms above
    //  the branch, where all KPL of them would be in flight at once and spill)
    if (use_cen) {
        asm volatile("; keys of |x - centre|" ::: "memory");
        return sel_load_keys_impl<TS, KPL, ARR, DESC, true, BT>(cube, plane_stride, x_stride, tile_off, marr, m_plane_stride, m_x_stride,
                                                            m_tile_off, rc, j, nz, col_in, kmin, span1, cen, key);
    }
    asm volatile("; keys of x" ::: "memory");
    return sel_load_keys_impl<TS, KPL, ARR, DESC, false, BT>(cube, plane_stride, x_stride, tile_off, marr, m_plane_stride, m_x_stride,
                                                         m_tile_off, rc, j, nz, col_in, kmin, span1, 0.f, key);
}

// LDS scratch of one block of TS rays
template <int TS>
struct SelShared {
    uint32_t hist[3][TS][16];
    uint32_t nvalid[TS], nextkey[TS], ncand[TS], sel_key[TS], sel_below[TS], sel_eq[TS], nlow[TS];
    uint32_t cand[TS][kCandMax];
    int level;                                                  // (SelCache: the pass a descent resumes after)
    int vote[8];                                                // "a ray's bin is too large to rank" after pass p / [7]: at the resume
    int narrow;                                                 // a ray keeps > 3/4 of its samples in one leading digit: no early votes
};

// The histograms an earlier descent over the SAME keys counted in its first four passes (4, 8, 12, 16 key bits), per ray, each with
// the prefix it was counted under.  A later descent for another rank (the clip loop: the median moves by a few ranks per iteration)
// walks them instead of counting - pass p's cached histogram serves as long as the new path's prefix above it is the one it was
// counted under, also when the new rank picks ANOTHER digit from it - and starts counting at the first pass whose cached
// histogram belongs to another bin (block-wide: the earliest of its rays' answers).  `prefix / below / eq / krel`: the walk's state
// after each pass, handed from the one lane per ray that walks to the others.
template <int TS>
struct SelCache {
    uint16_t hist[4][TS][16];
    uint32_t pre[4][TS];
    uint32_t prefix[4][TS], below[4][TS], eq[4][TS], krel[4][TS];
    int first_done;                                             // the pass after which the FIRST descent ranked its candidates (block-wide)
};

// every thread of the block: zero what a descent needs (followed by a barrier at the caller)
template <int TS, int BT = 256>
__device__ __forceinline__ void sel_reset(SelShared<TS>& S) {
    const int t = threadIdx.x;
    if (t < TS) { S.nvalid[t] = 0u; S.nextkey[t] = 0xffffffffu; S.ncand[t] = 0u; S.nlow[t] = 0u; }
    if (t == 0) S.level = 3;
    if (t < 8) S.vote[t] = 0;
    if (t == 8) S.narrow = 0;
    for (int i = t; i < 3 * TS * 16; i += BT) (&S.hist[0][0][0])[i] = 0u;
}

// 16 bits of the key known: few samples are left in the bin.  All rays of the block small enough -> rank them directly
// (k: rank inside the bin; on success prefix / below / eq describe the selected KEY).  Block-uniform result.
template <int TS, int KPL, int BT, class XF>
__device__ __forceinline__ bool sel_rank_candidates(SelShared<TS>& S, const uint32_t (&key)[KPL], const XF& xf, int r, int j, int n, int k,
                                                    uint32_t& prefix, int& below, int& eq, int cshift = 16, int cap = kCandMax, int slot = 7) {
    constexpr int kLanesPerRay = BT / TS;
    // the vote: one flag word per asking point of a descent (zeroed by sel_reset), one barrier
    if ((n > 0) & (eq > cap) & (j == 0)) S.vote[slot] = 1;
    __syncthreads();
    if (S.vote[slot] != 0) return false;
    if (n > 0) {
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const uint32_t ki = xf(key[i]);
            if (((ki ^ prefix) >> cshift) == 0u) {
                const uint32_t slot = atomicAdd(&S.ncand[r], 1u);
                S.cand[r][slot] = ki;
            }
        }
    }
    __syncthreads();
    if (n > 0) {
        for (int idx = j; idx < eq; idx += kLanesPerRay) {
            const uint32_t c = S.cand[r][idx];
            int less = 0, same_before = 0, same = 0;
            for (int m = 0; m < eq; ++m) {
                const uint32_t o = S.cand[r][m];
                less += (o < c) ? 1 : 0;
                same += (o == c) ? 1 : 0;
                same_before += (o == c && m < idx) ? 1 : 0;
            }
            if (less + same_before == k) { S.sel_key[r] = c; S.sel_below[r] = (uint32_t)(below + less); S.sel_eq[r] = (uint32_t)same; }
        }
    }
    __syncthreads();
    if (n > 0) { prefix = S.sel_key[r]; below = (int)S.sel_below[r]; eq = (int)S.sel_eq[r]; }
    return true;
}

// The descent over the registers of the block (all threads call it; barriers inside).  r / j: ray and slice of this
// lane, n: valid samples of the ray, rank0: samples of the key set that sort below them (the clip loop keeps its keys
// and moves a window over them).  Returns the keys of the two order statistics numpy's 'linear' percentile q
// interpolates between and the interpolation fraction.  S must have been reset (sel_reset + barrier).
// C: cache of this key set's descents (filled; consulted when `resume`, block-uniform), or nullptr.
template <int TS, int KPL, int BT = 256, class XF = KeyIdentity>
__device__ __forceinline__ void ray_select(SelShared<TS>& S, const uint32_t (&key)[KPL], const XF& xf, int r, int j, int n, double q,
                                           uint32_t& key_lo, uint32_t& key_hi, double& frac, int max_pass = 8, int rank0 = 0,
                                           SelCache<TS>* C = nullptr, bool resume = false, bool early = true) {
    const double pos = q / 100.0 * (double)(n > 0 ? n - 1 : 0);
    const double fl = floor(pos);
    int k = rank0 + (int)fl;                                     // rank still to be found inside the current prefix
    const int khi = rank0 + min((int)ceil(pos), max(n - 1, 0));
    frac = pos - fl;
    int below = 0;                                               // keys smaller than everything matching the prefix
    uint32_t prefix = 0u;
    int eq = 0;
    bool done = false;                                           // block-uniform
    int start = 0;
    if (C != nullptr && resume) {
        if (j == 0) {                                            // one lane per ray walks the cached histograms
            int lvl = 3;                                         // (rays without samples do not care)
            if (n > 0) {
                lvl = -1;
                uint32_t pf = 0u;
                int bl = 0, kk = k;
                bool live = true;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    live = live && (p == 0 || pf == C->pre[p][r]);
                    if (live) {
                        uint32_t dsel = 15u;
                        int e = 0;
                        bool found = false;
#pragma unroll
                        for (int d = 0; d < 16; ++d) {
                            const int c = (int)C->hist[p][r][d];
                            if (!found) {
                                if (kk < c) { dsel = d; found = true; e = c; }
                                else { kk -= c; bl += c; }
                            }
                        }
                        pf |= dsel << (28 - 4 * p);
                        C->prefix[p][r] = pf; C->below[p][r] = (uint32_t)bl; C->eq[p][r] = (uint32_t)e; C->krel[p][r] = (uint32_t)kk;
                        lvl = p;
                    }
                }
            }
            if (lvl < 3) atomicMin(&S.level, lvl);
        }
        __syncthreads();
        const int sl = S.level;
        if (sl >= 0) {
            prefix = C->prefix[sl][r];
            below = (int)C->below[sl][r];
            eq = (int)C->eq[sl][r];
            k = (int)C->krel[sl][r];
            start = sl + 1;
        }
    }
    // Ranking eq candidates costs a lane eq^2 / (lanes per ray) compares, a pass ~5 slots for each of its KPL keys: before 16 bits
    // are known the vote only passes where ranking is the cheaper of the two - 32 candidates with 16 lanes per ray, 11 with 2.
    constexpr int kL = BT / TS;
    constexpr int kEarlyCap = kL >= 16 ? 32 : (kL >= 8 ? 22 : (kL >= 4 ? 16 : 11));
    if (C != nullptr && !resume && j == 0) {
        // a first descent may stop before its fourth pass: what it does not count must not look like a cached histogram
        C->pre[1][r] = 0xffffffffu; C->pre[2][r] = 0xffffffffu; C->pre[3][r] = 0xffffffffu;
    }
    // A vote that fails costs a barrier: a resumed descent (the clip loop: the same keys, a window that moved by a few ranks) only asks
    // where the first descent of the block was answered - its bins are the same size give or take the clipped samples.
    const int vote_from = (C != nullptr && resume) ? min(C->first_done, 3) : (early ? SPC_SEL_FIRST_VOTE : 3);     // block-uniform
    int done_pass = 3;
    // resumed with enough bits known: the candidates of the bin straight away when every ray's bin is small (no pass runs)
    if (start >= 2 && start - 1 >= vote_from)
        done = sel_rank_candidates<TS, KPL, BT>(S, key, xf, r, j, n, k, prefix, below, eq, 32 - 4 * start, start < 4 ? kEarlyCap : kCandMax);
#pragma unroll 1
    for (int pass = start; pass < max_pass && !done; ++pass) {
        const int b = 28 - 4 * pass;
        unsigned long long accE = 0ull, accO = 0ull;             // even / odd digits, 8 bits each (<= 64 keys per lane)
        // (the first pass has no prefix; `pass` is made opaque so that its digit extraction - which does not depend on
        // anything the loop changes - is not hoisted out of the loop and kept in 64 more registers)
        int popaque = pass;
        asm volatile("" : "+s"(popaque));
        if (popaque == 0) count_keys<KPL, true>(key, prefix, b, xf, accE, accO);
        else count_keys<KPL, false>(key, prefix, b, xf, accE, accO);
        uint32_t* h = S.hist[pass % 3][r];
        uint32_t* hz = S.hist[(pass + 1) % 3][r];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const uint32_t c0 = (uint32_t)(accE >> (8 * d)) & 0xffu, c1 = (uint32_t)(accO >> (8 * d)) & 0xffu;
            if (c0) atomicAdd(&h[2 * d], c0);
            if (c1) atomicAdd(&h[2 * d + 1], c1);
        }
        if (j == 0) {
#pragma unroll
            for (int d = 0; d < 16; ++d) hz[d] = 0u;              // next pass's buffer (last read two passes ago)
        }
        __syncthreads();
        uint32_t dsel = 15u;
        bool found = false;
        const bool keep = C != nullptr && pass < 4 && j == 0;    // this pass's histogram and the prefix it was counted under
        if (keep) C->pre[pass][r] = prefix;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const int c = (int)h[d];
            if (keep) C->hist[pass][r][d] = (uint16_t)c;
            if (!found) {
                if (k < c) { dsel = d; found = true; eq = c; }
                else { k -= c; below += c; }
            }
        }
        prefix |= dsel << b;
        // (a ray that keeps more than 3/4 of its samples in one leading digit - positive data, a narrow relative range - will not have
        //  a small bin two passes later: the early votes of the whole block are dropped.  Written here, read behind the next pass's
        //  barrier.)
        if (pass == 0 && j == 0 && n > 0 && 4 * eq > 3 * n) S.narrow = 1;
        // Few samples are left in the bin: rank its candidates directly when every ray's bin is small, and ask again after every
        // further pass (data in a narrow relative range share their leading bits: their bins get small two passes later).  Round 4:
        // asked from the second pass on - a failed vote costs one barrier, and samples around zero (baseline-subtracted spectra)
        // have few keys per binade near their median: 8 or 12 bits leave <= 32 (median 1024^3: 1.86 -> 1.64 ms).
        if (pass >= vote_from && pass < 7 && (pass >= 3 || S.narrow == 0)) {
            done = sel_rank_candidates<TS, KPL, BT>(S, key, xf, r, j, n, k, prefix, below, eq, 28 - 4 * pass, pass < 3 ? kEarlyCap : kCandMax, pass);
            if (done) done_pass = min(pass, 3);
        }
    }
    if (C != nullptr && !resume && threadIdx.x == 0) C->first_done = done_pass;
    // prefix = the key of rank floor(pos); `eq` samples carry it, `below` are smaller
    key_lo = prefix;
    key_hi = prefix;
    const bool need_next = (n > 0) && (khi >= below + eq);       // ray-uniform
    if (__syncthreads_or(need_next ? 1 : 0)) {
        if (need_next) {
            uint32_t mn = 0xffffffffu;
#pragma unroll
            for (int i = 0; i < KPL; ++i) { const uint32_t ki = xf(key[i]); mn = (ki > prefix) ? min(mn, ki) : mn; }
            atomicMin(&S.nextkey[r], mn);
        }
        __syncthreads();
        if (need_next) key_hi = S.nextkey[r];
    }
}

// numpy's interpolation between the two order statistics (the mean of two for a median), in float64, rounded once
__device__ __forceinline__ float sel_value(uint32_t key_lo, uint32_t key_hi, double frac, double scale, bool median = true) {
    const double a = (double)funkey(key_lo), bb = (double)funkey(key_hi);
    // (np.nanmedian of an odd count is the middle sample itself - also an infinity; np.nanpercentile's lerp a + (b - a) t
    //  between two equal infinities is NaN, and so is it here)
    return (float)(((median && a == bb) ? a : (frac == 0.5 ? 0.5 * (a + bb) : a + (bb - a) * frac)) * scale);
}

// ---- rays with few valid samples (round 5) ---------------------------------------------------------------
// A signal mask (data > n sigma on narrow lines: the SURVEY's mask rule leaves 4 - 5 % of a cube) keeps a few dozen of a
// ray's 1024 samples, yet every digit pass and every statistics sweep walked all KPL key registers of every lane.  When NO
// ray of the block holds more than kCompactKeys x (lanes per ray) valid samples, the valid keys of a ray are packed into LDS -
// in sample order: per-lane counts, an exclusive prefix over the lanes of the ray, every lane writes its keys behind its
// prefix, so the packed order does not depend on scheduling - and dealt out again, kCompactKeys per lane; the descent and the
// clip loop then run on those instead of on KPL registers.  The packed set holds the same keys: the selection is identical, the
// clip loop's float64 sums are taken in another (fixed) order.
constexpr int kCompactKeys = 8;
template <int TS, int BT>
struct SelCompact {
    uint32_t keys[TS][kCompactKeys * (BT / TS)];       // a ray's valid keys in sample order
    uint32_t sorted[TS][kCompactKeys * (BT / TS)];     // ... ascending (clip_packed_waves)
    double ps[TS][BT / TS], pq[TS][BT / TS];           // lane partials of a ray's sums (clip_packed_waves)
    int cnt[TS][BT / TS];
    int tot[TS];
    uint32_t win[TS][2];
    int any_big;
};
template <int KPL, int TS, int BT>
constexpr bool sel_compactable() { return KPL >= 4 * kCompactKeys && (BT / TS) >= 8; }

// all threads of the block; true: ck holds this lane's share of the ray's n valid keys (block-uniform result)
template <int TS, int KPL, int BT>
__device__ __forceinline__ bool sel_compact(SelCompact<TS, BT>& Q, const uint32_t (&key)[KPL], int mine, int r, int j,
                                            uint32_t (&ck)[kCompactKeys], int& n) {
    constexpr int L = BT / TS, CAP = kCompactKeys * L;
    Q.cnt[r][j] = mine;
    if (threadIdx.x == 0) Q.any_big = 0;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll 4
    for (int l = 0; l < L; ++l) {
        const int c = Q.cnt[r][l];
        base += (l < j) ? c : 0;
        tot += c;
    }
    n = tot;
    if (j == 0) Q.tot[r] = tot;
    if (tot > CAP && j == 0) Q.any_big = 1;
    __syncthreads();
    if (Q.any_big) return false;
    int idx = base;
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
        if (key[i] != 0xffffffffu) { Q.keys[r][idx] = key[i]; ++idx; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kCompactKeys; ++i) {
        const int sidx = j + L * i;
        ck[i] = sidx < tot ? Q.keys[r][sidx] : 0xffffffffu;
    }
    return true;
}

// (LDS traffic between the lanes of ONE wave: the LDS serves a wave's operations in order, the fence keeps the compiler
//  from reordering them)
__device__ __forceinline__ void sel_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int TS, int KPL, bool ARR, bool DESC, int BT = 256>
// (blocks per CU the registers allow, as the compiler reports them - round-5 verdict: twelve instantiations asked for five and
//  got three; the bound now names what is reached: 128-ray tiles keep more state per lane)
__global__ __launch_bounds__(BT, BT == 256 ? (KPL == 128 ? 2 : (TS == 128 ? 3 : 5)) : (BT == 512 ? 4 : 1)) void select_reg_kernel(const SelArgs A) {
    __shared__ SelShared<TS> S;
    constexpr int kLanesPerRay = BT / TS;
    const int t = threadIdx.x;
    // ray of the tile, slice of the ray: adjacent lanes hold adjacent spaxels - or, for rays along x, adjacent samples
    const int r = A.along_ray ? t / kLanesPerRay : t % TS, j = A.along_ray ? t % kLanesPerRay : t / TS;
    const int64_t tiles_x = (A.nx + TS - 1) / TS;
    const int64_t tile = sel_tile_of_block(blockIdx.x, gridDim.x, A.xcd_group);
    const int64_t y = tile / tiles_x, x0 = (tile % tiles_x) * TS;
    const int nz = (int)A.nz;
    const bool col_in = x0 + r < A.nx;
    const bool use_cen = A.center != nullptr;
    const float cen = (use_cen && col_in) ? A.center[y * A.nx + x0 + r] : 0.f;
    sel_reset<TS, BT>(S);
    __syncthreads();
    // ---- the one read of the cube: keys into registers (excluded / NaN / beyond nz: the largest key)
    uint32_t key[KPL];
    const int rc = col_in ? r : (int)(A.nx - 1 - x0);
    const int64_t tile_off = y * A.row_stride + x0 * A.x_stride, m_tile_off = ARR ? y * A.mask.row_stride + x0 * A.m_x_stride : 0;
    const int mine = sel_load_keys<TS, KPL, ARR, DESC, BT>(A.cube, A.plane_stride, A.x_stride, tile_off, A.mask.arr, A.mask.plane_stride, A.m_x_stride,
                                                 m_tile_off, rc, j, nz, col_in, A.key_min, A.key_span, use_cen, cen, key);
    if (mine) atomicAdd(&S.nvalid[r], (uint32_t)mine);
    __syncthreads();
    const int n = (int)S.nvalid[r];
    uint32_t key_lo, key_hi;
    double frac;
#ifdef SPC_ABLATE
    if (A.ablate == 1) {
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) x ^= key[i];
        if (j == 0 && col_in) A.out[y * A.nx + x0 + r] = __uint_as_float(x + n);
        return;
    }
#endif
#ifdef SPC_ABLATE
    ray_select<TS, KPL, BT>(S, key, KeyIdentity{}, r, j, n, A.q, key_lo, key_hi, frac, A.ablate >= 2 ? A.ablate - 1 : 8);
#else
    ray_select<TS, KPL, BT>(S, key, KeyIdentity{}, r, j, n, A.q, key_lo, key_hi, frac);
#endif
    if (j == 0 && col_in) A.out[y * A.nx + x0 + r] = n > 0 ? sel_value(key_lo, key_hi, frac, (double)A.scale, A.q == 50.0) : NAN;
}

// ---- sigma clipping with the rays resident in registers -------------------------------------------------
// astropy.stats.sigma_clip(axis=0, masked=False) iterates centre / spread / clip over the cube; built from separate
// kernels every iteration reads it three times and writes it once (ops.sigma_clip_axis0: ~20 passes for the 5
// default iterations).  Here a block loads its TS rays once (the selection kernel's layout), iterates entirely in
// registers - valid count, sum and sum of squares per ray (float64, lane partials added in a fixed order: the result
// does not depend on scheduling), the median through ray_select, the bounds with spc_clip_bounds_f32's arithmetic,
// clipped samples turned into the "excluded" key - until no ray of the block changes or maxiters is reached, and
// writes the clipped rays once.  A ray that has converged recomputes the same bounds and clips nothing, so stopping per
// block equals astropy's global "until nothing changes".
struct ClipRegArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    float* out;                 // (nz, ny, nx) C-contiguous
    double lo_s, hi_s;
    int maxiters;               // < 0: until convergence
    int cen_mean;               // centre: 0 median, 1 mean
    int spread_mad;             // spread: 0 std, 1 mad_std
    uint32_t key_min, key_span;            // sel_key_range of the mask's predicate terms (filled by the launcher)
    int xcd_group;              // tiles of a row kept on one XCD (sel_tile_of_block)
    int compact;                // rays with few valid samples are packed (sel_compact; SPC_SELECT_COMPACT=0: off)
    const int* probe;           // NULL, or the largest valid count among the sampled rays (clip_probe_kernel): which block shape runs
};

// |x - centre| of the samples inside the clip window (the MAD's keys)
struct KeyAbsDevWin {
    float center;
    uint32_t wlo, wspan;
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const {
        const float v = fabsf(funkey(k) - center);
        return ((k - wlo) < wspan && v == v) ? fkey(v) : 0xffffffffu;
    }
};

// the clip loop over one set of resident keys (all KPL registers of the lane, or its share of a packed sparse ray): narrows
// the window [wlo, wlo + wspan) of the keys that are kept
template <int TS, int KPL, bool MAD, int BT>
__device__ __forceinline__ void clip_iterate(const ClipRegArgs& A, SelShared<TS>& S, SelCache<TS>& C, double (*part_s)[BT / TS],
                                             double (*part_q)[BT / TS], uint32_t (&key)[KPL], int r, int j, uint32_t& wlo, uint32_t& wspan) {
    constexpr int kLanesPerRay = BT / TS;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    // The keys stay as loaded.  Clipping v < lo || v > hi removes the two ENDS of a ray's sorted samples, so what is left
    // is a window of keys [wlo, wlo + wspan): an iteration counts and sums the samples inside it, asks the resident key
    // set for the rank (samples below the window) + (n - 1) / 2 - a descent that resumes from the cached bins of the
    // earlier iterations' descents, the median moving by a few ranks only (round 3a re-ran all four passes over keys it
    // had overwritten: 1.9 ms per iteration at 1024^3) - and narrows the window.
    int n_prev = -1;
    bool cached = false;                                         // block-uniform: C holds an earlier iteration's descent
#pragma unroll 1
    for (int it = 0; A.maxiters < 0 || it < A.maxiters; ++it) {
        // (the keys never change, so everything an iteration derives from them alone - 64 floats, doubles, digits - would be
        //  hoisted out of the loop and spilled: they are made opaque once per iteration, which costs no instruction)
#pragma unroll
        for (int i = 0; i < KPL; ++i) asm volatile("" : "+v"(key[i]));
        // ---- valid count, sum, sum of squares of the ray (what spc_stats_axis_f32 gives the unfused path)
        // (the previous iteration's descent ends with lanes reading S.nextkey[r] behind its last barrier: without this barrier a
        //  wave that is already here resets the word under them - the upper of an even count's two middle samples came back as
        //  the "excluded" key in one launch out of four, tools/stress_clip_determinism.py)
        if (it > 0) __syncthreads();
        sel_reset<TS, BT>(S);
        int cnt = 0, low = 0;
        double s = 0.0, ss = 0.0;
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const bool ok = (key[i] - wlo) < wspan;
            const double v = ok ? (double)funkey(key[i]) : 0.0;
            cnt += ok ? 1 : 0;
            low += (key[i] < wlo) ? 1 : 0;
            s += v;
            ss = fma(v, v, ss);
            if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
        }
        part_s[r][j] = s;
        part_q[r][j] = ss;
        __syncthreads();                                         // (also orders sel_reset before the counts)
        if (cnt) atomicAdd(&S.nvalid[r], (uint32_t)cnt);
        if (low) atomicAdd(&S.nlow[r], (uint32_t)low);
        double sum = 0.0, ssq = 0.0;
#pragma unroll 4
        for (int l = 0; l < kLanesPerRay; ++l) { sum += part_s[r][l]; ssq += part_q[r][l]; }
        __syncthreads();
        const int n = (int)S.nvalid[r], nl = (int)S.nlow[r];
        // (the previous iteration clipped nothing anywhere in the block: astropy stops there)
        if (it > 0 && !__syncthreads_or(n != n_prev ? 1 : 0)) break;
        n_prev = n;
        double mean = nan, sd = nan;
        if (n > 0) {
            mean = sum / (double)n;
            const double var = __dsub_rn(ssq / (double)n, __dmul_rn(mean, mean));
            sd = (var != var) ? nan : sqrt(var > 0.0 ? var : 0.0);
        }
        double cen = mean;
        float med = NAN;
#ifdef SPC_ABLATE
        if (A.xcd_group >= 1000 && it > 0) {                     // timing only: later iterations without their descent / without resuming
            if (A.xcd_group == 1000) { med = (float)mean; if (!A.cen_mean) cen = mean; }
            else {
                uint32_t key_lo, key_hi;
                double frac;
                ray_select<TS, KPL, BT>(S, key, KeyIdentity{}, r, j, n, 50.0, key_lo, key_hi, frac, 8, nl, &C, false);
                med = n > 0 ? sel_value(key_lo, key_hi, frac, 1.0) : NAN;
                if (!A.cen_mean) cen = (double)med;
            }
        } else
#endif
        if (!A.cen_mean || MAD) {
            uint32_t key_lo, key_hi;
            double frac;
            ray_select<TS, KPL, BT>(S, key, KeyIdentity{}, r, j, n, 50.0, key_lo, key_hi, frac, 8, nl, &C, cached);
            cached = true;
            med = n > 0 ? sel_value(key_lo, key_hi, frac, 1.0) : NAN;
            if (!A.cen_mean) cen = (double)med;
        }
        if (MAD) {
            // spread = 1.4826 x the median of |x - median| (astropy mad_std; float32 deviations like numpy's, the scale
            // in float32 like spc_percentile_axis0_f32's argument): a second descent over the transformed keys
            const KeyAbsDevWin xf{med, wlo, wspan};
            __syncthreads();                                     // the first descent's last reads of S
            sel_reset<TS, BT>(S);
            __syncthreads();
            int nm = n;
            if (__syncthreads_or((n > 0 && !(fabsf(med) <= 3.4028234664e38f)) ? 1 : 0)) {
                // an infinite median: inf - inf deviations are NaN and drop out of the ray
                int c = 0;
#pragma unroll
                for (int i = 0; i < KPL; ++i) c += (xf(key[i]) != 0xffffffffu) ? 1 : 0;
                if (c) atomicAdd(&S.nvalid[r], (uint32_t)c);
                __syncthreads();
                nm = (int)S.nvalid[r];
            }
            uint32_t key_lo, key_hi;
            double frac;
            ray_select<TS, KPL, BT>(S, key, xf, r, j, nm, 50.0, key_lo, key_hi, frac, 8, 0, nullptr, false, /*early*/ false);
            const float spread = nm > 0 ? sel_value(key_lo, key_hi, frac, (double)1.482602218505602f) : NAN;
            sd = (double)spread;
            if (A.cen_mean) cen = (double)(float)mean;           // (the loop of separate kernels hands a float32 mean map over)
        }
        const float lo = (float)__dsub_rn(cen, __dmul_rn(A.lo_s, sd));
        const float hi = (float)__dadd_rn(cen, __dmul_rn(A.hi_s, sd));
        // v < lo  <=>  key < key(lo), v > hi  <=>  key > key(hi); a bound of either zero keeps both zeros; NaN bounds clip nothing
        if (wspan != 0u) {
            const uint32_t klo = (lo == lo) ? fkey(lo == 0.f ? -0.f : lo) : 0u;
            const uint32_t khi = (hi == hi) ? fkey(hi == 0.f ? 0.f : hi) : 0xff800000u;
            const uint32_t nlo = max(wlo, klo), nhi = min(wlo + (wspan - 1u), khi);
            wlo = nlo;
            wspan = nhi >= nlo ? nhi - nlo + 1u : 0u;
        }
    }
}

// The clip loop over PACKED rays (sel_compact: no ray of the block holds more than 128 valid samples), every wave on its own:
// what an iteration costs on 8 keys per lane is not arithmetic but the block's rendezvous - ~20 barriers per iteration, three
// LDS histogram rounds per digit pass: 0.75 ms per iteration at 1024^3 whether a lane walks 64 keys or 8.  With 16 lanes per ray
// a wave takes four whole rays instead: it sorts each once (every lane ranks its <= 8 keys against the ray's, ties by sample
// index - n^2 / 16 compares per lane, n ~ 50 under a signal mask), after which the median of any clip window is two LDS reads
// (the window removes the two ends of the sorted keys), counts come from a 16-lane butterfly and the float64 sums from lane
// partials added in a fixed order.  No barrier, no atomics, no descent; a wave stops when none of ITS rays changes.  The
// arithmetic of the bounds is clip_iterate's (astropy's): the same windows up to the order of the float64 additions.
// REPRODUCIBILITY (round-5 advisor): which of the two loops a ray takes depends on ITS BLOCK (every ray of the block sparse) and
// on the 64-ray probe that picks the block shape, so the float64 mean / std of one spectrum can be added up in two different
// (each fixed) orders depending on the cube's width or on how it was sharded: the bounds then differ in their last bits and a
// sample that sits within 1e-16 relative of a bound can be clipped in one run and kept in the other.  Launch-to-launch results
// on the SAME cube are bit-identical (tests/test_gpu_fullsize.py); sigma_clip_spectrally is NOT among the operators whose
// sharded result is bit-identical to the unsharded one (DESIGN section 7 lists them) - like astropy's own result, which
// moves in the same way with numpy's pairwise summation block size.
// (its own function, not inlined: in line, the sort's registers on top of the 64 resident keys made the allocator spill 57 VGPRs and
//  the DENSE loop of the same kernel went from 7.7 to 9.4 ms)
struct ClipScalars { double lo_s, hi_s; int maxiters, cen_mean; };
template <int TS, int BT>
__device__ __attribute__((noinline)) void clip_packed_waves(const ClipScalars A, SelCompact<TS, BT> __attribute__((address_space(3)))* Qp) {
    static_assert(BT / TS == 16, "16 lanes per ray: four rays per wave");
    auto& Q = *Qp;                                                   // (an LDS pointer: ds_read / ds_write, not flat accesses)
    constexpr int KC = kCompactKeys;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rr = 4 * w + (lane >> 4), jl = lane & 15;
    const int n_all = Q.tot[rr];
    const int nmax = max(max(__builtin_amdgcn_readlane(n_all, 0), __builtin_amdgcn_readlane(n_all, 16)),
                         max(__builtin_amdgcn_readlane(n_all, 32), __builtin_amdgcn_readlane(n_all, 48)));
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    // ---- sort: position of a key = keys below it (ties: earlier samples first).  Candidates four at a time (one 16-byte LDS
    // read), two reads in flight; slots past a ray's count compare as the largest key.
    unsigned long long mine64[KC];
    int pos[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) {
        const int idx = jl + 16 * i;
        const uint32_t k = idx < n_all ? Q.keys[rr][idx] : 0xffffffffu;
        mine64[i] = ((unsigned long long)k << 32) | (unsigned)idx;
        pos[i] = 0;
    }
    const u32x4 __attribute__((address_space(3)))* cand = reinterpret_cast<const u32x4 __attribute__((address_space(3)))*>(&Q.keys[rr][0]);
    for (int m0 = 0; m0 < nmax; m0 += 8) {
        const u32x4 c0 = cand[m0 / 4], c1 = cand[m0 / 4 + 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + u;
            const uint32_t raw = u < 4 ? c0[u & 3] : c1[u & 3];
            const uint32_t o = m < n_all ? raw : 0xffffffffu;
            const unsigned long long o64 = ((unsigned long long)o << 32) | (unsigned)m;
#pragma unroll
            for (int i = 0; i < KC; ++i)
                if (16 * i < nmax) pos[i] += (o64 < mine64[i]) ? 1 : 0;      // (wave-uniform guard)
        }
    }
#pragma unroll
    for (int i = 0; i < KC; ++i)
        if (jl + 16 * i < n_all) Q.sorted[rr][pos[i]] = (uint32_t)(mine64[i] >> 32);
    sel_wave_sync();
    uint32_t sk[KC];
#pragma unroll
    for (int i = 0; i < KC; ++i) sk[i] = (jl + 16 * i < n_all) ? Q.sorted[rr][jl + 16 * i] : 0xffffffffu;
    // ---- iterate
    uint32_t wlo = 0u, wspan = 0xff800001u;
    int n_prev = -1;
#pragma unroll 1
    for (int it = 0; A.maxiters < 0 || it < A.maxiters; ++it) {
        int cnt = 0, low = 0;
        double s = 0.0, ss = 0.0;
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            if (16 * i < nmax) {
                const bool ok = (sk[i] - wlo) < wspan;
                const double v = ok ? (double)funkey(sk[i]) : 0.0;
                cnt += ok ? 1 : 0;
                low += (sk[i] < wlo) ? 1 : 0;
                s += v;
                ss = fma(v, v, ss);
            }
        }
        Q.ps[rr][jl] = s;
        Q.pq[rr][jl] = ss;
        int both = cnt | (low << 16);                                   // (<= 128 each)
        both += __builtin_amdgcn_ds_swizzle(both, 0x041f);               // butterfly over the 16 lanes of the ray: xor 1, 2, 4, 8
        both += __builtin_amdgcn_ds_swizzle(both, 0x081f);
        both += __builtin_amdgcn_ds_swizzle(both, 0x101f);
        both += __builtin_amdgcn_ds_swizzle(both, 0x201f);
        sel_wave_sync();
        double sum = 0.0, ssq = 0.0;
#pragma unroll
        for (int l = 0; l < 16; ++l) { sum += Q.ps[rr][l]; ssq += Q.pq[rr][l]; }
        sel_wave_sync();
        const int n = both & 0xffff, nl = both >> 16;
        if (it > 0 && __builtin_amdgcn_ballot_w64(n != n_prev) == 0ull) break;    // none of this wave's rays lost a sample
        n_prev = n;
        double mean = nan, sd = nan;
        if (n > 0) {
            mean = sum / (double)n;
            const double var = __dsub_rn(ssq / (double)n, __dmul_rn(mean, mean));
            sd = (var != var) ? nan : sqrt(var > 0.0 ? var : 0.0);
        }
        double cen = mean;
        if (!A.cen_mean) {
            float med = NAN;
            if (n > 0) {
                const double p = 0.5 * (double)(n - 1), fl = floor(p);
                const uint32_t key_lo = Q.sorted[rr][nl + (int)fl], key_hi = Q.sorted[rr][nl + min((int)ceil(p), n - 1)];
                med = sel_value(key_lo, key_hi, p - fl, 1.0);
            }
            cen = (double)med;
        }
        const float lo = (float)__dsub_rn(cen, __dmul_rn(A.lo_s, sd));
        const float hi = (float)__dadd_rn(cen, __dmul_rn(A.hi_s, sd));
        if (wspan != 0u) {
            const uint32_t klo = (lo == lo) ? fkey(lo == 0.f ? -0.f : lo) : 0u;
            const uint32_t khi = (hi == hi) ? fkey(hi == 0.f ? 0.f : hi) : 0xff800000u;
            const uint32_t nlo = max(wlo, klo), nhi = min(wlo + (wspan - 1u), khi);
            wlo = nlo;
            wspan = nhi >= nlo ? nhi - nlo + 1u : 0u;
        }
    }
    if (jl == 0) { Q.win[rr][0] = wlo; Q.win[rr][1] = wspan; }
}

// Which block shape suits the cube is a property of its mask: packed rays (<= 128 valid samples each) iterate for free, so the
// 512-thread blocks' wider runs per plane win (read + write floor 2.4 against 3.3 ms at 1024^3); rays that stay in the registers
// iterate faster in 256-thread blocks.  The host cannot know without waiting for the device: a probe kernel counts the valid
// samples of 64 rays spread over the map and leaves the largest count in the caller's workspace, BOTH shapes are launched, and
// the blocks of the one the count does not select retire at once (an empty grid of 65536 blocks: ~20 us).
constexpr int kProbeRays = 64, kProbeSparse = 96;
template <bool ARR>
__global__ __launch_bounds__(256) void clip_probe_kernel(const ClipRegArgs A, int* d_max) {
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const int64_t nsp = A.ny * A.nx;
    const int64_t sp = min(nsp - 1, (int64_t)blockIdx.x * (nsp / kProbeRays) + nsp / (2 * kProbeRays));
    const int64_t y = sp / A.nx, x = sp - y * A.nx;
    int c = 0;
    for (int64_t z = threadIdx.x; z < A.nz; z += 256) {
        const uint32_t k = fkey_bits(__float_as_uint(A.cube[z * A.plane_stride + y * A.row_stride + x]));
        bool ok = (k - A.key_min) < A.key_span;
        if (ARR) ok = ok && A.mask.arr[z * A.mask.plane_stride + y * A.mask.row_stride + x] != 0;
        c += ok ? 1 : 0;
    }
    if (c) atomicAdd(&total, c);
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(d_max, total);
}

template <int TS, int KPL, bool ARR, bool MAD, bool DESC, int BT = 256>
// (blocks per CU: what the registers of each shape allow, as the compiler reports it - the instantiations that asked for four
//  and reached two or three now say so; the dense-mask shape <16, 64> reaches its four)
__global__ __launch_bounds__(BT, KPL == 128 ? 1 : (MAD ? 2 : (BT != 256 ? 4 : (TS == 128 ? 2 : (TS == 32 ? ((KPL == 64 && !DESC) ? 2 : 3) : 4))))) void sigma_clip_reg_kernel(const ClipRegArgs A) {
    if (A.probe != nullptr && ((*A.probe <= kProbeSparse) != (BT == 512))) return;   // the other block shape runs this cube
    __shared__ SelShared<TS> S;
    __shared__ SelCache<TS> C;
    constexpr int kLanesPerRay = BT / TS;
    __shared__ double part_s[TS][kLanesPerRay], part_q[TS][kLanesPerRay];
    const int t = threadIdx.x;
    const int r = t % TS, j = t / TS;
    const int64_t tiles_x = (A.nx + TS - 1) / TS;
    const int64_t tile = sel_tile_of_block(blockIdx.x, gridDim.x, A.xcd_group);
    const int64_t y = tile / tiles_x, x0 = (tile % tiles_x) * TS;
    const int nz = (int)A.nz;
    const bool col_in = x0 + r < A.nx;
    uint32_t key[KPL];
    const int mine = sel_load_keys<TS, KPL, ARR, DESC, BT>(A.cube, A.plane_stride, 1, y * A.row_stride + x0, A.mask.arr, A.mask.plane_stride, 1,
                                ARR ? y * A.mask.row_stride + x0 : 0, col_in ? r : (int)(A.nx - 1 - x0), j, nz,
                                col_in, A.key_min, A.key_span, false, 0.f, key);
    uint32_t wlo = 0u, wspan = 0xff800001u;                      // every valid key (the excluded one lies above)
    bool packed = false;
    if constexpr (sel_compactable<KPL, TS, BT>()) {
        // (a probe that met a ray of more than 128 valid samples: a dense cube, whose blocks need not even ask)
        if (A.compact && (A.probe == nullptr || *A.probe <= kCompactKeys * (BT / TS))) {
            __shared__ SelCompact<TS, BT> Q;
            uint32_t ck[kCompactKeys];
            int nc = 0;
            packed = sel_compact<TS, KPL, BT>(Q, key, mine, r, j, ck, nc);
            if (packed) {
                if constexpr (!MAD && BT / TS == 16) {
                    if (A.compact >= 2) {
                        clip_packed_waves<TS, BT>(ClipScalars{A.lo_s, A.hi_s, A.maxiters, A.cen_mean}, (SelCompact<TS, BT> __attribute__((address_space(3)))*)&Q);
                        __syncthreads();
                        wlo = Q.win[r][0];
                        wspan = Q.win[r][1];
                    } else {
                        clip_iterate<TS, kCompactKeys, MAD, BT>(A, S, C, part_s, part_q, ck, r, j, wlo, wspan);
                    }
                } else {
                    clip_iterate<TS, kCompactKeys, MAD, BT>(A, S, C, part_s, part_q, ck, r, j, wlo, wspan);
                }
            }
        }
    }
    if (!packed) clip_iterate<TS, KPL, MAD, BT>(A, S, C, part_s, part_q, key, r, j, wlo, wspan);
    (void)mine;
    // ---- the clipped rays, written once
    if (col_in) {
        float* q = A.out + y * A.nx + x0 + r + (int64_t)j * A.ny * A.nx;
        const int64_t qstep = (int64_t)kLanesPerRay * A.ny * A.nx;
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int z = j + kLanesPerRay * i;
            if (z < nz) *q = ((key[i] - wlo) < wspan) ? funkey(key[i]) : NAN;
            q += qstep;
            if (i % 8 == 7) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- whole-cube order statistic (median / percentile / mad_std with axis=None) ----------------
// The cube is one long ray here, so the digits of the key are found with shared histograms: one
// streaming pass per key BYTE (4 passes) counts, among the samples whose key matches the prefix
// found so far, the next byte; the host walks the 256 counters.  Histograms are built in LDS -
// 16 interleaved copies ([bin][lane & 15]: same-bin updates of different copies fall on different
// banks; real data put most samples in a handful of top-byte bins) - and flushed with one 64-bit
// atomic per bin and block.  A fifth pass (minimum key above the selected one) is only needed
// when the two order statistics of an interpolated percentile are different values.
struct GSelArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    int64_t nrows, rowlen, row_a, row_b;   // as in the statistics kernel: 1 row when contiguous
    uint32_t prefix, pmask;
    int shift;
    int has_center;
    float center;
    unsigned long long* hist;              // [256]
    uint32_t* next;                        // MODE 1: minimum key > prefix
};

template <bool ARR, int MODE>
__global__ __launch_bounds__(256) void gselect_kernel(const GSelArgs A) {
    __shared__ uint32_t lh[256 * 16];
    __shared__ uint32_t s_min;
    const int t = threadIdx.x;
    if (MODE == 0) {
        for (int i = t; i < 256 * 16; i += 256) lh[i] = 0;
    } else if (t == 0) {
        s_min = 0xffffffffu;
    }
    __syncthreads();
    const int copy = t & 15;
    uint32_t mymin = 0xffffffffu;
    auto take = [&](float v, unsigned mk) {
        const bool ok = spc_pred_valid(A.mask, v) & (mk != 0);
        if (!ok) return;
        if (A.has_center) v = __builtin_fabsf(v - A.center);
        const uint32_t k = fkey(v);
        if (MODE == 0) {
            if ((k & A.pmask) == A.prefix) atomicAdd(&lh[((k >> A.shift) & 0xffu) * 16 + copy], 1u);
        } else {
            if (k > A.prefix) mymin = min(mymin, k);
        }
    };
    // (one contiguous run: every block strides through it; rows of a strided view: the blocks share the rows out)
    const bool rows = A.nrows > 1;
    const int64_t stride = rows ? 256 : (int64_t)gridDim.x * 256, first = rows ? 0 : (int64_t)blockIdx.x * 256;
    for (int64_t r = rows ? blockIdx.x : 0; r < A.nrows; r += rows ? gridDim.x : 1) {
        const int64_t off = !rows ? 0 : (r / A.ny) * A.row_a + (r % A.ny) * A.row_b;
        const int64_t moff = !rows ? 0 : (r / A.ny) * A.mask.plane_stride + (r % A.ny) * A.mask.row_stride;
        const float* p = A.cube + off;
        const uint8_t* pm = ARR ? A.mask.arr + moff : nullptr;
        const bool al = ((((uintptr_t)p) & 15) == 0) && (!ARR || ((((uintptr_t)pm) & 3) == 0));
        const int64_t n4 = al ? A.rowlen / 4 : 0;
        for (int64_t i = first + t; i < n4; i += stride) {
            const f32x4s v = __builtin_nontemporal_load(reinterpret_cast<const f32x4s*>(p) + i);
            const uint32_t m = ARR ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pm) + i) : 0x01010101u;
#pragma unroll
            for (int c = 0; c < 4; ++c) take(v[c], (m >> (8 * c)) & 0xffu);
        }
        for (int64_t j = n4 * 4 + first + t; j < A.rowlen; j += stride) take(p[j], ARR ? pm[j] : 1u);
    }
    if (MODE == 0) {
        __syncthreads();
        unsigned long long tot = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) tot += lh[t * 16 + c];
        if (tot) atomicAdd(&A.hist[t], tot);
    } else {
        atomicMin(&s_min, mymin);
        __syncthreads();
        if (t == 0 && s_min != 0xffffffffu) atomicMin(A.next, s_min);
    }
}

}  // namespace

extern "C" int spc_percentile_axis0_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                        double q, const float* d_center, float scale, float* d_out) {
    int rc = spc_check_cube_any_order(cube);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    SPC_REQUIRE(q >= 0.0 && q <= 100.0, "Percentiles must be in the range [0, 100]");
    SelArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    sel_key_range(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, &A.key_min, &A.key_span);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.q = q; A.center = d_center; A.scale = scale; A.out = d_out;
    A.x_stride = 1; A.m_x_stride = 1; A.along_ray = 0;
    { const char* ab = getenv("SPC_SELECT_ABLATE"); A.ablate = ab ? atoi(ab) : 0; }
    // (16 neighbouring tiles of a row per XCD: a block's half - or quarter - of a 128-byte line is found in the L2 that fetched it
    //  for its neighbour.  1024^3: the clip kernel's read + write floor 4.21 -> 3.28 ms, the 256-thread selection 2.97 -> 2.65 ms;
    //  groups of 2 / 4 / 8: 3.95 / 3.54 / 3.32 - 3.6 ms; no effect on the 512-thread selection.  SPC_XCD_GROUP=0: block = tile)
    { const char* xg = getenv("SPC_XCD_GROUP"); A.xcd_group = xg ? atoi(xg) : 16; }
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const bool v4 = (cube->nx % 4 == 0) && (cube->row_stride % 4 == 0) && (cube->plane_stride % 4 == 0) &&
                    ((((uintptr_t)cube->d_data) & 15) == 0) &&
                    (!arr || ((A.mask.row_stride % 4 == 0) && (A.mask.plane_stride % 4 == 0) && ((((uintptr_t)A.mask.arr) & 3) == 0)));
    hipStream_t st = (hipStream_t)stream;
    // rays in registers (one read of the cube): up to 128 keys per lane, nz <= 512 / 1024 / 4096 for 32 / 16 / 8 spaxels
    // per block; any strides (a y-ray view included)
    const char* renv = getenv("SPC_SELECT_REG");
    if ((renv ? atoi(renv) != 0 : true) && cube->nz <= 4096 && cube->ny * ((cube->nx + 7) / 8) < (1LL << 31)) {
        // Blocks of 512 threads (round 3): the lanes of a wave cover 64 / 32 / 16 / 8 ADJACENT spaxels for rays of up to 512 /
        // 1024 / 2048 / 4096 samples - a plane's samples of a block are one 256 / 128 / 64 / 32-byte run.  With 256 threads
        // (16 spaxels, 64-byte runs at 1024 channels) the one read of the cube ran at 2.1 TB/s and made up 2.0 of the 2.9 ms
        // at 1024^3 (timing-only ablation, tools/bench_select_ablate.py); 32 spaxels: 0.92 ms.  SPC_SELECT_BT=256: the former table.
        {
            const char* be = getenv("SPC_SELECT_BT");
            // (rays of up to 256 samples keep the 256-thread table - 32 spaxels per block already, and the per-ray costs of the
            //  descent dominate: 2.26 against 2.46 ms at 256 x 2048 x 2048 - and so do rays above 2048 samples with a mask ARRAY,
            //  whose 8 spaxels per block make 8-byte runs of mask bytes either way: 4.1 against 5.1 ms at 4096 x 512 x 512)
            const bool keep256 = cube->nz <= 256 || (cube->nz > 2048 && arr);
            // Short rays (up to 256 samples): few lanes per ray.  What a pass costs beyond counting the keys - the lane's share
            // of the histogram atomics, the barrier, every lane's walk over the 16 totals - is paid per LANE, and with 8 lanes
            // on a ray of 100 samples it is most of the pass (0.8 - 1.1 ms per pass at 100 x 2048 x 4096 against 0.25 ms for
            // the same bytes in rays of 1024): 2 lanes per ray up to 128 samples, 4 up to 256, 64 keys each - and 128 / 64
            // adjacent spaxels per block.  SPC_SELECT_SHORT=0: the former table.
            if (cube->nz <= 256 && !(be && atoi(be) != 0) && spc_env_on("SPC_SELECT_SHORT")) {
                const int64_t nzr = cube->nz;
                const int tss = nzr <= 128 ? 128 : 64;
                const int kpls = nzr <= 32 ? 16 : (nzr <= 64 ? 32 : 64);
                const bool descs = sel_desc_fits(tss, std::max(cube->plane_stride, arr ? A.mask.plane_stride : 0), 1, 256) && spc_env_on("SPC_SELECT_DESC");
                dim3 grids((unsigned)(cube->ny * ((cube->nx + tss - 1) / tss)));
#define SPC_LAUNCH_SHORT(TS_, K_)                                                                                                    \
                do {                                                                                                                \
                    if (descs) { if (arr) hipLaunchKernelGGL((select_reg_kernel<TS_, K_, true, true, 256>), grids, dim3(256), 0, st, A);   \
                                 else hipLaunchKernelGGL((select_reg_kernel<TS_, K_, false, true, 256>), grids, dim3(256), 0, st, A); }    \
                    else { if (arr) hipLaunchKernelGGL((select_reg_kernel<TS_, K_, true, false, 256>), grids, dim3(256), 0, st, A);        \
                           else hipLaunchKernelGGL((select_reg_kernel<TS_, K_, false, false, 256>), grids, dim3(256), 0, st, A); }         \
                } while (0)
                if (tss == 64) SPC_LAUNCH_SHORT(64, 64);
                else if (kpls == 16) SPC_LAUNCH_SHORT(128, 16);
                else if (kpls == 32) SPC_LAUNCH_SHORT(128, 32);
                else SPC_LAUNCH_SHORT(128, 64);
#undef SPC_LAUNCH_SHORT
                SPC_LAUNCH_CHECK();
                return SPC_OK;
            }
            if (!(be && atoi(be) == 256) && !keep256) {
                const int64_t nzr = cube->nz;
                const int ts2 = nzr <= 512 ? 64 : (nzr <= 1024 ? 32 : (nzr <= 2048 ? 16 : 8));
                const int kpl2 = 64;
                const bool desc2 = sel_desc_fits(ts2, std::max(cube->plane_stride, arr ? A.mask.plane_stride : 0), 1, 512) && spc_env_on("SPC_SELECT_DESC");
                dim3 grid2((unsigned)(cube->ny * ((cube->nx + ts2 - 1) / ts2)));
#define SPC_LAUNCH_BT(TS_, K_)                                                                                                       \
                do {                                                                                                                \
                    if (desc2) { if (arr) hipLaunchKernelGGL((select_reg_kernel<TS_, K_, true, true, 512>), grid2, dim3(512), 0, st, A);   \
                                 else hipLaunchKernelGGL((select_reg_kernel<TS_, K_, false, true, 512>), grid2, dim3(512), 0, st, A); }    \
                    else { if (arr) hipLaunchKernelGGL((select_reg_kernel<TS_, K_, true, false, 512>), grid2, dim3(512), 0, st, A);        \
                           else hipLaunchKernelGGL((select_reg_kernel<TS_, K_, false, false, 512>), grid2, dim3(512), 0, st, A); }         \
                } while (0)
                (void)kpl2;
                if (ts2 == 64) SPC_LAUNCH_BT(64, 64);
                else if (ts2 == 32) SPC_LAUNCH_BT(32, 64);
                else if (ts2 == 16) SPC_LAUNCH_BT(16, 64);
                else SPC_LAUNCH_BT(8, 64);
#undef SPC_LAUNCH_BT
                SPC_LAUNCH_CHECK();
                return SPC_OK;
            }
        }
        const int ts = cube->nz <= 512 ? 32 : (cube->nz <= 1024 ? 16 : 8);
        // (descriptor loads: measured no gain for the selection - 3.64 against 3.25 ms with a uint8 mask, 2.61 against 2.65 ms
        //  without, 1024^3 - its time is in the digit passes; they pay for the clip kernel: 10.1 against 10.8 ms, mad_std 28.3
        //  against 31.9 ms.  SPC_SELECT_DESC=1 switches them on here.)
        const bool desc = sel_desc_fits(ts, std::max(cube->plane_stride, arr ? A.mask.plane_stride : 0), 1) && spc_env_set("SPC_SELECT_DESC");
        const int lanes = 256 / ts;
        const int need = (int)((cube->nz + lanes - 1) / lanes);
        const int kpl = need <= 16 ? 16 : (need <= 32 ? 32 : (need <= 64 ? 64 : 128));   // 128: 2049 - 4096 channels, 8 spaxels per block
        const int64_t nblk = cube->ny * ((cube->nx + ts - 1) / ts);
        dim3 grid((unsigned)nblk);
#define SPC_LAUNCH_REG(TS_, K_)                                                                                     \
        do {                                                                                                        \
            if (desc) { if (arr) hipLaunchKernelGGL((select_reg_kernel<TS_, K_, true, true>), grid, dim3(256), 0, st, A);   \
                        else hipLaunchKernelGGL((select_reg_kernel<TS_, K_, false, true>), grid, dim3(256), 0, st, A); }    \
            else { if (arr) hipLaunchKernelGGL((select_reg_kernel<TS_, K_, true, false>), grid, dim3(256), 0, st, A);       \
                   else hipLaunchKernelGGL((select_reg_kernel<TS_, K_, false, false>), grid, dim3(256), 0, st, A); }        \
        } while (0)
#define SPC_LAUNCH_REG_K(TS_)                                                                                       \
        do {                                                                                                        \
            if (kpl == 16) SPC_LAUNCH_REG(TS_, 16); else if (kpl == 32) SPC_LAUNCH_REG(TS_, 32); else SPC_LAUNCH_REG(TS_, 64); \
        } while (0)
        if (kpl == 128) SPC_LAUNCH_REG(8, 128); else if (ts == 32) SPC_LAUNCH_REG_K(32); else if (ts == 16) SPC_LAUNCH_REG_K(16); else SPC_LAUNCH_REG_K(8);
#undef SPC_LAUNCH_REG_K
#undef SPC_LAUNCH_REG
        SPC_LAUNCH_CHECK();
        return SPC_OK;
    }
    const char* env = getenv("SPC_SELECT_RADIX16");
    if (v4 && cube->nz <= 65535 && (env ? atoi(env) != 0 : true)) {
        const int64_t n = cube->ny * (cube->nx / 4);
        dim3 grid((unsigned)((n + 255) / 256));
        if (arr) hipLaunchKernelGGL(select16_axis0_kernel<true>, grid, dim3(256), 0, st, A);
        else hipLaunchKernelGGL(select16_axis0_kernel<false>, grid, dim3(256), 0, st, A);
    } else if (v4) {
        const int64_t n = cube->ny * (cube->nx / 4);
        dim3 grid((unsigned)((n + 255) / 256));
        if (arr) hipLaunchKernelGGL((select_axis0_kernel<4, true>), grid, dim3(256), 0, st, A);
        else hipLaunchKernelGGL((select_axis0_kernel<4, false>), grid, dim3(256), 0, st, A);
    } else {
        const int64_t n = cube->ny * cube->nx;
        dim3 grid((unsigned)((n + 255) / 256));
        if (arr) hipLaunchKernelGGL((select_axis0_kernel<1, true>), grid, dim3(256), 0, st, A);
        else hipLaunchKernelGGL((select_axis0_kernel<1, false>), grid, dim3(256), 0, st, A);
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

extern "C" int spc_percentile_global_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                         double q, int has_center, float center, double* h_out,
                                         void* d_workspace, size_t workspace_bytes) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(h_out != nullptr, "h_out is NULL");
    SPC_REQUIRE(q >= 0.0 && q <= 100.0, "Percentiles must be in the range [0, 100]");
    GSelArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.has_center = has_center; A.center = center;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const bool contig = (A.row_stride == A.nx) && (A.plane_stride == A.ny * A.nx) &&
                        (!arr || (A.mask.row_stride == A.nx && A.mask.plane_stride == A.ny * A.nx));
    if (contig) { A.nrows = 1; A.rowlen = A.nz * A.ny * A.nx; A.row_a = 0; A.row_b = 0; }
    else { A.nrows = A.nz * A.ny; A.rowlen = A.nx; A.row_a = A.plane_stride; A.row_b = A.row_stride; }
    const int64_t per_block = 256 * 4 * 8;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(2048, contig ? (A.rowlen + per_block - 1) / per_block : (A.nrows + 3) / 4));
    SpcWorkspace ws(d_workspace, workspace_bytes);
    SPC_WS_TAKE(d_hist, ws, unsigned long long, 257);  // 256 counters + (as uint32) the "next key" cell
    A.hist = d_hist;
    A.next = reinterpret_cast<uint32_t*>(d_hist + 256);
    unsigned long long h[256];
    hipError_t e = hipSuccess;
    unsigned long long n = 0, below = 0, eq = 0;
    long long klo = 0, khi = 0, k = 0;
    double frac = 0.0;
    for (int pass = 0; pass < 4 && e == hipSuccess; ++pass) {
        A.shift = 24 - 8 * pass;
        e = hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * 256, st);
        if (e != hipSuccess) break;
        if (arr) hipLaunchKernelGGL((gselect_kernel<true, 0>), dim3(nblocks), dim3(256), 0, st, A);
        else hipLaunchKernelGGL((gselect_kernel<false, 0>), dim3(nblocks), dim3(256), 0, st, A);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h, d_hist, sizeof(h), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) break;
        if (pass == 0) {
            for (int d = 0; d < 256; ++d) n += h[d];
            if (n == 0) break;
            const double pos = q / 100.0 * (double)(n - 1);
            klo = (long long)floor(pos);
            khi = std::min<long long>((long long)ceil(pos), (long long)n - 1);
            frac = pos - floor(pos);
            k = klo;
        }
        int d = 0;
        while (d < 255 && (unsigned long long)k >= h[d]) { k -= (long long)h[d]; below += h[d]; ++d; }
        eq = h[d];
        A.prefix |= (uint32_t)d << A.shift;
        A.pmask |= 0xffu << A.shift;
    }
    uint32_t key_lo = A.prefix, key_hi = A.prefix;
    if (e == hipSuccess && n > 0 && (unsigned long long)khi >= below + eq) {     // the upper statistic is the next larger value
        e = hipMemsetAsync(A.next, 0xff, sizeof(uint32_t), st);
        if (e == hipSuccess) {
            if (arr) hipLaunchKernelGGL((gselect_kernel<true, 1>), dim3(nblocks), dim3(256), 0, st, A);
            else hipLaunchKernelGGL((gselect_kernel<false, 1>), dim3(nblocks), dim3(256), 0, st, A);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&key_hi, A.next, sizeof(key_hi), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    SPC_HIP(e);
    if (n == 0) { *h_out = NAN; return SPC_OK; }
    auto unkey = [](uint32_t kk) { uint32_t u = (kk & 0x80000000u) ? (kk & 0x7fffffffu) : ~kk; float f; memcpy(&f, &u, 4); return (double)f; };
    const double a = unkey(key_lo), b = unkey(key_hi);
    *h_out = (q == 50.0 && a == b) ? a : ((frac == 0.5) ? 0.5 * (a + b) : a + (b - a) * frac);
    return SPC_OK;
}

// The same statistic along x (median / percentile / mad_std with axis=2) without a transposed copy of the cube: the
// register-resident kernel on the view whose "spaxels" are the (z, y) pairs and whose rays are the rows - consecutive
// lanes take consecutive samples of a ray (contiguous in memory).  d_center / d_out: (nz, ny).  Rows of more than 4096
// samples: SPC_ERR_UNSUPPORTED (the caller transposes, spc_fill_masked_transpose_f32, and selects along y).
extern "C" int spc_percentile_axis2_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                        double q, const float* d_center, float scale, float* d_out) {
    int rc = spc_check_cube_any_order(cube);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    SPC_REQUIRE(q >= 0.0 && q <= 100.0, "Percentiles must be in the range [0, 100]");
    if (cube->nx > 4096 || cube->nz * ((cube->ny + 7) / 8) >= (1LL << 31)) {
        spc_set_error("rows of %lld samples do not fit the registers of a block (4096 at most)", (long long)cube->nx);
        return SPC_ERR_UNSUPPORTED;
    }
    SelArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    sel_key_range(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, &A.key_min, &A.key_span);
    if (rc) return rc;
    SPC_DEVICE(device);
    // the view: samples along x (step 1), rows = channels (step plane_stride), adjacent spaxels = adjacent rows y
    A.cube = cube->d_data;
    A.nz = cube->nx; A.ny = cube->nz; A.nx = cube->ny;
    A.plane_stride = 1; A.row_stride = cube->plane_stride; A.x_stride = cube->row_stride;
    A.m_x_stride = A.mask.row_stride;
    { const int64_t mrow = A.mask.plane_stride; A.mask.plane_stride = 1; A.mask.row_stride = mrow; }
    A.along_ray = 1;
    A.q = q; A.center = d_center; A.scale = scale; A.out = d_out;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    hipStream_t st = (hipStream_t)stream;
    const int ts = A.nz <= 512 ? 32 : (A.nz <= 1024 ? 16 : 8);
    const bool desc = sel_desc_fits(ts, 1, std::max(A.x_stride, arr ? A.m_x_stride : 0)) && spc_env_set("SPC_SELECT_DESC");
    const int lanes = 256 / ts;
    const int need = (int)((A.nz + lanes - 1) / lanes);
    const int kpl = need <= 16 ? 16 : (need <= 32 ? 32 : (need <= 64 ? 64 : 128));
    dim3 grid((unsigned)(A.ny * ((A.nx + ts - 1) / ts)));
#define SPC_LAUNCH_REG(TS_, K_)                                                                                     \
    do {                                                                                                            \
        if (desc) { if (arr) hipLaunchKernelGGL((select_reg_kernel<TS_, K_, true, true>), grid, dim3(256), 0, st, A);       \
                    else hipLaunchKernelGGL((select_reg_kernel<TS_, K_, false, true>), grid, dim3(256), 0, st, A); }        \
        else { if (arr) hipLaunchKernelGGL((select_reg_kernel<TS_, K_, true, false>), grid, dim3(256), 0, st, A);           \
               else hipLaunchKernelGGL((select_reg_kernel<TS_, K_, false, false>), grid, dim3(256), 0, st, A); }            \
    } while (0)
#define SPC_LAUNCH_REG_K(TS_)                                                                                       \
    do {                                                                                                            \
        if (kpl == 16) SPC_LAUNCH_REG(TS_, 16); else if (kpl == 32) SPC_LAUNCH_REG(TS_, 32); else SPC_LAUNCH_REG(TS_, 64); \
    } while (0)
    if (kpl == 128) SPC_LAUNCH_REG(8, 128); else if (ts == 32) SPC_LAUNCH_REG_K(32); else if (ts == 16) SPC_LAUNCH_REG_K(16); else SPC_LAUNCH_REG_K(8);
#undef SPC_LAUNCH_REG_K
#undef SPC_LAUNCH_REG
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

// astropy.stats.sigma_clip(axis=0, masked=False, cenfunc = median | mean, stdfunc = std | mad_std) with the rays resident in
// registers (sigma_clip_reg_kernel): one read and one write of the cube for all iterations.  d_out: (nz, ny, nx)
// C-contiguous float32, masked and clipped samples NaN.  Rays longer than 4096 channels: SPC_ERR_UNSUPPORTED (the
// caller iterates spc_percentile_axis0_f32 / spc_stats_axis_f32 / spc_clip_outside_f32 instead).
extern "C" int spc_sigma_clip_axis0_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                        double sigma_lower, double sigma_upper, int maxiters, int center_is_mean,
                                        int spread_is_mad, float* d_out, void* d_workspace, size_t workspace_bytes) {
    int rc = spc_check_cube_any_order(cube);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    if (cube->nz > 4096 || cube->ny * ((cube->nx + 7) / 8) >= (1LL << 31)) {
        spc_set_error("rays of %lld channels do not fit the registers of a block (4096 at most)", (long long)cube->nz);
        return SPC_ERR_UNSUPPORTED;
    }
    ClipRegArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    sel_key_range(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, &A.key_min, &A.key_span);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.out = d_out;
    A.lo_s = sigma_lower; A.hi_s = sigma_upper; A.maxiters = maxiters; A.cen_mean = center_is_mean ? 1 : 0;
    A.spread_mad = spread_is_mad ? 1 : 0;
    // (16 neighbouring tiles of a row per XCD: a block's half - or quarter - of a 128-byte line is found in the L2 that fetched it
    //  for its neighbour.  1024^3: the clip kernel's read + write floor 4.21 -> 3.28 ms, the 256-thread selection 2.97 -> 2.65 ms;
    //  groups of 2 / 4 / 8: 3.95 / 3.54 / 3.32 - 3.6 ms; no effect on the 512-thread selection.  SPC_XCD_GROUP=0: block = tile)
    { const char* xg = getenv("SPC_XCD_GROUP"); A.xcd_group = xg ? atoi(xg) : 16; }
    { const char* ce = getenv("SPC_SELECT_COMPACT"); A.compact = ce ? atoi(ce) : 2; }   // 0: off, 1: packed rays through the block's loop, 2: every wave on its own
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    hipStream_t st = (hipStream_t)stream;
    // 256-thread blocks for the median forms: with 512 threads (wider runs per plane, see spc_percentile_axis0_f32) the read +
    // write floor of the kernel drops from 4.2 to 2.3 ms at 1024^3, but an iteration with a descent costs 2.0 instead of 1.4 ms
    // (more barriers across 8 waves, the cached descents resume at the shallowest of 32 rays' levels, the loop ends with the
    // slowest of 32 rays): 10.4 against 8.6 ms for astropy's defaults.  Without a descent (centre = mean, spread = std) or
    // with a single iteration the wider blocks win: 2.3 - 3.4 against 4.2 ms.  SPC_SIGMA_BT=256 / 512 forces either.
    {
        const char* be = getenv("SPC_SIGMA_BT");
        const int force = be ? atoi(be) : 0;
        const bool wide = force == 512 || (force != 256 && (A.cen_mean || A.maxiters == 1));
        const bool wide_ok = cube->nz > 512 && cube->nz <= 1024 && !A.spread_mad &&
                             sel_desc_fits(32, std::max(cube->plane_stride, arr ? A.mask.plane_stride : 0), 1, 512);
        // neither forced nor decided by the form: the mask decides (clip_probe_kernel), given a workspace word to decide in
        if (!wide && wide_ok && force == 0 && A.compact >= 2 && d_workspace != nullptr && workspace_bytes >= sizeof(int) &&
            cube->ny * cube->nx >= 4 * kProbeRays && spc_env_on("SPC_SIGMA_PROBE")) {
            int* d_max = static_cast<int*>(d_workspace);
            SPC_HIP(hipMemsetAsync(d_max, 0, sizeof(int), st));
            if (arr) hipLaunchKernelGGL(clip_probe_kernel<true>, dim3(kProbeRays), dim3(256), 0, st, A, d_max);
            else hipLaunchKernelGGL(clip_probe_kernel<false>, dim3(kProbeRays), dim3(256), 0, st, A, d_max);
            A.probe = d_max;
        }
        if ((wide || A.probe != nullptr) && wide_ok) {
            dim3 grid2((unsigned)(cube->ny * ((cube->nx + 31) / 32)));
            if (arr) hipLaunchKernelGGL((sigma_clip_reg_kernel<32, 64, true, false, true, 512>), grid2, dim3(512), 0, st, A);
            else hipLaunchKernelGGL((sigma_clip_reg_kernel<32, 64, false, false, true, 512>), grid2, dim3(512), 0, st, A);
            SPC_LAUNCH_CHECK();
            if (A.probe == nullptr) return SPC_OK;
        }
    }
    // short rays: few lanes per ray (see spc_percentile_axis0_f32): 2 for 65 .. 128 samples, 4 up to 256
    if (cube->nz > 64 && cube->nz <= 256 && !A.spread_mad && spc_env_on("SPC_SIGMA_SHORT") &&
        sel_desc_fits(cube->nz <= 128 ? 128 : 64, std::max(cube->plane_stride, arr ? A.mask.plane_stride : 0), 1, 256)) {
        const int tss = cube->nz <= 128 ? 128 : 64;
        dim3 grids((unsigned)(cube->ny * ((cube->nx + tss - 1) / tss)));
        if (tss == 128) {
            if (arr) hipLaunchKernelGGL((sigma_clip_reg_kernel<128, 64, true, false, true, 256>), grids, dim3(256), 0, st, A);
            else hipLaunchKernelGGL((sigma_clip_reg_kernel<128, 64, false, false, true, 256>), grids, dim3(256), 0, st, A);
        } else {
            if (arr) hipLaunchKernelGGL((sigma_clip_reg_kernel<64, 64, true, false, true, 256>), grids, dim3(256), 0, st, A);
            else hipLaunchKernelGGL((sigma_clip_reg_kernel<64, 64, false, false, true, 256>), grids, dim3(256), 0, st, A);
        }
        SPC_LAUNCH_CHECK();
        return SPC_OK;
    }
    const int ts = cube->nz <= 512 ? 32 : (cube->nz <= 1024 ? 16 : 8);
    const bool desc = sel_desc_fits(ts, std::max(cube->plane_stride, arr ? A.mask.plane_stride : 0), 1) && spc_env_on("SPC_SELECT_DESC");
    const int lanes = 256 / ts;
    const int need = (int)((cube->nz + lanes - 1) / lanes);
    const int kpl = need <= 16 ? 16 : (need <= 32 ? 32 : (need <= 64 ? 64 : 128));
    dim3 grid((unsigned)(cube->ny * ((cube->nx + ts - 1) / ts)));
#define SPC_LAUNCH_CLIP(TS_, K_)                                                                                    \
    do {                                                                                                            \
        if (A.spread_mad) {                                                                                         \
            if (arr) { if (desc) hipLaunchKernelGGL((sigma_clip_reg_kernel<TS_, K_, true, true, true>), grid, dim3(256), 0, st, A); else hipLaunchKernelGGL((sigma_clip_reg_kernel<TS_, K_, true, true, false>), grid, dim3(256), 0, st, A); }   \
            else { if (desc) hipLaunchKernelGGL((sigma_clip_reg_kernel<TS_, K_, false, true, true>), grid, dim3(256), 0, st, A); else hipLaunchKernelGGL((sigma_clip_reg_kernel<TS_, K_, false, true, false>), grid, dim3(256), 0, st, A); }      \
        } else {                                                                                                    \
            if (arr) { if (desc) hipLaunchKernelGGL((sigma_clip_reg_kernel<TS_, K_, true, false, true>), grid, dim3(256), 0, st, A); else hipLaunchKernelGGL((sigma_clip_reg_kernel<TS_, K_, true, false, false>), grid, dim3(256), 0, st, A); }  \
            else { if (desc) hipLaunchKernelGGL((sigma_clip_reg_kernel<TS_, K_, false, false, true>), grid, dim3(256), 0, st, A); else hipLaunchKernelGGL((sigma_clip_reg_kernel<TS_, K_, false, false, false>), grid, dim3(256), 0, st, A); }     \
        }                                                                                                           \
    } while (0)
#define SPC_LAUNCH_CLIP_K(TS_)                                                                                      \
    do {                                                                                                            \
        if (kpl == 16) SPC_LAUNCH_CLIP(TS_, 16); else if (kpl == 32) SPC_LAUNCH_CLIP(TS_, 32); else SPC_LAUNCH_CLIP(TS_, 64); \
    } while (0)
    if (kpl == 128) SPC_LAUNCH_CLIP(8, 128); else if (ts == 32) SPC_LAUNCH_CLIP_K(32); else if (ts == 16) SPC_LAUNCH_CLIP_K(16); else SPC_LAUNCH_CLIP_K(8);
#undef SPC_LAUNCH_CLIP_K
#undef SPC_LAUNCH_CLIP
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

// One pass of the whole-cube selection as its own entry point, for cubes that are sharded over ranks: the ranks sum
// their 256 counters (or take the minimum of their next keys) before the walk, see distributed.sharded_percentile.
//   h_hist != NULL: h_hist[d] = number of included samples whose key agrees with `prefix` on the bits of `pmask` and
//                   has byte value d at bit `shift` (24, 16, 8, 0);
//   h_next != NULL: *h_next = the smallest key above `prefix` (0xffffffff when there is none).
// Keys are the order-preserving integers of the samples (of |x - center| with has_center); spc_key_to_f32 maps back.
extern "C" int spc_key_histogram_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                     uint32_t prefix, uint32_t pmask, int shift, int has_center, float center,
                                     uint64_t* h_hist, uint32_t* h_next, void* d_workspace, size_t workspace_bytes) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE((h_hist != nullptr) != (h_next != nullptr), "exactly one of h_hist / h_next must be given");
    SPC_REQUIRE(h_next || shift == 24 || shift == 16 || shift == 8 || shift == 0, "shift must be 24, 16, 8 or 0");
    GSelArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.has_center = has_center; A.center = center;
    A.prefix = prefix; A.pmask = pmask; A.shift = shift;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const bool contig = (A.row_stride == A.nx) && (A.plane_stride == A.ny * A.nx) &&
                        (!arr || (A.mask.row_stride == A.nx && A.mask.plane_stride == A.ny * A.nx));
    if (contig) { A.nrows = 1; A.rowlen = A.nz * A.ny * A.nx; A.row_a = 0; A.row_b = 0; }
    else { A.nrows = A.nz * A.ny; A.rowlen = A.nx; A.row_a = A.plane_stride; A.row_b = A.row_stride; }
    const int64_t per_block = 256 * 4 * 8;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(2048, contig ? (A.rowlen + per_block - 1) / per_block : (A.nrows + 3) / 4));
    SpcWorkspace ws(d_workspace, workspace_bytes);
    SPC_WS_TAKE(d_hist, ws, unsigned long long, 257);
    A.hist = d_hist;
    A.next = reinterpret_cast<uint32_t*>(d_hist + 256);
    hipError_t e;
    if (h_hist) {
        e = hipMemsetAsync(d_hist, 0, sizeof(unsigned long long) * 256, st);
        if (e == hipSuccess) {
            if (arr) hipLaunchKernelGGL((gselect_kernel<true, 0>), dim3(nblocks), dim3(256), 0, st, A);
            else hipLaunchKernelGGL((gselect_kernel<false, 0>), dim3(nblocks), dim3(256), 0, st, A);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h_hist, d_hist, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost, st);
    } else {
        e = hipMemsetAsync(A.next, 0xff, sizeof(uint32_t), st);
        if (e == hipSuccess) {
            if (arr) hipLaunchKernelGGL((gselect_kernel<true, 1>), dim3(nblocks), dim3(256), 0, st, A);
            else hipLaunchKernelGGL((gselect_kernel<false, 1>), dim3(nblocks), dim3(256), 0, st, A);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h_next, A.next, sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    SPC_HIP(e);
    return SPC_OK;
}

size_t spc_ws_sigma_clip(void) { return 256; }

extern "C" float spc_key_to_f32(uint32_t key) {
    const uint32_t u = (key & 0x80000000u) ? (key & 0x7fffffffu) : ~key;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

size_t spc_ws_percentile_global(void) { return spc_ws_round(sizeof(unsigned long long) * 257) + 256; }
