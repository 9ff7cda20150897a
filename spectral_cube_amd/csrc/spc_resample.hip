// Spectral linear interpolation and spatial bilinear resampling.
//
// spc_spectral_lerp_f32 replaces interp_wrapper / scipy.interpolate.interp1d
// (kind='linear', bounds_error=False) inside DaskSpectralCubeMixin.
// spectral_interpolate (spectral_cube/dask_spectral_cube.py:1342-1353).
// spc_resample_bilinear_f32 replaces the resampler of reproject.
// reproject_interp(order='bilinear') called by BaseSpectralCube.reproject
// (spectral_cube/spectral_cube.py:2726-2732).
//
// Both are pure HBM streams: lanes run along x (coalesced), each lane keeps
// its bracketing input samples in registers while the (wave-uniform) output
// channel index advances, so every input plane is read once.
#include "spc_common.h"
#include <algorithm>

namespace {

struct LerpArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    int64_t nz_out;
    const int32_t* lo;
    const double* t;
    const double* inv_dx;
    float fill;
    float* out;
    int64_t out_row_stride, out_plane_stride;
    int64_t jchunk;
    int nt_store;
};

__device__ __forceinline__ float load_filled(const float* p, const uint8_t* pm, const MaskDev& m, int64_t off, int64_t moff) {
    const float v = p[off];
    bool inc = spc_pred_valid(m, v);                         // (an excluded sample becomes NaN: rejecting NaN itself changes nothing)
    if (pm) inc = inc && pm[moff] != 0;
    return inc ? v : NAN;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));
template <int VEC> struct LV { using F = float; };
template <> struct LV<4> { using F = f32x4; };
__device__ __forceinline__ float lget(const float& v, int) { return v; }
__device__ __forceinline__ float lget(const f32x4& v, int i) { return v[i]; }
__device__ __forceinline__ void lset(float& v, int, float x) { v = x; }
__device__ __forceinline__ void lset(f32x4& v, int i, float x) { v[i] = x; }

// VEC consecutive x per lane (16-byte loads/stores when the rows allow it)
template <int VEC, bool NT = true>
__device__ __forceinline__ typename LV<VEC>::F load_filled_v(const float* p, const uint8_t* pm, const MaskDev& m,
                                                             int64_t off, int64_t moff) {
    using F = typename LV<VEC>::F;
    F v = NT ? __builtin_nontemporal_load(reinterpret_cast<const F*>(p + off)) : *reinterpret_cast<const F*>(p + off);
    if (m.flags) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const float x = lget(v, i);
            bool inc = spc_pred_valid(m, x);
            if (pm) inc = inc && pm[moff + i] != 0;
            lset(v, i, inc ? x : NAN);
        }
    }
    return v;
}

// Each lane owns VEC adjacent spaxels and marches along the output channels in groups of U: the
// bracketing input planes of a whole group are requested first (lo[] is wave-uniform, so which planes
// those are is scalar control flow), then the group is interpolated and stored.  U = 1 is what is
// launched: at C5 (2048 -> 4096 channels, two thirds of the traffic are stores) it runs at 5.07 TB/s,
// U = 4 at 4.84 and U = 8 at 3.80 - eight waves per SIMD already cover the load latency, and deeper
// groups only make the store stream burstier and cost occupancy.
template <int VEC, int U>
__global__ __launch_bounds__(256) void spectral_lerp_kernel(const LerpArgs A) {
    using F = typename LV<VEC>::F;
    const int64_t gpr = A.nx / VEC;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= A.ny * gpr) return;
    const int64_t y = g / gpr, x = (g - y * gpr) * VEC;
    const float* p = A.cube + y * A.row_stride + x;
    const uint8_t* pm = (A.mask.flags & SPC_MASK_ARRAY) ? A.mask.arr + y * A.mask.row_stride + x : nullptr;
    float* po = A.out + y * A.out_row_stride + x;
    const int64_t jb = (int64_t)blockIdx.y * A.jchunk;
    const int64_t je = min(A.nz_out, jb + A.jchunk);
    int cur = -2;
    F ylo{}, yhi{};
    for (int64_t j0 = jb; j0 < je; j0 += U) {
        int los[U];
        F a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int lo = A.lo[min(j0 + u, je - 1)];          // wave-uniform
            los[u] = lo;
            if (lo >= 0 && lo != cur) {
                if (lo == cur + 1) ylo = yhi;
                else ylo = load_filled_v<VEC>(p, pm, A.mask, (int64_t)lo * A.plane_stride, (int64_t)lo * A.mask.plane_stride);
                yhi = load_filled_v<VEC>(p, pm, A.mask, (int64_t)(lo + 1) * A.plane_stride, (int64_t)(lo + 1) * A.mask.plane_stride);
                cur = lo;
            }
            a[u] = ylo; b[u] = yhi;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = j0 + u;
            if (j >= je) break;
            F res;
            if (los[u] < 0) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) lset(res, i, A.fill);
            } else {
                // scipy: slope = (y_hi - y_lo) / (x_hi - x_lo); y = slope * (x_new - x_lo) + y_lo
                const double wj = A.inv_dx[j] * A.t[j];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float diff = lget(b[u], i) - lget(a[u], i);       // float32 like numpy's f32 - f32
                    lset(res, i, (float)((double)diff * wj + (double)lget(a[u], i)));
                }
            }
            if (A.nt_store) __builtin_nontemporal_store(res, reinterpret_cast<F*>(po + j * A.out_plane_stride));
            else *reinterpret_cast<F*>(po + j * A.out_plane_stride) = res;
        }
    }
}

// Round 5: the same operator as SHORT-LIVED blocks.  tools/micro/copy_ceiling.hip: a textbook copy - one 16-byte load and store per
// lane, a block per 4 KiB - reaches 6.3 - 6.6 TB/s where every persistent form (this march included: 4.7 - 5.1 TB/s) stays
// below 5.5; tools/micro/lerp_patterns.hip: blocks that own one 4-KiB row segment of JC = 4 consecutive output channels - the
// three input planes they need requested together, then four stores, then the block ends - interpolate 2048 -> 4096 channels
// in 4.46 ms against the march's 5.1 (the input planes are fetched x1.5, from the L2 / Infinity Cache the second time).
// A block = 256 lanes x 16 bytes of ONE row x JC output channels (grid: x = row segments, y = channel groups, x fastest: the
// blocks in flight cover whole planes of one group).  lo[] is monotone: the planes base, base + 1, base + 2 serve every output
// of the group whose lo is base or base + 1; an output beyond that (down-sampling grids) starts the next batch of three.
template <int JC>
__global__ __launch_bounds__(256) void spectral_lerp_tiles_kernel(const LerpArgs A, int blocks_per_row) {
    const int y = (int)(blockIdx.x / (unsigned)blocks_per_row);                    // (scalar arithmetic)
    const int bx = (int)blockIdx.x - y * blocks_per_row;
    const int x = (bx * 256 + (int)threadIdx.x) * 4;
    if (x >= A.nx) return;
    const int64_t j0 = (int64_t)blockIdx.y * JC;
    const float* p = A.cube + (int64_t)y * A.row_stride + x;
    const uint8_t* pm = (A.mask.flags & SPC_MASK_ARRAY) ? A.mask.arr + (int64_t)y * A.mask.row_stride + x : nullptr;
    float* po = A.out + (int64_t)y * A.out_row_stride + x;
    int los[JC];
    double wj[JC];
#pragma unroll
    for (int u = 0; u < JC; ++u) {
        const int64_t j = min(j0 + u, A.nz_out - 1);
        los[u] = A.lo[j];
        wj[u] = A.inv_dx[j] * A.t[j];
    }
    int base = -2;
    f32x4 p0{}, p1{}, p2{};
    auto emit = [&](const f32x4& a, const f32x4& b, int u) {
        // scipy: slope = (y_hi - y_lo) / (x_hi - x_lo); y = slope * (x_new - x_lo) + y_lo
        f32x4 res;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float diff = b[i] - a[i];                               // float32 like numpy's f32 - f32
            res[i] = (float)((double)diff * wj[u] + (double)a[i]);
        }
        *reinterpret_cast<f32x4*>(po + (j0 + u) * A.out_plane_stride) = res;
    };
#pragma unroll
    for (int u = 0; u < JC; ++u) {
        if (j0 + u >= A.nz_out) break;
        const int lo = los[u];                                            // wave-uniform
        if (lo < 0) {
            *reinterpret_cast<f32x4*>(po + (j0 + u) * A.out_plane_stride) = f32x4{A.fill, A.fill, A.fill, A.fill};
            continue;
        }
        if (base < 0 || lo > base + 1 || lo < base) {
            base = lo;
            const int64_t l2 = min((int64_t)lo + 2, A.nz - 1);
            p0 = load_filled_v<4, false>(p, pm, A.mask, (int64_t)lo * A.plane_stride, (int64_t)lo * A.mask.plane_stride);
            p1 = load_filled_v<4, false>(p, pm, A.mask, (int64_t)(lo + 1) * A.plane_stride, (int64_t)(lo + 1) * A.mask.plane_stride);
            p2 = load_filled_v<4, false>(p, pm, A.mask, l2 * A.plane_stride, l2 * A.mask.plane_stride);
        }
        if (lo == base) emit(p0, p1, u);
        else emit(p1, p2, u);
    }
}

struct BilArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    float fill;
    int64_t ny_out, nx_out;
    const double* xs;
    const double* ys;
    float* out;
    int64_t out_row_stride, out_plane_stride;
    uint8_t* footprint;
    int64_t zchunk;
    // LDS-staged pass: one byte per 32 x 32 output tile, 0 = done by bilinear_lds_kernel
    unsigned char* status;
    int64_t tiles32_x, ntiles32;    // tile grid of the LDS pass (tiles of 1 << tile_shift pixels)
    int tile_shift;
    int64_t zchunk_lds;
    int nearest;               // order 0: the nearest sample alone (scipy map_coordinates order=0)
    unsigned int* any_valid;   // optional device word: set to 1 by a block that wrote a non-NaN value
    // LERP (spc_resample_bilinear_lerp_f32): the spectral interpolation folded in.  Output channel j is the linear blend of the
    // RESAMPLED input planes lo[j] and lo[j] + 1 (spectral_lerp_kernel's arithmetic); jfirst[z] = the first output channel
    // whose left bracket is >= z (jfirst[0] = j_begin, jfirst[nz - 1] = j_end: lerp_index_kernel), channels outside
    // [j_begin, j_end) have lo < 0 and are NaN planes.
    int64_t nz_out;
    const int32_t* lo;
    const double* t;
    const double* inv_dx;
    const int32_t* jfirst;     // nz entries
};

// jfirst[] of a lerp plan whose non-negative lo[] ascend (one block; nz and nz_out are a few thousand)
__global__ __launch_bounds__(256) void lerp_index_kernel(const int32_t* lo, int nz_out, int nz, int32_t* jfirst) {
    __shared__ int s_ja, s_jb;
    if (threadIdx.x == 0) { s_ja = nz_out; s_jb = 0; }
    __syncthreads();
    int ja = nz_out, jb = 0;
    for (int j = threadIdx.x; j < nz_out; j += blockDim.x)
        if (lo[j] >= 0) { ja = min(ja, j); jb = max(jb, j + 1); }
    atomicMin(&s_ja, ja); atomicMax(&s_jb, jb);
    __syncthreads();
    ja = s_ja; jb = max(s_jb, s_ja);
    for (int z = threadIdx.x; z < nz; z += blockDim.x) {
        int a = ja, b = jb;                    // first j in [ja, jb) with lo[j] >= z
        while (a < b) { const int mid = (a + b) >> 1; if (lo[mid] >= z) b = mid; else a = mid + 1; }
        jfirst[z] = (z == nz - 1) ? jb : a;
    }
}

// one output value of the folded-in spectral interpolation: scipy's slope form, as spectral_lerp_kernel
__device__ __forceinline__ float bil_lerp(float a, float b, double wj) {
    const float diff = b - a;
    return (float)((double)diff * wj + (double)a);
}

// Lane <-> output pixel of a 16 x 4 tile per wavefront (a 64 x 16 tile per block): the
// source footprint of a compact tile touches ~3x fewer cache lines per gather than a
// 64-pixel output row does for a rotated grid, while stores stay 64 B contiguous per row.
template <bool LERP>
__global__ __launch_bounds__(256) void bilinear_kernel(const BilArgs A) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tiles_x = (A.nx_out + 63) / 64;
    const int64_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int64_t xo = bx * 64 + (wave * 16) + (lane & 15);
    const int64_t yo = by * 4 + (lane >> 4);
    if (xo >= A.nx_out || yo >= A.ny_out) return;
    if (A.status && spc_flag_get(A.status + (yo >> A.tile_shift) * A.tiles32_x + (xo >> A.tile_shift)) == 0) return;   // wave-uniform: tile done via LDS
    const int64_t pix = yo * A.nx_out + xo;
    const double xs = A.xs[pix], ys = A.ys[pix];
    const bool inside = (xs >= -0.5) && (xs <= (double)A.nx - 0.5) && (ys >= -0.5) && (ys <= (double)A.ny - 0.5);
    if (blockIdx.y == 0 && A.footprint) A.footprint[pix] = inside ? 1 : 0;
    const int64_t zb = (int64_t)blockIdx.y * A.zchunk;
    const int64_t ze = min(LERP ? A.nz - 1 : A.nz, zb + A.zchunk);       // LERP: the left brackets [zb, ze) this block serves
    float* po = A.out + yo * A.out_row_stride + xo;
    if (LERP && blockIdx.y == 0) {                                        // channels outside the input range
        const int64_t ja = A.jfirst[0], jb = A.jfirst[A.nz - 1];
        for (int64_t j = 0; j < ja; ++j) po[j * A.out_plane_stride] = NAN;
        for (int64_t j = jb; j < A.nz_out; ++j) po[j * A.out_plane_stride] = NAN;
    }
    if (!inside) {
        if (LERP) { for (int64_t j = A.jfirst[zb]; j < A.jfirst[ze]; ++j) po[j * A.out_plane_stride] = NAN; }
        else for (int64_t z = zb; z < ze; ++z) po[z * A.out_plane_stride] = NAN;
        return;
    }
    // reproject's resampler: scipy map_coordinates(order=1) on the image padded by one
    // edge-replicated pixel.  floor(xs) in [-1, nx-1]; neighbours -1 / nx are the border pixel
    // itself, so within half a pixel of the border both neighbours coincide.
    // order 0: the sample at floor(x + 0.5) (border zone: the border pixel), both "neighbours" coincide
    const double xf = A.nearest ? floor(xs + 0.5) : floor(xs), yf = A.nearest ? floor(ys + 0.5) : floor(ys);
    const int64_t x0 = min(max((int64_t)xf, (int64_t)0), A.nx - 1), y0 = min(max((int64_t)yf, (int64_t)0), A.ny - 1);
    const int64_t x1 = A.nearest ? x0 : min((int64_t)xf + 1, A.nx - 1), y1 = A.nearest ? y0 : min((int64_t)yf + 1, A.ny - 1);
    const float fx = (float)(xs - xf), fy = (float)(ys - yf);
    const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
    const int64_t o00 = y0 * A.row_stride + x0, o01 = y0 * A.row_stride + x1;
    const int64_t o10 = y1 * A.row_stride + x0, o11 = y1 * A.row_stride + x1;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const int64_t m00 = y0 * A.mask.row_stride + x0, m01 = y0 * A.mask.row_stride + x1;
    const int64_t m10 = y1 * A.mask.row_stride + x0, m11 = y1 * A.mask.row_stride + x1;
    const bool anymask = A.mask.flags != 0;
    constexpr int U = 4;                      // channels in flight per lane (8 measured slower)
    bool anyv = false;
    if (LERP) {
        // (the path of the tiles the LDS pass could not take: one input plane at a time)
        const bool anymask_l = anymask;
        auto plane = [&](int64_t z) {
            const float* p = A.cube + z * A.plane_stride;
            float aa = p[o00], bb = p[o01], cc = p[o10], dd = p[o11];
            if (anymask_l) {
                const uint8_t* pm = arr ? A.mask.arr + z * A.mask.plane_stride : nullptr;
                bool i0 = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, aa);
                bool i1 = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, bb);
                bool i2 = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, cc);
                bool i3 = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, dd);
                if (arr) { i0 = i0 && pm[m00]; i1 = i1 && pm[m01]; i2 = i2 && pm[m10]; i3 = i3 && pm[m11]; }
                aa = i0 ? aa : A.fill; bb = i1 ? bb : A.fill; cc = i2 ? cc : A.fill; dd = i3 ? dd : A.fill;
            }
            return A.nearest ? aa : fmaf(w11, dd, fmaf(w10, cc, fmaf(w01, bb, w00 * aa)));
        };
        float prev = plane(zb);
        for (int64_t z = zb + 1; z <= ze; ++z) {
            const float cur = plane(z);
            for (int64_t j = A.jfirst[z - 1]; j < A.jfirst[z]; ++j) {
                const float r = bil_lerp(prev, cur, A.inv_dx[j] * A.t[j]);
                anyv = anyv || (r == r);
                __builtin_nontemporal_store(r, po + j * A.out_plane_stride);
            }
            prev = cur;
        }
        if (A.any_valid && __any(anyv) && lane == 0) atomicOr(A.any_valid, 1u);
        return;
    }
    for (int64_t zq = zb; zq < ze; zq += U) {
        float a[U], b[U], c[U], d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t z = min(zq + u, ze - 1);
            const float* p = A.cube + z * A.plane_stride;
            a[u] = p[o00]; b[u] = p[o01]; c[u] = p[o10]; d[u] = p[o11];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t z = zq + u;
            if (z >= ze) break;
            float aa = a[u], bb = b[u], cc = c[u], dd = d[u];
            if (anymask) {
                const uint8_t* pm = arr ? A.mask.arr + z * A.mask.plane_stride : nullptr;
                // excluded voxels are replaced by the cube's fill value (spectral_cube.py:2709-2712)
                bool i0 = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, aa);
                bool i1 = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, bb);
                bool i2 = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, cc);
                bool i3 = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, dd);
                if (arr) { i0 = i0 && pm[m00]; i1 = i1 && pm[m01]; i2 = i2 && pm[m10]; i3 = i3 && pm[m11]; }
                aa = i0 ? aa : A.fill; bb = i1 ? bb : A.fill; cc = i2 ? cc : A.fill; dd = i3 ? dd : A.fill;
            }
            // plain weighted sum like scipy: a NaN neighbour propagates even with weight 0
            const float r = A.nearest ? aa : fmaf(w11, dd, fmaf(w10, cc, fmaf(w01, bb, w00 * aa)));
            anyv = anyv || (r == r);
            __builtin_nontemporal_store(r, po + z * A.out_plane_stride);
        }
    }
    if (A.any_valid && __any(anyv) && lane == 0) atomicOr(A.any_valid, 1u);
}

// ---- LDS-staged bilinear resampling ------------------------------------------------------
// The gather kernel above is bound by the texture-address path: a rotated 16 x 4 tile makes
// every one of its 4 loads per channel touch ~14 cache lines.  Here a block owns a 32 x 32
// output tile for a chunk of channels.  Once per block the pixel map of the tile is turned
// into (a) the exact source footprint, row by row: [xmin(y), xmax(y)] spans found with LDS
// atomics, packed by a prefix sum, and (b) per-thread lists of source offsets.  Per channel
// the footprint (~1.2x the tile area for a rotation) is then read with coalesced row-span
// loads into LDS, the 4 neighbours of every output pixel come from LDS, and each lane writes
// 4 adjacent pixels with one 16-byte store.  The per-pixel arithmetic is the gather
// kernel's, so both give identical bits; a tile whose footprint does not fit (strong
// down-sampling) is flagged and left to the gather kernel.
// Tile geometry.  Row spans of a 32 x 32 tile average 23 floats for a 30 degree rotation, so the 64-byte
// sector granularity alone amplifies the reads x1.7 (PMC, round 1); a 64 x 64 tile has spans of ~55 floats
// (x1.3) at the price of one 1024-thread block per CU and 4 instead of 8 staged channels.
template <int TILE> struct BilTile;
template <> struct BilTile<32> {
    static constexpr int kThreads = 256, kRowsMax = 64, kElemsMax = 2048, kStageU = 8;
};
template <> struct BilTile<64> {
    static constexpr int kThreads = 1024, kRowsMax = 128, kElemsMax = 6144, kStageU = 4;
};

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int TILE, bool ARR, bool ANYMASK, bool LERP = false>
__global__ __launch_bounds__(BilTile<TILE>::kThreads) void bilinear_lds_kernel(const BilArgs A) {
    using G = BilTile<TILE>;
    constexpr int kTile = TILE, kRowsMax = G::kRowsMax, kElemsMax = G::kElemsMax, kStageU = G::kStageU;
    constexpr int kThreads = G::kThreads, kFill = kElemsMax / kThreads;       // samples per thread and channel
    constexpr int kLanesX = TILE / 4;                                        // threads along x (4 adjacent pixels each)
    __shared__ float stage[kStageU * kElemsMax];
    __shared__ int s_xmin[kRowsMax], s_xmax[kRowsMax], s_off[kRowsMax + 1], s_tot[kRowsMax / 64];
    __shared__ int s_ymin, s_ymax;
    const int t = threadIdx.x;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
    // consecutive block ids would put neighbouring tiles - whose source footprints share cache lines -
    // on different L2s.  Block b works on tile (b % 8) * ceil(ntiles / 8) + b / 8: every XCD walks a
    // contiguous band of the tile grid.
    const int64_t per_xcd = (A.ntiles32 + 7) / 8;
    const int64_t tile_id = (int64_t)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (tile_id >= A.ntiles32) return;
    const int64_t bx = tile_id % A.tiles32_x, by = tile_id / A.tiles32_x;
    const int64_t xo = bx * kTile + (t % kLanesX) * 4;     // first of this lane's 4 adjacent pixels
    const int64_t yo = by * kTile + (t / kLanesX);

    bool inside[4];
    int x0[4], y0[4], dx[4], dy[4];
    float w00[4], w01[4], w10[4], w11[4];
    if (t == 0) { s_ymin = 0x7fffffff; s_ymax = -1; }
    if (t < kRowsMax) { s_xmin[t] = 0x7fffffff; s_xmax[t] = -1; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        inside[q] = false; x0[q] = 0; y0[q] = 0; dx[q] = 0; dy[q] = 0;
        w00[q] = w01[q] = w10[q] = w11[q] = 0.f;
        if (xo + q < A.nx_out && yo < A.ny_out) {
            const int64_t pix = yo * A.nx_out + xo + q;
            const double xs = A.xs[pix], ys = A.ys[pix];
            inside[q] = (xs >= -0.5) && (xs <= (double)A.nx - 0.5) && (ys >= -0.5) && (ys <= (double)A.ny - 0.5);
            if (blockIdx.y == 0 && A.footprint) A.footprint[pix] = inside[q] ? 1 : 0;
            if (inside[q]) {
                const double xf = A.nearest ? floor(xs + 0.5) : floor(xs), yf = A.nearest ? floor(ys + 0.5) : floor(ys);      // see bilinear_kernel
                x0[q] = min(max((int)xf, 0), (int)A.nx - 1); y0[q] = min(max((int)yf, 0), (int)A.ny - 1);
                dx[q] = A.nearest ? 0 : min((int)xf + 1, (int)A.nx - 1) - x0[q];   // 0 inside the replicated border, else 1
                dy[q] = A.nearest ? 0 : min((int)yf + 1, (int)A.ny - 1) - y0[q];
                const float fx = (float)(xs - xf), fy = (float)(ys - yf);
                w00[q] = (1.f - fy) * (1.f - fx); w01[q] = (1.f - fy) * fx; w10[q] = fy * (1.f - fx); w11[q] = fy * fx;
                atomicMin(&s_ymin, y0[q]);
                atomicMax(&s_ymax, y0[q] + dy[q]);
            }
        }
    }
    __syncthreads();
    const int ymin = s_ymin;
    const int nrows = (s_ymax < 0) ? 0 : s_ymax - ymin + 1;
    const int64_t tile = by * A.tiles32_x + bx;
    if (nrows > kRowsMax) { if (t == 0) spc_flag_set(A.status + tile); return; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (inside[q]) {
            const int r = y0[q] - ymin;
            atomicMin(&s_xmin[r], x0[q]);         atomicMax(&s_xmax[r], x0[q] + dx[q]);
            atomicMin(&s_xmin[r + dy[q]], x0[q]); atomicMax(&s_xmax[r + dy[q]], x0[q] + dx[q]);
        }
    }
    __syncthreads();
    if (t < kRowsMax) {                        // kRowsMax / 64 waves pack the row spans (inclusive scan per wave ...)
        int len = (t < nrows && s_xmax[t] >= 0) ? s_xmax[t] - s_xmin[t] + 1 : 0;
        int acc = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(acc, d, 64);
            if ((t & 63) >= d) acc += up;
        }
        s_off[t + 1] = acc;
        if ((t & 63) == 63) s_tot[t >> 6] = acc;
        if (t == 0) s_off[0] = 0;
    }
    __syncthreads();
    if (kRowsMax > 64) {                       // ... then the totals of the waves before are added)
        if (t >= 64 && t < kRowsMax) {
            int base = 0;
            for (int w = 0; w < (t >> 6); ++w) base += s_tot[w];
            s_off[t + 1] += base;
        }
        __syncthreads();
    }
    const int E = s_off[nrows];
    if (E > kElemsMax) { if (t == 0) spc_flag_set(A.status + tile); return; }

    // per-thread fill list: staged element e = t + 256 k lives in source row ymin + r
    int foff[kFill], moff[kFill];
#pragma unroll
    for (int k = 0; k < kFill; ++k) {
        const int e = t + kThreads * k;
        foff[k] = -1; moff[k] = 0;
        if (e < E) {
            int lo = 0, hi = nrows;            // largest r with s_off[r] <= e
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= e) lo = mid; else hi = mid; }
            const int xcol = s_xmin[lo] + (e - s_off[lo]);
            foff[k] = (ymin + lo) * (int)A.row_stride + xcol;
            if (ARR) moff[k] = (ymin + lo) * (int)A.mask.row_stride + xcol;
        }
    }
    int l0[4], l1[4];                          // LDS index of the (y0, x0) and (y0 + dy, x0) neighbours
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        l0[q] = 0; l1[q] = 0;
        if (inside[q]) {
            const int r = y0[q] - ymin;
            l0[q] = s_off[r] + x0[q] - s_xmin[r];
            l1[q] = s_off[r + dy[q]] + x0[q] - s_xmin[r + dy[q]];
        }
    }
    // LERP: this block serves the output channels whose left bracket lies in [zb, zb + zchunk_lds) and stages one plane more
    const int64_t zb = (int64_t)blockIdx.y * A.zchunk_lds;
    const int64_t ze = LERP ? min(A.nz, zb + A.zchunk_lds + 1) : min(A.nz, zb + A.zchunk_lds);
    if (zb >= (LERP ? A.nz - 1 : ze)) return;
    const bool vec_ok = (xo + 4 <= A.nx_out) && (A.out_row_stride % 4 == 0) && (A.out_plane_stride % 4 == 0) &&
                        ((((uintptr_t)A.out) & 15) == 0);
    typedef float f32x4 __attribute__((ext_vector_type(4)));

    float pre[kStageU][kFill];
    auto fetch = [&](int64_t zq) {
#pragma unroll
        for (int u = 0; u < kStageU; ++u) {
            const int64_t z = min(zq + u, ze - 1);
            const float* p = A.cube + z * A.plane_stride;
            const uint8_t* pm = ARR ? A.mask.arr + z * A.mask.plane_stride : nullptr;
#pragma unroll
            for (int k = 0; k < kFill; ++k) {
                if (kThreads * k < E) {        // block-uniform
                    float v = 0.f;
                    if (foff[k] >= 0) {
                        v = p[foff[k]];
                        if (ANYMASK) {
                            // excluded voxels are replaced by the cube's fill value (spectral_cube.py:2709-2712)
                            bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, v);
                            if (ARR) inc = inc && pm[moff[k]];
                            v = inc ? v : A.fill;
                        }
                    }
                    pre[u][k] = v;
                }
            }
        }
    };
    fetch(zb);
    bool anyv = false;
    float prev[4] = {NAN, NAN, NAN, NAN};
    auto put = [&](int64_t plane, const float (&r4)[4]) {
        float* po = A.out + plane * A.out_plane_stride + yo * A.out_row_stride + xo;
        if (yo < A.ny_out) {
            if (vec_ok) __builtin_nontemporal_store(f32x4{r4[0], r4[1], r4[2], r4[3]}, reinterpret_cast<f32x4*>(po));
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (xo + q < A.nx_out) po[q] = r4[q];
            }
        }
    };
    if (LERP && blockIdx.y == 0) {                                        // channels outside the input range
        const float nan4[4] = {NAN, NAN, NAN, NAN};
        const int64_t ja = A.jfirst[0], jb = A.jfirst[A.nz - 1];
        for (int64_t j = 0; j < ja; ++j) put(j, nan4);
        for (int64_t j = jb; j < A.nz_out; ++j) put(j, nan4);
    }
    // LERP: the output channels of a group's planes and their blend weights go through LDS with the group's samples - read as
    // scalar loads inside the loop below they sit behind the stores (the compiler must assume the stores alias them: every
    // output waited out a scalar round trip)
    constexpr int kWj = 256;
    __shared__ double s_wj[LERP ? kWj : 1];
    __shared__ int s_jf[LERP ? kStageU + 1 : 1];
    for (int64_t zq = zb; zq < ze; zq += kStageU) {
#pragma unroll
        for (int u = 0; u < kStageU; ++u)
#pragma unroll
            for (int k = 0; k < kFill; ++k)
                if (kThreads * k < E && foff[k] >= 0) stage[u * kElemsMax + t + kThreads * k] = pre[u][k];
        if (LERP) {
            // s_jf[u] = first output channel whose left bracket is plane zq + u - 1 (u = 0 .. kStageU); weights of the group's outputs
            if (t <= kStageU) s_jf[t] = A.jfirst[min(max(zq + t - 1, (int64_t)0), A.nz - 1)];
            const int jlo = A.jfirst[min(max(zq - 1, (int64_t)0), A.nz - 1)];
            const int jhi = A.jfirst[min(zq + kStageU - 1, A.nz - 1)];
            for (int i = t; i < min(jhi - jlo, kWj); i += kThreads) s_wj[i] = A.inv_dx[jlo + i] * A.t[jlo + i];
        }
        lds_only_barrier();
        if (zq + kStageU < ze) fetch(zq + kStageU);        // in flight while this group is resampled
#pragma unroll
        for (int u = 0; u < kStageU; ++u) {
            const int64_t z = zq + u;
            if (z >= ze) break;
            const float* sp = stage + u * kElemsMax;
            float r4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float aa = sp[l0[q]], bb = sp[l0[q] + dx[q]], cc = sp[l1[q]], dd = sp[l1[q] + dx[q]];
                const float r = A.nearest ? aa : fmaf(w11[q], dd, fmaf(w10[q], cc, fmaf(w01[q], bb, w00[q] * aa)));
                r4[q] = inside[q] ? r : NAN;
                if (!LERP) anyv = anyv || (r4[q] == r4[q]);
            }
            if (!LERP) put(z, r4);
            else {
                if (z > zb) {
                    const int jbase = s_jf[0], j1 = s_jf[u + 1];
                    for (int j = s_jf[u]; j < j1; ++j) {
                        const double wj = (j - jbase < kWj) ? s_wj[j - jbase] : A.inv_dx[j] * A.t[j];
                        float o4[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { o4[q] = bil_lerp(prev[q], r4[q], wj); anyv = anyv || (o4[q] == o4[q]); }
                        put(j, o4);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) prev[q] = r4[q];
            }
        }
        lds_only_barrier();
    }
    if (A.any_valid && __any(anyv) && (t & 63) == 0) atomicOr(A.any_valid, 1u);
}


// ---- reprojection pixel map (FITS paper II, float64) ----------------------------------------------
// Same sequence of operations as spectral_cube_amd/wcs.py (celestial_pix2world of the target followed
// by celestial_world2pix of the source), which is validated against astropy.wcs; on the host this map
// costs 0.8 s for 1024^2 pixels - a hundred times the resampling kernel it feeds.
// has_rot: bit 0 = rotate, bit 1 = remove the E-terms vector et[0..2] before (FK4 target), bit 2 = add et[3..5] after (FK4 source)
struct WcsPair { spc_celestial_wcs o, i; int64_t ny, nx; double* xs; double* ys; int has_rot; double rot[9]; double et[6]; };

// SIP polynomial sum_{p + q <= n} c[p][q] u^p v^q and its two partial derivatives (Horner in v inside Horner in u;
// row p of the triangular table starts at p * 10 - p * (p - 1) / 2)
__device__ __forceinline__ void sip_eval(const double* c, int n, double u, double v, double& f, double& fu, double& fv) {
    f = 0.0; fu = 0.0; fv = 0.0;
    for (int p = n; p >= 0; --p) {
        const double* row = c + (p * (SPC_SIP_MAX_ORDER + 1) - p * (p - 1) / 2);
        double r = 0.0, rv = 0.0;
        for (int q = n - p; q >= 0; --q) { rv = rv * v + r; r = r * v + row[q]; }
        fu = fu * u + f;          // d/du of (f * u + r)
        f = f * u + r;
        fv = fv * u + rv;
    }
}

__global__ __launch_bounds__(256) void wcs_pixel_map_kernel(const WcsPair A) {
    const int64_t x = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t y = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= A.nx || y >= A.ny) return;
    const double D2R = 0.017453292519943295, R2D = 57.29577951308232, PI = 3.141592653589793;
    // target pixel -> native spherical -> celestial
    double dx = (double)x + 1.0 - A.o.crpix[0], dy = (double)y + 1.0 - A.o.crpix[1];
    if (A.o.sip_order > 0) {       // astropy's all_pix2world: the SIP polynomials, then wcslib's core
        double f, g, t0, t1;
        sip_eval(A.o.sip_a, A.o.sip_order, dx, dy, f, t0, t1);
        sip_eval(A.o.sip_b, A.o.sip_order, dx, dy, g, t0, t1);
        dx += f; dy += g;
    }
    const double px = A.o.lin[0] * dx + A.o.lin[1] * dy + A.o.plane0[0];
    const double py = A.o.lin[2] * dx + A.o.lin[3] * dy + A.o.plane0[1];
    double phi, theta;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    if (A.o.proj >= 5) {        // (pseudo-)cylindrical: CAR, SFL, CEA, MER, AIT
        const double xr = px * D2R, yr = py * D2R;
        switch (A.o.proj) {
            case 5: phi = xr; theta = yr; break;
            case 6: { theta = yr; const double c = cos(theta); phi = c != 0.0 ? xr / c : (px == 0.0 ? 0.0 : nan); break; }
            case 7: { const double s_ = A.o.pv1 * yr; theta = fabs(s_) <= 1.0 + 1e-13 ? asin(fmin(fmax(s_, -1.0), 1.0)) : nan; phi = xr; break; }
            case 8: theta = 2.0 * atan(exp(yr)) - PI / 2; phi = xr; break;
            default: {
                const double z2 = 1.0 - (xr / 4.0) * (xr / 4.0) - (yr / 2.0) * (yr / 2.0);
                const double zz = z2 >= 0.5 - 1e-13 ? sqrt(fmax(z2, 0.5)) : nan;           // outside the ellipse: not on the sky
                phi = 2.0 * atan2(zz * xr / 2.0, 2.0 * zz * zz - 1.0);
                theta = asin(fmin(fmax(yr * zz, -1.0), 1.0));
            }
        }
        // the native sphere ends at |phi| = 180, |theta| = 90 (wcslib's bounds check)
        if (fabs(phi) > PI * (1 + 1e-12) || fabs(theta) > PI / 2 * (1 + 1e-12)) phi = nan;
    } else {
        const double rr = hypot(px, py) * D2R;
        phi = atan2(px, -py);
        switch (A.o.proj) {
            case 0: theta = atan2(1.0, rr); break;
            case 1: theta = acos(fmin(fmax(rr, -1.0), 1.0)); break;
            case 2: theta = PI / 2 - rr; break;
            case 3: theta = PI / 2 - 2.0 * atan(rr / 2.0); break;
            default: theta = PI / 2 - 2.0 * asin(fmin(fmax(rr / 2.0, -1.0), 1.0)); break;
        }
    }
    double st = sin(theta), ct = cos(theta);
    double dphi = phi - A.o.phi_p;
    double sdp = sin(A.o.delta_p), cdp = cos(A.o.delta_p);
    // celestial unit vector about the native pole's meridian; the latitude from atan2 of its components (asin of the third
    // alone loses half the digits near the celestial poles: 1e-5 pixel at declination 89.9 - wcslib switches to acos there)
    const double xc = st * cdp - ct * sdp * cos(dphi), yc = -ct * sin(dphi), zc = st * sdp + ct * cdp * cos(dphi);
    double lon = A.o.alpha_p + atan2(yc, xc);
    double lat = atan2(zc, hypot(xc, yc));
    double lon_deg = fmod(lon * R2D, 360.0);
    if (lon_deg < 0.0) lon_deg += 360.0;
    lon = lon_deg * D2R;
    lat = (lat * R2D) * D2R;
    if (A.has_rot) {       // target frame -> source frame (wcs.py::frame_transform): a rotation of the unit vector, with the
                           // E-terms of aberration taken off / put on where a side is FK4
        const double cla = cos(lat);
        double vx = cla * cos(lon), vy = cla * sin(lon), vz = sin(lat);
        if (A.has_rot & 2) {
            const double dv = A.et[0] * vx + A.et[1] * vy + A.et[2] * vz;
            const double ax = vx - A.et[0] + dv * vx, ay = vy - A.et[1] + dv * vy, az = vz - A.et[2] + dv * vz;
            const double n = sqrt(ax * ax + ay * ay + az * az);
            vx = ax / n; vy = ay / n; vz = az / n;
        }
        double wx = A.rot[0] * vx + A.rot[1] * vy + A.rot[2] * vz;
        double wy = A.rot[3] * vx + A.rot[4] * vy + A.rot[5] * vz;
        double wz = A.rot[6] * vx + A.rot[7] * vy + A.rot[8] * vz;
        if (A.has_rot & 4) {
            const double x0 = wx, y0 = wy, z0 = wz;
            for (int it = 0; it < 10; ++it) {
                const double den = 1.0 + A.et[3] * wx + A.et[4] * wy + A.et[5] * wz;
                wx = (A.et[3] + x0) / den; wy = (A.et[4] + y0) / den; wz = (A.et[5] + z0) / den;
            }
            const double n = sqrt(wx * wx + wy * wy + wz * wz);
            wx /= n; wy /= n; wz /= n;
        }
        double l2 = fmod(atan2(wy, wx) * R2D, 360.0);
        if (l2 < 0.0) l2 += 360.0;
        lon = l2 * D2R;
        lat = (atan2(wz, hypot(wx, wy)) * R2D) * D2R;
    }
    // celestial -> native unit vector of the source -> source pixel
    const double da = lon - A.i.alpha_p;
    const double sl = sin(lat), cl = cos(lat);
    sdp = sin(A.i.delta_p); cdp = cos(A.i.delta_p);
    const double xn = -cl * sin(da);
    const double yn = sl * cdp - cl * sdp * cos(da);
    const double zn = sl * sdp + cl * cdp * cos(da);
    const double rho = hypot(xn, yn);
    double ph = A.i.phi_p + atan2(xn, yn);
    double ix, iy;
    if (A.i.proj >= 5) {
        ph = fmod(ph + PI, 2 * PI);
        if (ph < 0.0) ph += 2 * PI;
        ph -= PI;
        const double th = atan2(zn, rho);
        switch (A.i.proj) {
            case 5: ix = ph * R2D; iy = th * R2D; break;
            case 6: ix = ph * cos(th) * R2D; iy = th * R2D; break;
            case 7: ix = ph * R2D; iy = R2D * sin(th) / A.i.pv1; break;
            case 8: ix = ph * R2D; iy = fabs(th) < PI / 2 ? R2D * log(tan(PI / 4 + th / 2)) : nan; break;
            default: {
                const double gam = R2D * sqrt(2.0 / (1.0 + cos(th) * cos(ph / 2.0)));
                ix = 2.0 * gam * cos(th) * sin(ph / 2.0); iy = gam * sin(th);
            }
        }
    } else {
        double r;
        switch (A.i.proj) {
            case 0: r = zn > 0 ? R2D * rho / zn : nan; break;
            case 1: r = zn >= 0 ? R2D * rho : nan; break;
            case 2: r = R2D * atan2(rho, zn); break;
            case 3: r = 2.0 * R2D * rho / (1.0 + zn); break;
            default: r = 2.0 * R2D * rho / sqrt(2.0 * (1.0 + zn)); break;
        }
        ix = r * sin(ph); iy = -r * cos(ph);
    }
    ix -= A.i.plane0[0]; iy -= A.i.plane0[1];
    double su = A.i.lin_inv[0] * ix + A.i.lin_inv[1] * iy;
    double sv = A.i.lin_inv[2] * ix + A.i.lin_inv[3] * iy;
    if (A.i.sip_order > 0) {
        // astropy's all_world2pix inverts the FORWARD polynomials (a fixed-point iteration stopped at 1e-4 pixel); this is
        // Newton's method on the same equations, to 1e-13: u + A(u, v) = su, v + B(u, v) = sv.  No convergence (far outside
        // the image, where the polynomial folds over): not a pixel of the source.
        double u = su, v = sv;
        bool done = false;
        for (int it = 0; it < 50 && !done; ++it) {
            double f, fu, fv, g, gu, gv;
            sip_eval(A.i.sip_a, A.i.sip_order, u, v, f, fu, fv);
            sip_eval(A.i.sip_b, A.i.sip_order, u, v, g, gu, gv);
            const double r0 = u + f - su, r1 = v + g - sv;
            const double j00 = 1.0 + fu, j01 = fv, j10 = gu, j11 = 1.0 + gv;
            const double det = j00 * j11 - j01 * j10;
            const double du = (j11 * r0 - j01 * r1) / det, dv = (j00 * r1 - j10 * r0) / det;
            u -= du; v -= dv;
            done = fabs(du) <= 1e-13 * fmax(1.0, fabs(u)) && fabs(dv) <= 1e-13 * fmax(1.0, fabs(v));     // false for NaN
        }
        su = done ? u : nan; sv = done ? v : nan;
    }
    double sx = su + A.i.crpix[0] - 1.0;
    double sy = sv + A.i.crpix[1] - 1.0;
    const bool fin = (fabs(sx) <= 1.79e308) && (fabs(sy) <= 1.79e308);      // false for NaN / Inf
    A.xs[y * A.nx + x] = fin ? sx : -1e30;
    A.ys[y * A.nx + x] = fin ? sy : -1e30;
}

// ---- spline resampling (order 2 / 3): scipy.ndimage.map_coordinates on the edge-padded planes ------------------------
// reproject_interp(order='biquadratic' | 'bicubic') (the orders BaseSpectralCube.reproject documents, spectral_cube.py:
// 2667-2676): replicate the border by one pixel, B-spline prefilter with MIRROR boundaries along y and x (scipy filters
// along z as well; sampled at integer channels that filter is undone exactly), then a 3 x 3 / 4 x 4 gather with
// mirror-folded support indices.  All float64, like scipy (oracle/oracle_np.py::resample_spline restates the same steps and
// is pinned against scipy).  The coefficients of a slab of planes live in the caller's workspace as float64.
struct SplArgs {
    const float* cube; int64_t nz, ny, nx, row_stride, plane_stride;
    double* coef;                 // (nz, ny + 2, nx + 2)
    int order; double pole;
    int64_t ny_out, nx_out; const double* xs; const double* ys;
    float* out; int64_t out_row_stride, out_plane_stride;
    uint8_t* footprint;
};

// initial value of the causal recursion for a mirror boundary (ni_splines.c::_init_causal_mirror): the sum runs over the
// whole line in scipy; |pole|^k < 1e-17 beyond 64 terms (|pole| <= 0.268), so the loop stops there and the reflected term
// pole^(n - 1) only takes part for lines short enough for it to matter
template <class Get>
__device__ __forceinline__ double spline_causal_init(Get get, int64_t n, double z) {
    const bool short_line = n <= 66;
    const double z_n_1 = short_line ? pow(z, (double)(n - 1)) : 0.0;
    double c0 = get(0) + z_n_1 * (short_line ? get(n - 1) : 0.0);
    double z_i = z;
    const int64_t last = short_line ? n - 1 : 65;
    for (int64_t i = 1; i < last; ++i) {
        c0 += z_i * (get(i) + (short_line ? z_n_1 * get(n - 1 - i) : 0.0));
        z_i *= z;
    }
    return c0 / (1.0 - z_n_1 * z_n_1);
}

// along y: one thread per padded column, rows in sequence (a wave reads / writes 64 consecutive columns of a row)
__global__ __launch_bounds__(256) void spline_y_kernel(const SplArgs A) {
    const int64_t xp = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t z = blockIdx.y;
    const int64_t nyp = A.ny + 2, nxp = A.nx + 2;
    if (xp >= nxp) return;
    const int64_t xc = min(max(xp - 1, (int64_t)0), A.nx - 1);
    const float* src = A.cube + z * A.plane_stride + xc;
    double* c = A.coef + z * nyp * nxp + xp;
    const double zp = A.pole, gain = (1.0 - zp) * (1.0 - 1.0 / zp);
    auto get = [&](int64_t i) { return gain * (double)src[min(max(i - 1, (int64_t)0), A.ny - 1) * A.row_stride]; };
    double prev = spline_causal_init(get, nyp, zp);
    c[0] = prev;
    for (int64_t i = 1; i < nyp; ++i) {
        prev = get(i) + zp * prev;
        c[i * nxp] = prev;
    }
    double nxt = (zp * c[(nyp - 2) * nxp] + prev) * zp / (zp * zp - 1.0);
    c[(nyp - 1) * nxp] = nxt;
    for (int64_t i = nyp - 2; i >= 0; --i) {
        nxt = zp * (nxt - c[i * nxp]);
        c[i * nxp] = nxt;
    }
}

// along x, in place: a block takes 64 padded rows and walks them in 64-column tiles through LDS (coalesced tile loads, a
// thread per row inside the tile); the causal carry goes left to right, the anticausal one back
__global__ __launch_bounds__(64) void spline_x_kernel(const SplArgs A) {
    __shared__ double tile[64][65];
    const int t = threadIdx.x;
    const int64_t nyp = A.ny + 2, nxp = A.nx + 2;
    const int64_t r0 = (int64_t)blockIdx.x * 64, z = blockIdx.y;
    double* base = A.coef + z * nyp * nxp;
    const int64_t nrows = min((int64_t)64, nyp - r0);
    const double zp = A.pole, gain = (1.0 - zp) * (1.0 - 1.0 / zp);
    const int64_t myrow = r0 + min((int64_t)t, nrows - 1);
    double* line = base + myrow * nxp;
    auto get = [&](int64_t i) { return gain * line[i]; };
    double carry = spline_causal_init(get, nxp, zp);            // (strided reads of <= 66 samples per row)
    const int64_t ntile = (nxp + 63) / 64;
    for (int64_t k = 0; k < ntile; ++k) {
        const int64_t x0 = k * 64, w = min((int64_t)64, nxp - x0);
        for (int64_t r = 0; r < nrows; ++r) if (t < w) tile[r][t] = base[(r0 + r) * nxp + x0 + t];
        __syncthreads();
        if (t < nrows) {
            for (int64_t i = 0; i < w; ++i) {
                carry = (x0 + i == 0) ? carry : gain * tile[t][i] + zp * carry;
                tile[t][i] = carry;
            }
        }
        __syncthreads();
        for (int64_t r = 0; r < nrows; ++r) if (t < w) base[(r0 + r) * nxp + x0 + t] = tile[r][t];
        __syncthreads();
    }
    double nxt = 0.0;
    for (int64_t k = ntile - 1; k >= 0; --k) {
        const int64_t x0 = k * 64, w = min((int64_t)64, nxp - x0);
        for (int64_t r = 0; r < nrows; ++r) if (t < w) tile[r][t] = base[(r0 + r) * nxp + x0 + t];
        __syncthreads();
        if (t < nrows) {
            for (int64_t i = w - 1; i >= 0; --i) {
                if (x0 + i == nxp - 1) nxt = (zp * line[nxp - 2] + tile[t][i]) * zp / (zp * zp - 1.0);
                else nxt = zp * (nxt - tile[t][i]);
                tile[t][i] = nxt;
            }
        }
        __syncthreads();
        for (int64_t r = 0; r < nrows; ++r) if (t < w) base[(r0 + r) * nxp + x0 + t] = tile[r][t];
        __syncthreads();
    }
}

// support start and weights of scipy's get_spline_interpolation_weights
__device__ __forceinline__ int64_t spline_weights(double x, int order, double* w) {
    if (order == 3) {
        const double f = floor(x), y = x - f, zz = 1.0 - y;
        w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
        w[2] = (zz * zz * (zz - 2.0) * 3.0 + 4.0) / 6.0;
        w[0] = zz * zz * zz / 6.0;
        w[3] = 1.0 - w[0] - w[1] - w[2];
        return (int64_t)f - 1;
    }
    const double f = floor(x + 0.5), y = x - f, tt = 0.5 - y;
    w[1] = 0.75 - y * y;
    w[0] = 0.5 * tt * tt;
    w[2] = 1.0 - w[0] - w[1];
    w[3] = 0.0;
    return (int64_t)f - 1;
}

__global__ __launch_bounds__(256) void spline_gather_kernel(const SplArgs A) {
    const int64_t x = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t y = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= A.nx_out || y >= A.ny_out) return;
    const double sx = A.xs[y * A.nx_out + x], sy = A.ys[y * A.nx_out + x];
    const bool inside = sx >= -0.5 && sx <= (double)A.nx - 0.5 && sy >= -0.5 && sy <= (double)A.ny - 0.5;     // false for NaN
    if (A.footprint && blockIdx.z == 0) A.footprint[y * A.nx_out + x] = inside ? 1 : 0;
    const float nanf = __int_as_float(0x7fc00000);
    const int64_t nyp = A.ny + 2, nxp = A.nx + 2;
    double wx[4], wy[4];
    int64_t ix[4], iy[4];
    const int nt = A.order + 1;
    if (inside) {
        const int64_t x0 = spline_weights(sx + 1.0, A.order, wx), y0 = spline_weights(sy + 1.0, A.order, wy);
        for (int k = 0; k < 4; ++k) {                      // mirror about the first / last sample
            int64_t i = x0 + k; i = i < 0 ? -i : i; ix[k] = min(i > nxp - 1 ? 2 * (nxp - 1) - i : i, nxp - 1);
            int64_t j = y0 + k; j = j < 0 ? -j : j; iy[k] = min(j > nyp - 1 ? 2 * (nyp - 1) - j : j, nyp - 1);
        }
    }
    for (int64_t z = blockIdx.z; z < A.nz; z += gridDim.z) {
        float v = nanf;
        if (inside) {
            const double* c = A.coef + z * nyp * nxp;
            double acc = 0.0;
            for (int j = 0; j < nt; ++j) {
                const double* row = c + iy[j] * nxp;
                double r = 0.0;
                for (int i = 0; i < nt; ++i) r += wx[i] * row[ix[i]];
                acc += wy[j] * r;
            }
            v = (float)acc;
        }
        A.out[z * A.out_plane_stride + y * A.out_row_stride + x] = v;
    }
}

}  // namespace

extern "C" {

int spc_spectral_lerp_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                          int64_t nz_out, const int32_t* d_lo, const double* d_t, const double* d_inv_dx,
                          float fill, float* d_out, int64_t out_row_stride, int64_t out_plane_stride) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(nz_out > 0, "nz_out must be positive");
    SPC_REQUIRE(d_lo && d_t && d_inv_dx && d_out, "NULL pointer argument");
    SPC_REQUIRE(cube->nz >= 2, "need at least 2 input channels");
    LerpArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.nz_out = nz_out; A.lo = d_lo; A.t = d_t; A.inv_dx = d_inv_dx; A.fill = fill;
    A.out = d_out;
    A.out_row_stride = out_row_stride ? out_row_stride : cube->nx;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : cube->ny * A.out_row_stride;
    const bool v4 = (cube->nx % 4 == 0) && (cube->row_stride % 4 == 0) && (cube->plane_stride % 4 == 0) &&
                    (A.out_row_stride % 4 == 0) && (A.out_plane_stride % 4 == 0) &&
                    (((uintptr_t)cube->d_data) % 16 == 0) && (((uintptr_t)d_out) % 16 == 0);
    const int64_t ncols = cube->ny * (cube->nx / (v4 ? 4 : 1));
    const int64_t nblocks = (ncols + 255) / 256;
    int nsplit = 1;
    if (nblocks < 2048) nsplit = (int)std::max<int64_t>(1, std::min<int64_t>((2048 + nblocks - 1) / nblocks, nz_out / 16));
    A.jchunk = (nz_out + nsplit - 1) / nsplit;
    A.nt_store = 1;
    { const char* e = getenv("SPC_LERP_JCHUNK"); if (e && atoi(e) > 0) A.jchunk = atoi(e); e = getenv("SPC_LERP_NT"); if (e) A.nt_store = atoi(e); }
    nsplit = (int)((nz_out + A.jchunk - 1) / A.jchunk);
    SPC_REQUIRE(nsplit <= 65535, "too many channel groups for one launch");
    {   // short-lived blocks (see spectral_lerp_tiles_kernel): 16-byte rows, enough work to fill the chip; SPC_LERP_TILES=0 keeps the march
        const char* e = getenv("SPC_LERP_TILES");
        const int want = e ? atoi(e) : 1;
        constexpr int JC = 4;
        const int64_t bpr = (cube->nx / 4 + 255) / 256, groups = (nz_out + JC - 1) / JC;
        if (want && v4 && cube->nz >= 3 && groups <= 65535 && bpr * cube->ny < (1ll << 31) && bpr * cube->ny * groups >= 4096) {
            hipLaunchKernelGGL((spectral_lerp_tiles_kernel<JC>), dim3((unsigned)(bpr * cube->ny), (unsigned)groups), dim3(256), 0,
                               (hipStream_t)stream, A, (int)bpr);
            SPC_LAUNCH_CHECK();
            return SPC_OK;
        }
    }
    dim3 grid((unsigned)nblocks, (unsigned)nsplit);
    if (v4) hipLaunchKernelGGL((spectral_lerp_kernel<4, 1>), grid, dim3(256), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL((spectral_lerp_kernel<1, 1>), grid, dim3(256), 0, (hipStream_t)stream, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_wcs_pixel_map_f64(int device, void* stream, const spc_celestial_wcs* wcs_out, const spc_celestial_wcs* wcs_in,
                          const double* frame_rot, int64_t ny_out, int64_t nx_out, double* d_xs, double* d_ys) {
    SPC_REQUIRE(wcs_out && wcs_in && d_xs && d_ys, "NULL pointer argument");
    SPC_REQUIRE(ny_out > 0 && nx_out > 0, "output shape must be positive");
    SPC_REQUIRE(wcs_out->proj >= 0 && wcs_out->proj <= 9 && wcs_in->proj >= 0 && wcs_in->proj <= 9, "unknown projection code");
    SPC_REQUIRE(wcs_out->sip_order >= 0 && wcs_out->sip_order <= SPC_SIP_MAX_ORDER && wcs_in->sip_order >= 0 &&
                wcs_in->sip_order <= SPC_SIP_MAX_ORDER, "SIP order must be 0 (none) .. 9");
    SPC_REQUIRE((ny_out + 3) / 4 <= 65535, "too many rows for one launch");
    SPC_DEVICE(device);
    WcsPair A{*wcs_out, *wcs_in, ny_out, nx_out, d_xs, d_ys, 0, {1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0, 0, 0, 0}};
    if (frame_rot) {        // ABI 4: 15 doubles - the rotation, the E-terms removed before it, the E-terms added after it
        A.has_rot = 1;
        for (int k = 0; k < 9; ++k) A.rot[k] = frame_rot[k];
        for (int k = 0; k < 6; ++k) A.et[k] = frame_rot[9 + k];
        if (A.et[0] != 0.0 || A.et[1] != 0.0 || A.et[2] != 0.0) A.has_rot |= 2;
        if (A.et[3] != 0.0 || A.et[4] != 0.0 || A.et[5] != 0.0) A.has_rot |= 4;
    }
    hipLaunchKernelGGL(wcs_pixel_map_kernel, dim3((unsigned)((nx_out + 63) / 64), (unsigned)((ny_out + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_resample_spline_f32(int device, void* stream, const spc_cube_f32* cube, int order, int64_t ny_out, int64_t nx_out,
                            const double* d_xs, const double* d_ys, float* d_out, int64_t out_row_stride,
                            int64_t out_plane_stride, uint8_t* d_footprint, void* d_workspace, size_t workspace_bytes) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(ny_out > 0 && nx_out > 0, "output shape must be positive");
    SPC_REQUIRE(d_xs && d_ys && d_out, "NULL pointer argument");
    SPC_REQUIRE(order == 2 || order == 3, "order must be 2 (biquadratic) or 3 (bicubic), got %d", order);
    SPC_REQUIRE(cube->ny >= 1 && cube->nx >= 1, "empty plane");
    const size_t need = (size_t)cube->nz * (size_t)(cube->ny + 2) * (size_t)(cube->nx + 2) * sizeof(double);
    SPC_REQUIRE(d_workspace && workspace_bytes >= need, "workspace too small: %zu bytes, need %zu (nz x (ny + 2) x (nx + 2) float64)",
                workspace_bytes, need);
    SPC_REQUIRE((ny_out + 3) / 4 <= 65535 && cube->nz <= 65535, "too many rows / channels for one launch (resample a slab of channels)");
    SPC_DEVICE(device);
    SplArgs A{};
    A.cube = cube->d_data; A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.coef = (double*)d_workspace; A.order = order;
    A.pole = order == 3 ? sqrt(3.0) - 2.0 : sqrt(8.0) - 3.0;
    A.ny_out = ny_out; A.nx_out = nx_out; A.xs = d_xs; A.ys = d_ys; A.out = d_out;
    A.out_row_stride = out_row_stride ? out_row_stride : nx_out;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : ny_out * A.out_row_stride;
    A.footprint = d_footprint;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(spline_y_kernel, dim3((unsigned)((cube->nx + 2 + 255) / 256), (unsigned)cube->nz), dim3(256), 0, st, A);
    SPC_LAUNCH_CHECK();
    hipLaunchKernelGGL(spline_x_kernel, dim3((unsigned)((cube->ny + 2 + 63) / 64), (unsigned)cube->nz), dim3(64), 0, st, A);
    SPC_LAUNCH_CHECK();
    const unsigned zs = (unsigned)std::min<int64_t>(cube->nz, 64);
    hipLaunchKernelGGL(spline_gather_kernel, dim3((unsigned)((nx_out + 63) / 64), (unsigned)((ny_out + 3) / 4), zs), dim3(256), 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

// spc_resample_bilinear_f32 (nz_out = 0) and spc_resample_bilinear_lerp_f32 (the spectral interpolation folded in)
static int bilinear_launch(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                           float fill, int64_t ny_out, int64_t nx_out, const double* d_xs,
                           const double* d_ys, int64_t nz_out, const int32_t* d_lo, const double* d_t, const double* d_inv_dx,
                           float* d_out, int64_t out_row_stride,
                           int64_t out_plane_stride, uint8_t* d_footprint, int order, uint32_t* d_any_valid,
                           void* d_workspace, size_t workspace_bytes) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    const bool lerp = nz_out > 0;
    SPC_REQUIRE(ny_out > 0 && nx_out > 0, "output shape must be positive");
    SPC_REQUIRE(d_xs && d_ys && d_out, "NULL pointer argument");
    SPC_REQUIRE(order == 0 || order == 1, "order must be 1 (bilinear) or 0 (nearest neighbour), got %d", order);
    if (lerp) {
        SPC_REQUIRE(d_lo && d_t && d_inv_dx, "NULL pointer argument (lerp plan)");
        SPC_REQUIRE(cube->nz >= 2, "the spectral interpolation needs at least two input channels");
        SPC_REQUIRE(cube->nz < (1ll << 30) && nz_out < (1ll << 30), "too many channels");
    }
    BilArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.fill = fill; A.ny_out = ny_out; A.nx_out = nx_out; A.xs = d_xs; A.ys = d_ys;
    A.out = d_out;
    A.out_row_stride = out_row_stride ? out_row_stride : nx_out;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : ny_out * A.out_row_stride;
    A.footprint = d_footprint;
    A.nearest = order == 0; A.any_valid = d_any_valid;
    A.nz_out = nz_out; A.lo = d_lo; A.t = d_t; A.inv_dx = d_inv_dx;
    hipStream_t st = (hipStream_t)stream;
    SpcWorkspace ws(d_workspace, workspace_bytes);
    if (lerp) {
        SPC_WS_TAKE(d_jfirst, ws, int32_t, cube->nz);
        hipLaunchKernelGGL(lerp_index_kernel, dim3(1), dim3(256), 0, st, d_lo, (int)nz_out, (int)cube->nz, d_jfirst);
        SPC_LAUNCH_CHECK();
        A.jfirst = d_jfirst;
    }
    const int64_t nzw = lerp ? cube->nz - 1 : cube->nz;                     // units of work along z: planes, or left brackets
    if (d_any_valid) SPC_HIP(hipMemsetAsync(d_any_valid, 0, sizeof(uint32_t), st));
    const int64_t nblocks = ((nx_out + 63) / 64) * ((ny_out + 3) / 4);      // 64 x 4 pixel tiles
    int nsplit = 1;
    if (nblocks < 2048) nsplit = (int)std::max<int64_t>(1, std::min<int64_t>((2048 + nblocks - 1) / nblocks, nzw / 8));
    A.zchunk = (nzw + nsplit - 1) / nsplit;
    nsplit = (int)((nzw + A.zchunk - 1) / A.zchunk);
    // LDS-staged pass over 32 x 32 tiles first; the gather kernel then only does flagged tiles
    const char* env = getenv("SPC_BILINEAR_LDS");
    const bool want = env ? atoi(env) != 0 : true;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const bool fits = cube->ny * cube->row_stride < (1ll << 31) &&
                      (!arr || cube->ny * A.mask.row_stride < (1ll << 31));
    A.status = nullptr;
    if (want && fits) {
        const char* tenv = getenv("SPC_BILINEAR_TILE");
        const int tile = tenv ? (atoi(tenv) == 32 ? 32 : 64) : ((nx_out >= 128 && ny_out >= 128) ? 64 : 32);
        const int kStageU = tile == 64 ? BilTile<64>::kStageU : BilTile<32>::kStageU;
        A.tile_shift = tile == 64 ? 6 : 5;
        A.tiles32_x = (nx_out + tile - 1) / tile;
        const int64_t ntiles = A.tiles32_x * ((ny_out + tile - 1) / tile);
        A.ntiles32 = ntiles;
        int ns = 1;
        const int64_t want_blocks = tile == 64 ? 1024 : 4096;
        if (ntiles < want_blocks) ns = (int)std::max<int64_t>(1, std::min<int64_t>((want_blocks + ntiles - 1) / ntiles, nzw / 64));
        // at most 256 channels per block (C5, 32 x 32 tiles: 9.4 ms with 1024-channel chunks, 8.5 ms with 128 - 256, 9.4 with 64
        // where the per-block footprint set-up starts to show).  The gain is scheduling - more, shorter blocks even out the
        // tail - not cache reuse: FETCH_SIZE did not move (profiles/r01_pmc_traffic.txt).
        // (LERP: 512 input planes per block - C5 in one pass 6.6 / 6.2 / 6.17 / 6.1 ms with 64 / 128 / 256 / 512)
        A.zchunk_lds = std::min<int64_t>((nzw + ns - 1) / ns, lerp ? 512 : 256);
        if (const char* zc = getenv("SPC_BILINEAR_ZCHUNK")) A.zchunk_lds = std::max(8, atoi(zc));
        A.zchunk_lds = ((A.zchunk_lds + kStageU - 1) / kStageU) * kStageU;
        ns = (int)((nzw + A.zchunk_lds - 1) / A.zchunk_lds);
        SPC_WS_TAKE(d_status, ws, unsigned char, ntiles);
        SPC_HIP(spc_flags_clear(d_status, (size_t)ntiles, st));
        A.status = d_status;
        dim3 g((unsigned)(((ntiles + 7) / 8) * 8), (unsigned)ns);
#define SPC_BIL_LAUNCH(T_, L_)                                                                                       \
        do {                                                                                                         \
            dim3 b(BilTile<T_>::kThreads);                                                                           \
            if (arr) hipLaunchKernelGGL((bilinear_lds_kernel<T_, true, true, L_>), g, b, 0, st, A);                   \
            else if (A.mask.flags) hipLaunchKernelGGL((bilinear_lds_kernel<T_, false, true, L_>), g, b, 0, st, A);    \
            else hipLaunchKernelGGL((bilinear_lds_kernel<T_, false, false, L_>), g, b, 0, st, A);                     \
        } while (0)
        if (tile == 64) { if (lerp) SPC_BIL_LAUNCH(64, true); else SPC_BIL_LAUNCH(64, false); }
        else { if (lerp) SPC_BIL_LAUNCH(32, true); else SPC_BIL_LAUNCH(32, false); }
#undef SPC_BIL_LAUNCH
        SPC_LAUNCH_CHECK();
    }
    if (lerp) hipLaunchKernelGGL(bilinear_kernel<true>, dim3((unsigned)nblocks, (unsigned)nsplit), dim3(256), 0, st, A);
    else hipLaunchKernelGGL(bilinear_kernel<false>, dim3((unsigned)nblocks, (unsigned)nsplit), dim3(256), 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_resample_bilinear_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                              float fill, int64_t ny_out, int64_t nx_out, const double* d_xs,
                              const double* d_ys, float* d_out, int64_t out_row_stride,
                              int64_t out_plane_stride, uint8_t* d_footprint, int order, uint32_t* d_any_valid,
                              void* d_workspace, size_t workspace_bytes) {
    return bilinear_launch(device, stream, cube, mask, fill, ny_out, nx_out, d_xs, d_ys, 0, nullptr, nullptr, nullptr, d_out,
                           out_row_stride, out_plane_stride, d_footprint, order, d_any_valid, d_workspace, workspace_bytes);
}

int spc_resample_bilinear_lerp_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                                   float fill, int64_t ny_out, int64_t nx_out, const double* d_xs, const double* d_ys,
                                   int64_t nz_out, const int32_t* d_lo, const double* d_t, const double* d_inv_dx,
                                   float* d_out, int64_t out_row_stride, int64_t out_plane_stride, uint8_t* d_footprint,
                                   int order, uint32_t* d_any_valid, void* d_workspace, size_t workspace_bytes) {
    SPC_REQUIRE(nz_out > 0, "nz_out must be positive");
    return bilinear_launch(device, stream, cube, mask, fill, ny_out, nx_out, d_xs, d_ys, nz_out, d_lo, d_t, d_inv_dx, d_out,
                           out_row_stride, out_plane_stride, d_footprint, order, d_any_valid, d_workspace, workspace_bytes);
}

}  // extern "C"

size_t spc_ws_resample_bilinear(int64_t ny_out, int64_t nx_out) {
    return spc_ws_round((size_t)(((nx_out + 31) / 32) * ((ny_out + 31) / 32))) + 256;     // tile flags (32 x 32: the finer grid)
}
size_t spc_ws_resample_bilinear_lerp(int64_t nz, int64_t ny_out, int64_t nx_out) {
    return spc_ws_resample_bilinear(ny_out, nx_out) + spc_ws_round(sizeof(int32_t) * (size_t)nz) + 256;     // + jfirst[]
}
