// Fused masked moment 0/1/2 (+argmax/argmin/max/min/count) along the spectral
// axis of a (nz, ny, nx) float32 cube, for gfx950.
//
// Replaces (see include/spcube_hip.h for the full list):
//   spectral_cube/dask_spectral_cube.py:1083-1104, :54-59
//   spectral_cube/_moments.py:30-193, spectral_cube/np_compat.py:3-27
//   spectral_cube/spectral_cube.py:793-819 (argmax/argmin)
//
// Mapping: x is the fastest axis, so one lane owns VEC consecutive spaxels and
// marches over z; a wave reads 64*VEC*4 B contiguous per plane (1 KiB for
// VEC=4).  All sums are carried in fp64 registers (the reference computes in
// fp64; one-pass moment 2 from raw sums needs it).  The cube is read exactly
// once: 4 B/voxel (+1 B/voxel with a uint8 mask array).
//
// Parallelism: blockDim = (64, ZW): the ZW waves of a block take interleaved
// planes of the same 64*VEC columns and are combined through LDS; gridDim.y
// splits z further when the map is too small to fill 256 CUs, with a second
// tiny combine kernel over fp64 partial sums.
//
// Issue budget (round 4, tools/micro/read_patterns.hip): this march reads 6.7 - 7.1 TB/s when nothing is
// computed, so the instructions per voxel decide whether the kernel is bound by HBM or by issue.  The mask
// predicate is therefore a template parameter in the canonical form of spc_canonical_pred (one compare when
// the mask has no threshold term, three otherwise - never the five flag-guarded compares of spc_pred), the
// valid count is only carried when it is asked for (otherwise one scalar "any valid" bit per column), the
// include select happens in float32 before the widening, the channel coordinate of a plane is a scalar load
// (the wave index goes through readfirstlane, so a plane's address and coordinate are provably uniform).
// 27.7 -> 12.4 instructions per voxel; the kernel now runs within 2 - 7 % of the same two streams read with no
// arithmetic at all (tools/micro/mask_patterns.hip), which is what is left of the 8 TB/s in this access pattern
// (profiles/r04_moments_issue.log).  SPC_MOMENTS_XCD=1 hands the blocks of one XCD neighbouring column groups
// (+3 % on the bare read, nothing on the kernel: off).
#include "spc_common.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int kLanes = 64;

struct MomArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    const double* cen;
    double dv, m1_add;
    spc_moment_outputs out;
    int64_t out_row_stride;
    int64_t groups_per_row;  // nx / VEC
    int64_t ngroups;         // ny * groups_per_row
    int nsplit;              // gridDim.y
    int64_t zchunk;          // planes per split
    double* ws;              // partial sums workspace (nsplit > 1)
    float lim, lo, hi;       // the mask predicate in canonical form: |v| <= lim && !(v <= lo) && !(v >= hi)
    int xcd_group;           // blocks of one XCD take neighbouring column groups
};

// per-column running state
struct Acc {
    double s0, s1, s2;
    int n;
    bool any;                // EXT == 0: "a valid sample was met" instead of the count (a lane mask in scalar registers)
    float bmax, bmin;
    int imax, imin;
};

// EXT: 0 = the three sums (+ any-valid bit), 1 = + valid count, 2 = + extrema and their channels
constexpr int kSums = 0, kCount = 1, kExtrema = 2;

__device__ __forceinline__ void acc_init(Acc& a, int z0) {
    a.s0 = a.s1 = a.s2 = 0.0;
    a.n = 0;
    a.any = false;
    a.bmax = -INFINITY; a.bmin = INFINITY;
    a.imax = z0; a.imin = z0;
}

// PRED 0: |v| <= lim (lim = +inf: not NaN; FLT_MAX: finite), 1: the full canonical form
template <int PRED>
__device__ __forceinline__ bool mom_pred(float v, float lim, float lo, float hi) {
    bool ok = fabsf(v) <= lim;
    if (PRED) ok = ok && !(v <= lo) && !(v >= hi);
    return ok;
}

// ok already holds every term of the mask and rejects NaN
template <int EXT>
__device__ __forceinline__ void acc_add(Acc& a, float v, bool ok, double c, double c2, int z) {
    float w = ok ? v : 0.f;                  // select in float32 (one v_cndmask), then widen:
    asm volatile("" : "+v"(w));              // left alone LLVM widens first and selects both halves
    const double wd = (double)w;
    a.s0 += wd;
    a.s1 = fma(wd, c, a.s1);
    a.s2 = fma(wd, c2, a.s2);
    if (EXT >= kCount) a.n += ok ? 1 : 0;
    else a.any = a.any || ok;
    if (EXT == kExtrema) {
        const float hi = ok ? v : -INFINITY;
        const float lo = ok ? v : INFINITY;
        if (hi > a.bmax) { a.bmax = hi; a.imax = z; }
        if (lo < a.bmin) { a.bmin = lo; a.imin = z; }
    }
}

// merge b into a; ties keep the smaller channel index (first-index rule)
template <int EXT>
__device__ __forceinline__ void acc_merge(Acc& a, const Acc& b) {
    a.s0 += b.s0; a.s1 += b.s1; a.s2 += b.s2; a.n += b.n;
    if (EXT == kExtrema) {
        if (b.bmax > a.bmax || (b.bmax == a.bmax && b.imax < a.imax)) { a.bmax = b.bmax; a.imax = b.imax; }
        if (b.bmin < a.bmin || (b.bmin == a.bmin && b.imin < a.imin)) { a.bmin = b.bmin; a.imin = b.imin; }
    }
}

__device__ __forceinline__ void finalize(const MomArgs& A, const Acc& a, int64_t y, int64_t x) {
    const int64_t o = y * A.out_row_stride + x;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const double mu = a.s1 / a.s0;  // 0/0 -> NaN for empty rays, like the reference
    if (A.out.d_m0) A.out.d_m0[o] = a.n > 0 ? A.dv * a.s0 : nan;
    if (A.out.d_m1) A.out.d_m1[o] = mu + A.m1_add;
    if (A.out.d_m2) A.out.d_m2[o] = a.s2 / a.s0 - mu * mu;
    if (A.out.d_mu) A.out.d_mu[o] = mu;
    if (A.out.d_s0) A.out.d_s0[o] = a.s0;
    if (A.out.d_argmax) A.out.d_argmax[o] = a.n > 0 ? (int64_t)a.imax : 0;
    if (A.out.d_argmin) A.out.d_argmin[o] = a.n > 0 ? (int64_t)a.imin : 0;
    if (A.out.d_vmax) A.out.d_vmax[o] = a.n > 0 ? a.bmax : NAN;
    if (A.out.d_vmin) A.out.d_vmin[o] = a.n > 0 ? a.bmin : NAN;
    if (A.out.d_nvalid) A.out.d_nvalid[o] = a.n;
}

// workspace layout for nsplit > 1: [split][field][column], 8 fields of 8 bytes
constexpr int kWsFields = 6;
__device__ __forceinline__ void ws_store(const MomArgs& A, const Acc& a, int split, int64_t col, int64_t ncols) {
    double* base = A.ws + (int64_t)split * kWsFields * ncols + col;
    base[0 * ncols] = a.s0;
    base[1 * ncols] = a.s1;
    base[2 * ncols] = a.s2;
    base[3 * ncols] = __longlong_as_double((long long)a.n);
    base[4 * ncols] = __longlong_as_double(((long long)__float_as_int(a.bmax) << 32) | (unsigned int)a.imax);
    base[5 * ncols] = __longlong_as_double(((long long)__float_as_int(a.bmin) << 32) | (unsigned int)a.imin);
}
__device__ __forceinline__ void ws_load(const MomArgs& A, Acc& a, int split, int64_t col, int64_t ncols) {
    const double* base = A.ws + (int64_t)split * kWsFields * ncols + col;
    a.s0 = base[0 * ncols];
    a.s1 = base[1 * ncols];
    a.s2 = base[2 * ncols];
    a.n = (int)__double_as_longlong(base[3 * ncols]);
    long long p = __double_as_longlong(base[4 * ncols]);
    a.bmax = __int_as_float((int)(p >> 32)); a.imax = (int)(p & 0xffffffffLL);
    p = __double_as_longlong(base[5 * ncols]);
    a.bmin = __int_as_float((int)(p >> 32)); a.imin = (int)(p & 0xffffffffLL);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int VEC> struct VecT;
template <> struct VecT<4> { using F = f32x4; using M = uint32_t; };   // 4 mask bytes in one dword
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <> struct VecT<2> { using F = f32x2; using M = uint16_t; };
template <> struct VecT<1> { using F = float; using M = unsigned char; };

__device__ __forceinline__ float vget(const f32x4& v, int i) { return v[i]; }
__device__ __forceinline__ float vget(const f32x2& v, int i) { return v[i]; }
__device__ __forceinline__ float vget(const float& v, int) { return v; }
__device__ __forceinline__ unsigned vget(const uint16_t& v, int i) { return (v >> (8 * i)) & 0xffu; }
__device__ __forceinline__ unsigned vget(const uint32_t& v, int i) { return (v >> (8 * i)) & 0xffu; }
__device__ __forceinline__ unsigned vget(const unsigned char& v, int) { return v; }

template <int VEC, int ZW, int U, bool ARR, int EXT, int PRED>
__global__ __launch_bounds__(kLanes * ZW) void moments_kernel(const MomArgs A) {
    using F = typename VecT<VEC>::F;
    using M = typename VecT<VEC>::M;
    const int lane = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);   // blockDim.x = 64: one wave per y
    // the hardware hands consecutive blocks to consecutive XCDs: give the blocks of one XCD one contiguous range of
    // column groups (their 1-KiB row segments then follow each other in that XCD's L2 channels and DRAM pages)
    int64_t blk = blockIdx.x;
    if (A.xcd_group) {
        const int64_t nb = gridDim.x, q = nb >> 3, r = nb & 7, k = blk & 7;
        blk = k * q + (k < r ? k : r) + (blk >> 3);
    }
    const int64_t g = blk * kLanes + lane;
    const bool live = g < A.ngroups;
    const int64_t gg = live ? g : 0;
    const int64_t y = gg / A.groups_per_row;
    const int64_t x = (gg - y * A.groups_per_row) * VEC;
    const int split = blockIdx.y;
    const int64_t zb = (int64_t)split * A.zchunk;
    const int64_t ze = min(A.nz, zb + A.zchunk);

    const float* p = A.cube + y * A.row_stride + x;
    const uint8_t* pm = ARR ? A.mask.arr + y * A.mask.row_stride + x : nullptr;
    const float lim = A.lim, lo = A.lo, hi = A.hi;
    // the channel coordinates through the constant address space: a uniform index is then ALWAYS a scalar load (left in the
    // global address space the ZW = 8 instantiation fell back to eight per-lane loads per iteration - the no-clobber analysis
    // that licenses a scalar load of global memory gives up on the larger kernel)
    typedef const double __attribute__((address_space(4))) cdouble;
    cdouble* cenp = (cdouble*)A.cen;

    Acc acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc_init(acc[i], (int)min(zb + w, ze - 1));

    if (live) {
        int64_t z = zb + w;
        // main loop: U planes (stride ZW) in flight per lane
        for (; z + (int64_t)(U - 1) * ZW < ze; z += (int64_t)U * ZW) {
            F v[U];
            M m[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t zz = z + (int64_t)u * ZW;
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const F*>(p + zz * A.plane_stride));
                if (ARR) m[u] = __builtin_nontemporal_load(reinterpret_cast<const M*>(pm + zz * A.mask.plane_stride));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t zz = z + (int64_t)u * ZW;
                const double c = cenp[zz];                       // uniform index: a scalar load
                const double c2 = c * c;
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float val = vget(v[u], i);
                    bool ok = mom_pred<PRED>(val, lim, lo, hi);
                    if (ARR) ok = ok && (vget(m[u], i) != 0);
                    acc_add<EXT>(acc[i], val, ok, c, c2, (int)zz);
                }
            }
        }
        // tail
        for (; z < ze; z += ZW) {
            const F v = *reinterpret_cast<const F*>(p + z * A.plane_stride);
            M m = 0;
            if (ARR) m = *reinterpret_cast<const M*>(pm + z * A.mask.plane_stride);
            const double c = cenp[z];
            const double c2 = c * c;
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float val = vget(v, i);
                bool ok = mom_pred<PRED>(val, lim, lo, hi);
                if (ARR) ok = ok && (vget(m, i) != 0);
                acc_add<EXT>(acc[i], val, ok, c, c2, (int)z);
            }
        }
    }
    if (EXT < kCount) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i].n = acc[i].any ? 1 : 0;
    }

    // ---- combine the ZW waves of the block through LDS (one VEC element at a time)
    if (ZW > 1) {
        __shared__ double sh[(ZW > 1 ? ZW - 1 : 1) * 6 * kLanes];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            if (w > 0) {
                double* s = sh + (w - 1) * 6 * kLanes + lane;
                s[0 * kLanes] = acc[i].s0;
                s[1 * kLanes] = acc[i].s1;
                s[2 * kLanes] = acc[i].s2;
                s[3 * kLanes] = __longlong_as_double((long long)acc[i].n);
                s[4 * kLanes] = __longlong_as_double(((long long)__float_as_int(acc[i].bmax) << 32) | (unsigned int)acc[i].imax);
                s[5 * kLanes] = __longlong_as_double(((long long)__float_as_int(acc[i].bmin) << 32) | (unsigned int)acc[i].imin);
            }
            __syncthreads();
            if (w == 0) {
#pragma unroll
                for (int ww = 1; ww < ZW; ++ww) {
                    const double* s = sh + (ww - 1) * 6 * kLanes + lane;
                    Acc b;
                    b.s0 = s[0 * kLanes]; b.s1 = s[1 * kLanes]; b.s2 = s[2 * kLanes];
                    b.n = (int)__double_as_longlong(s[3 * kLanes]);
                    long long q = __double_as_longlong(s[4 * kLanes]);
                    b.bmax = __int_as_float((int)(q >> 32)); b.imax = (int)(q & 0xffffffffLL);
                    q = __double_as_longlong(s[5 * kLanes]);
                    b.bmin = __int_as_float((int)(q >> 32)); b.imin = (int)(q & 0xffffffffLL);
                    acc_merge<EXT>(acc[i], b);
                }
            }
            __syncthreads();
        }
    }

    if (w == 0 && live) {
        const int64_t ncols = A.ngroups * VEC;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            if (A.nsplit == 1) finalize(A, acc[i], y, x + i);
            else ws_store(A, acc[i], split, g * VEC + i, ncols);
        }
    }
}

template <int EXT>
__global__ __launch_bounds__(256) void moments_combine_kernel(const MomArgs A, int vec) {
    const int64_t ncols = A.ngroups * vec;
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncols) return;
    Acc a;
    ws_load(A, a, 0, col, ncols);
    for (int s = 1; s < A.nsplit; ++s) {
        Acc b;
        ws_load(A, b, s, col, ncols);
        acc_merge<EXT>(a, b);
    }
    const int64_t g = col / vec;
    const int64_t y = g / A.groups_per_row;
    const int64_t x = (g - y * A.groups_per_row) * vec + (col - g * vec);
    finalize(A, a, y, x);
}

// ---- second pass for order N: sum v*(c-mu)^N / S0 -------------------------
struct OrdArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    const double* cen;
    const double* mu;
    const double* s0;
    double* out;
    int64_t out_row_stride;
    int order;
};

__global__ __launch_bounds__(256) void moment_order_kernel(const OrdArgs A) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ny * A.nx) return;
    const int64_t y = col / A.nx, x = col - y * A.nx;
    const float* p = A.cube + y * A.row_stride + x;
    const uint8_t* pm = (A.mask.flags & SPC_MASK_ARRAY) ? A.mask.arr + y * A.mask.row_stride + x : nullptr;
    const double mu = A.mu[y * A.out_row_stride + x];
    double s = 0.0;
    for (int64_t z = 0; z < A.nz; ++z) {
        const float v = p[z * A.plane_stride];
        bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, v);
        if (pm) inc = inc && pm[z * A.mask.plane_stride] != 0;
        if (inc && v == v) {
            const double d = A.cen[z] - mu;
            double pw = d;
            for (int k = 1; k < A.order; ++k) pw *= d;
            s = fma((double)v, pw, s);
        }
    }
    A.out[y * A.out_row_stride + x] = s / A.s0[y * A.out_row_stride + x];
}

// same sum with the access pattern of the main kernel: a lane owns 4 adjacent spaxels (16-byte loads),
// the 4 waves of a block split z (4 planes in flight), LDS combine
template <bool ARR>
__global__ __launch_bounds__(256) void moment_order_v4_kernel(const OrdArgs A) {
    constexpr int ZW = 4, U = 4;
    typedef float f32x4o __attribute__((ext_vector_type(4)));
    __shared__ double sh[ZW - 1][4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t gpr = A.nx / 4;
    const int64_t g = (int64_t)blockIdx.x * 64 + lane;
    const bool live = g < A.ny * gpr;
    const int64_t gg = live ? g : 0;
    const int64_t y = gg / gpr, x = (gg - y * gpr) * 4;
    const float* p = A.cube + y * A.row_stride + x;
    const uint8_t* pm = ARR ? A.mask.arr + y * A.mask.row_stride + x : nullptr;
    double mu[4], s[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { mu[c] = A.mu[y * A.out_row_stride + x + c]; s[c] = 0.0; }
    for (int64_t z0 = w; z0 < A.nz; z0 += ZW * U) {
        f32x4o v[U];
        uint32_t m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t z = min(z0 + (int64_t)u * ZW, A.nz - 1);
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4o*>(p + z * A.plane_stride));
            m[u] = ARR ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pm + z * A.mask.plane_stride)) : 0x01010101u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t z = z0 + (int64_t)u * ZW;
            if (z >= A.nz) break;                              // wave-uniform
            const double cz = A.cen[z];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float val = v[u][c];
                const bool ok = spc_pred_valid(A.mask, val) & (((m[u] >> (8 * c)) & 0xffu) != 0);
                const double d = cz - mu[c];
                double pw = d;
                for (int k = 1; k < A.order; ++k) pw *= d;
                s[c] = fma(ok ? (double)val : 0.0, ok ? pw : 0.0, s[c]);
            }
        }
    }
    if (w > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) sh[w - 1][c][lane] = s[c];
    }
    __syncthreads();
    if (w != 0 || !live) return;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int k = 0; k < ZW - 1; ++k) s[c] += sh[k][c][lane];
        A.out[y * A.out_row_stride + x + c] = s[c] / A.s0[y * A.out_row_stride + x + c];
    }
}

struct Plan { int vec, zw, u, nsplit; int64_t zchunk; };

template <int VEC, int ZW, int U, bool ARR, int EXT>
int launch_main(const MomArgs& A, hipStream_t st, bool thresholds) {
    dim3 block(kLanes, ZW);
    dim3 grid((unsigned)((A.ngroups + kLanes - 1) / kLanes), (unsigned)A.nsplit);
    if (thresholds) hipLaunchKernelGGL((moments_kernel<VEC, ZW, U, ARR, EXT, 1>), grid, block, 0, st, A);
    else hipLaunchKernelGGL((moments_kernel<VEC, ZW, U, ARR, EXT, 0>), grid, block, 0, st, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

// 16-byte lanes: ZW in {1, 4, 8}, U in {2, 4, 8}; the narrower lanes (odd nx, unaligned views): ZW in {1, 4}, U = 4
template <bool ARR, int EXT>
int launch_plan(const MomArgs& A, hipStream_t st, const Plan& p, bool thr) {
    if (p.vec == 4) {
        switch (p.zw * 16 + p.u) {
            case 1 * 16 + 2: return launch_main<4, 1, 2, ARR, EXT>(A, st, thr);
            case 1 * 16 + 4: return launch_main<4, 1, 4, ARR, EXT>(A, st, thr);
            case 1 * 16 + 8: return launch_main<4, 1, 8, ARR, EXT>(A, st, thr);
            case 4 * 16 + 2: return launch_main<4, 4, 2, ARR, EXT>(A, st, thr);
            case 4 * 16 + 4: return launch_main<4, 4, 4, ARR, EXT>(A, st, thr);
            case 4 * 16 + 8: return launch_main<4, 4, 8, ARR, EXT>(A, st, thr);
            case 8 * 16 + 2: return launch_main<4, 8, 2, ARR, EXT>(A, st, thr);
            case 8 * 16 + 4: return launch_main<4, 8, 4, ARR, EXT>(A, st, thr);
            default: return launch_main<4, 8, 8, ARR, EXT>(A, st, thr);
        }
    }
    if (p.vec == 2) return p.zw > 1 ? launch_main<2, 4, 4, ARR, EXT>(A, st, thr) : launch_main<2, 1, 4, ARR, EXT>(A, st, thr);
    return p.zw > 1 ? launch_main<1, 4, 4, ARR, EXT>(A, st, thr) : launch_main<1, 1, 4, ARR, EXT>(A, st, thr);
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

Plan make_plan(const spc_cube_f32* c, const MaskDev& m, bool ext) {
    Plan p;
    auto aligned = [&](int v) {
        const bool arr = (m.flags & SPC_MASK_ARRAY) != 0;
        return (c->nx % v == 0) && (c->row_stride % v == 0) && (c->plane_stride % v == 0) &&
               (((uintptr_t)c->d_data) % (4 * v) == 0) &&
               (!arr || ((m.row_stride % v == 0) && (m.plane_stride % v == 0) && (((uintptr_t)m.arr) % v == 0)));
    };
    // tuning hooks (SPC_MOMENTS_*): defaults chosen from MI355X measurements
    int vec = env_int("SPC_MOMENTS_VEC", 4);
    if (vec != 1 && vec != 2 && vec != 4) vec = 4;
    while (vec > 1 && !aligned(vec)) vec >>= 1;
    p.vec = vec;
    // MI355X sweeps at 1024^3 (tools/tune_moments.py; profiles/r01_tune_moments.log, r04_moments_issue.log)
    const bool arr_ = (m.flags & SPC_MASK_ARRAY) != 0;
    (void)arr_;
    p.u = env_int("SPC_MOMENTS_U", ext ? 2 : 8);
    if (p.u != 2 && p.u != 4 && p.u != 8) p.u = 4;
    const int64_t ngroups = c->ny * (c->nx / p.vec);
    const int64_t nblocks = (ngroups + kLanes - 1) / kLanes;
    // in-block z split: ZW waves share the columns when z is long enough
    // eight waves per block pay on planes up to 8 MiB (C2: 0.866 -> 0.827 ms), not on the north star's 16 MiB planes (13.06 -> 13.22 ms)
    p.zw = env_int("SPC_MOMENTS_ZW", c->nz >= 512 && !ext && c->ny * c->nx <= (1 << 21) ? 8 : (c->nz >= 16 ? 4 : 1));
    if (p.zw != 1 && p.zw != 4 && p.zw != 8) p.zw = 1;
    if (p.vec != 4) { p.u = 4; if (p.zw > 4) p.zw = 4; }
    // grid z split only when the map alone cannot fill the chip (256 CUs x ~8 blocks)
    int nsplit = 1;
    const int64_t target = 2048;
    if (nblocks < target && c->nz >= 256)
        nsplit = (int)std::min<int64_t>(std::min<int64_t>((target + nblocks - 1) / nblocks, c->nz / 64), 16);
    nsplit = std::max(1, env_int("SPC_MOMENTS_NSPLIT", nsplit));
    p.zchunk = (c->nz + nsplit - 1) / nsplit;
    p.nsplit = (int)((c->nz + p.zchunk - 1) / p.zchunk);
    return p;
}

}  // namespace

extern "C" {

size_t spc_moments_workspace_bytes(int64_t nz, int64_t ny, int64_t nx) {
    if (nz <= 0 || ny <= 0 || nx <= 0) return 0;
    // upper bound on nsplit used by make_plan
    int64_t nsplit_max = std::max<int64_t>(1, std::min<int64_t>(nz / 64, 16));
    const char* env = getenv("SPC_MOMENTS_NSPLIT");
    if (env) nsplit_max = std::max<int64_t>(nsplit_max, atoi(env));
    const int64_t nblocks = (ny * nx / 4 + kLanes - 1) / kLanes;
    if (!env && !getenv("SPC_MOMENTS_VEC") && (nblocks >= 2048 || nz < 256)) return 0;
    return (size_t)nsplit_max * kWsFields * sizeof(double) * (size_t)(ny * nx);
}

int spc_moments_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                    const double* d_cen, double dv, double m1_add, const spc_moment_outputs* out,
                    void* d_workspace, size_t workspace_bytes) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(out != nullptr, "outputs struct is NULL");
    SPC_REQUIRE(d_cen != nullptr, "d_cen is NULL");
    SPC_REQUIRE(cube->nz < (1LL << 31), "nz too large");
    MomArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.cen = d_cen; A.dv = dv; A.m1_add = m1_add;
    A.out = *out;
    A.out_row_stride = out->out_row_stride ? out->out_row_stride : cube->nx;
    const bool ext = out->d_argmax || out->d_argmin || out->d_vmax || out->d_vmin;
    Plan p = make_plan(cube, A.mask, ext);
    spc_canonical_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, &A.lim, &A.lo, &A.hi);
    const bool thr = (A.mask.flags & (SPC_MASK_GT | SPC_MASK_GE | SPC_MASK_LT | SPC_MASK_LE)) != 0;
    A.xcd_group = env_int("SPC_MOMENTS_XCD", 0);    // measured: +- 1 % either way on the real kernel (profiles/r04_moments_issue.log)
    A.groups_per_row = cube->nx / p.vec;
    A.ngroups = cube->ny * A.groups_per_row;
    A.nsplit = p.nsplit;
    A.zchunk = p.zchunk;
    A.ws = (double*)d_workspace;
    if (p.nsplit > 1) {
        const size_t need = (size_t)p.nsplit * kWsFields * sizeof(double) * (size_t)(cube->ny * cube->nx);
        if (!d_workspace || workspace_bytes < need) {
            // not enough workspace: fall back to a single z range (still correct)
            A.nsplit = 1; A.zchunk = cube->nz; p.nsplit = 1;
        }
    }
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    hipStream_t st = (hipStream_t)stream;
    const int what = ext ? kExtrema : (out->d_nvalid ? kCount : kSums);
    if (arr) rc = what == kExtrema ? launch_plan<true, kExtrema>(A, st, p, thr)
                : what == kCount ? launch_plan<true, kCount>(A, st, p, thr) : launch_plan<true, kSums>(A, st, p, thr);
    else rc = what == kExtrema ? launch_plan<false, kExtrema>(A, st, p, thr)
            : what == kCount ? launch_plan<false, kCount>(A, st, p, thr) : launch_plan<false, kSums>(A, st, p, thr);
    if (rc) return rc;
    if (A.nsplit > 1) {
        const int64_t ncols = cube->ny * cube->nx;
        dim3 grid((unsigned)((ncols + 255) / 256));
        if (ext) hipLaunchKernelGGL(moments_combine_kernel<kExtrema>, grid, dim3(256), 0, st, A, p.vec);
        else hipLaunchKernelGGL(moments_combine_kernel<kCount>, grid, dim3(256), 0, st, A, p.vec);
        SPC_LAUNCH_CHECK();
    }
    return SPC_OK;
}

int spc_moment_order_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                         const double* d_cen, int order, const double* d_mu, const double* d_s0,
                         double* d_out, int64_t out_row_stride) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(order >= 1 && order <= 64, "order must be in [1,64], got %d", order);
    SPC_REQUIRE(d_cen && d_mu && d_s0 && d_out, "NULL pointer argument");
    OrdArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.cen = d_cen; A.mu = d_mu; A.s0 = d_s0; A.out = d_out; A.order = order;
    A.out_row_stride = out_row_stride ? out_row_stride : cube->nx;
    const int64_t ncols = cube->ny * cube->nx;
    const bool arr_ = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const bool v4 = (cube->nx % 4 == 0) && (cube->row_stride % 4 == 0) && (cube->plane_stride % 4 == 0) &&
                    ((((uintptr_t)cube->d_data) & 15) == 0) &&
                    (!arr_ || ((A.mask.row_stride % 4 == 0) && (A.mask.plane_stride % 4 == 0) && ((((uintptr_t)A.mask.arr) & 3) == 0)));
    if (v4) {
        const int64_t ng = cube->ny * (cube->nx / 4);
        if (arr_) hipLaunchKernelGGL(moment_order_v4_kernel<true>, dim3((unsigned)((ng + 63) / 64)), dim3(256), 0, (hipStream_t)stream, A);
        else hipLaunchKernelGGL(moment_order_v4_kernel<false>, dim3((unsigned)((ng + 63) / 64)), dim3(256), 0, (hipStream_t)stream, A);
    } else
        hipLaunchKernelGGL(moment_order_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, A);
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

}  // extern "C"
