// spc_moments_f64.hip - moments along the spectral axis of a float64 cube (gfx950).
//
// The reference keeps a float64 source in float64 (np.result_type(dtype, 0.0), spectral_cube/masks.py:225; a BITPIX = -64 /
// 32 / 64 FITS image arrives as float64 through astropy) and its moment maps carry that precision
// (spectral_cube/_moments.py:30-193, dask_spectral_cube.py:1083-1104).  The float32 kernels round every SAMPLE to 24 bits
// on its way to HBM; this path keeps the samples as they are: 8 bytes per voxel read once (16-byte loads, two adjacent
// spaxels per lane), the sums in float64 like moments_kernel, the mask thresholds compared in float64.
//
//   pass 1 (moments_f64_kernel):  S0 = sum v, S1 = sum v c[z], count, extrema  ->  m0 = dv S0, mu = S1 / S0, m1 = mu + m1_add
//   pass 2 (moment_order_f64_kernel, order N >= 2):  sum v (c[z] - mu)^N / S0 - the reference's own two-pass form
//   (_moments.py:185-193): the one-pass S2 / S0 - mu^2 of the float32 kernel loses (band / line width)^2 ulps, which the
//   1e-5 contract of the float32 configurations absorbs and a float64 result does not.
//
// HBM-bound (8 B per voxel per pass); eight waves of a block split the channels, their partial sums meet in LDS.
#include "spc_common.h"
#include "spc_wide.h"

#include <algorithm>

namespace {

constexpr int kLanes = 64;
constexpr int kZW = 8;                  // waves per block, each takes every 8th channel
constexpr int kU = 8;                   // loads in flight per lane (round 4: 4 -> 8, like the float32 kernel)

struct Mom64Args {
    const double* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev64 mask;
    const double* cen;
    double dv, m1_add;
    spc_moment_outputs_f64 out;
    int64_t out_row_stride;
    int64_t groups_per_row, ngroups;
    // second pass
    const double* mu;
    const double* s0;
    double* ord_out;
    int order;
};

struct Acc64 {
    double s0, s1;
    int n;
    double bmax, bmin;
    int imax, imin;
};

typedef double f64x2 __attribute__((ext_vector_type(2)));
template <int VEC> struct Vec64;
template <> struct Vec64<2> { using F = f64x2; using M = uint16_t; };
template <> struct Vec64<1> { using F = double; using M = unsigned char; };
__device__ __forceinline__ double vget(const f64x2& v, int i) { return v[i]; }
__device__ __forceinline__ double vget(const double& v, int) { return v; }
__device__ __forceinline__ unsigned mget(const uint16_t& v, int i) { return (v >> (8 * i)) & 0xffu; }
__device__ __forceinline__ unsigned mget(const unsigned char& v, int) { return v; }

template <bool EXT>
__device__ __forceinline__ void merge64(Acc64& a, const Acc64& b) {
    a.s0 += b.s0; a.s1 += b.s1; a.n += b.n;
    if (EXT) {                                                   // ties keep the smaller channel (first-index rule)
        if (b.bmax > a.bmax || (b.bmax == a.bmax && b.imax < a.imax)) { a.bmax = b.bmax; a.imax = b.imax; }
        if (b.bmin < a.bmin || (b.bmin == a.bmin && b.imin < a.imin)) { a.bmin = b.bmin; a.imin = b.imin; }
    }
}

// ORDER == 0: first pass (S0, S1, count, extrema).  ORDER == 1: second pass, s0 accumulates v (c - mu)^order.
template <int VEC, bool ARR, bool EXT, int ORDER>
__global__ __launch_bounds__(kLanes * kZW) void moments_f64_kernel(const Mom64Args A) {
    using F = typename Vec64<VEC>::F;
    using M = typename Vec64<VEC>::M;
    const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y);   // blockDim.x = 64: a plane's coordinate is a scalar load
    const int64_t g = (int64_t)blockIdx.x * kLanes + lane;
    const bool live = g < A.ngroups;
    const int64_t gg = live ? g : 0;
    const int64_t y = gg / A.groups_per_row;
    const int64_t x = (gg - y * A.groups_per_row) * VEC;
    const double* p = A.cube + y * A.row_stride + x;
    const uint8_t* pm = ARR ? A.mask.arr + y * A.mask.row_stride + x : nullptr;

    Acc64 acc[VEC];
    double mu[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        acc[i].s0 = acc[i].s1 = 0.0; acc[i].n = 0;
        acc[i].bmax = -INFINITY; acc[i].bmin = INFINITY;
        acc[i].imax = acc[i].imin = (int)min((int64_t)w, A.nz - 1);
        mu[i] = (ORDER && live) ? A.mu[y * A.out_row_stride + x + i] : 0.0;
    }
    auto take = [&](const F& v, const M& m, int64_t z) {
        const double c = ((const double __attribute__((address_space(4)))*)A.cen)[z];   // uniform index: a scalar load
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            const double val = vget(v, i);
            bool ok = pred64(A.mask, val);
            if (ARR) ok = ok & (mget(m, i) != 0);
            const double wd = ok ? val : 0.0;
            if (ORDER) {
                const double d = c - mu[i];
                double pw = d;
                for (int k = 1; k < A.order; ++k) pw *= d;
                // (an excluded sample adds nothing even where pw overflowed or mu is NaN)
                acc[i].s0 = ok ? fma(val, pw, acc[i].s0) : acc[i].s0;
            } else {
                acc[i].s0 += wd;
                acc[i].s1 = fma(wd, c, acc[i].s1);
                acc[i].n += ok ? 1 : 0;
                if (EXT) {
                    const double hi = ok ? val : -INFINITY, lo = ok ? val : INFINITY;
                    if (hi > acc[i].bmax) { acc[i].bmax = hi; acc[i].imax = (int)z; }
                    if (lo < acc[i].bmin) { acc[i].bmin = lo; acc[i].imin = (int)z; }
                }
            }
        }
    };
    if (live) {
        int64_t z = w;
        for (; z + (int64_t)(kU - 1) * kZW < A.nz; z += (int64_t)kU * kZW) {
            F v[kU];
            M m[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int64_t zz = z + (int64_t)u * kZW;
                v[u] = __builtin_nontemporal_load(reinterpret_cast<const F*>(p + zz * A.plane_stride));
                m[u] = ARR ? __builtin_nontemporal_load(reinterpret_cast<const M*>(pm + zz * A.mask.plane_stride)) : (M)0;
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) take(v[u], m[u], z + (int64_t)u * kZW);
        }
        for (; z < A.nz; z += kZW) {
            const F v = *reinterpret_cast<const F*>(p + z * A.plane_stride);
            const M m = ARR ? *reinterpret_cast<const M*>(pm + z * A.mask.plane_stride) : (M)0;
            take(v, m, z);
        }
    }
    // the block's eight waves: partial sums through LDS, wave 0 finishes (one VEC element at a time)
    __shared__ double sh[(kZW - 1) * 5 * kLanes];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        if (w > 0) {
            double* s = sh + (w - 1) * 5 * kLanes + lane;
            s[0 * kLanes] = acc[i].s0;
            s[1 * kLanes] = acc[i].s1;
            s[2 * kLanes] = acc[i].bmax;
            s[3 * kLanes] = acc[i].bmin;
            s[4 * kLanes] = __longlong_as_double(((long long)acc[i].n << 42) | ((long long)acc[i].imax << 21) | (long long)acc[i].imin);
        }
        __syncthreads();
        if (w == 0) {
            for (int ww = 1; ww < kZW; ++ww) {
                const double* s = sh + (ww - 1) * 5 * kLanes + lane;
                Acc64 b;
                b.s0 = s[0 * kLanes]; b.s1 = s[1 * kLanes]; b.bmax = s[2 * kLanes]; b.bmin = s[3 * kLanes];
                const long long q = __double_as_longlong(s[4 * kLanes]);
                b.n = (int)(q >> 42); b.imax = (int)((q >> 21) & 0x1fffff); b.imin = (int)(q & 0x1fffff);
                merge64<EXT>(acc[i], b);
            }
        }
        __syncthreads();
    }
    if (w != 0 || !live) return;
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        const int64_t o = y * A.out_row_stride + x + i;
        const Acc64& a = acc[i];
        if (ORDER) {
            A.ord_out[o] = a.s0 / A.s0[o];
            continue;
        }
        const double m = a.s1 / a.s0;                            // 0 / 0 -> NaN for empty rays, like the reference
        if (A.out.d_m0) A.out.d_m0[o] = a.n > 0 ? A.dv * a.s0 : nan;
        if (A.out.d_m1) A.out.d_m1[o] = m + A.m1_add;
        if (A.out.d_mu) A.out.d_mu[o] = m;
        if (A.out.d_s0) A.out.d_s0[o] = a.s0;
        if (A.out.d_argmax) A.out.d_argmax[o] = a.n > 0 ? (int64_t)a.imax : 0;
        if (A.out.d_argmin) A.out.d_argmin[o] = a.n > 0 ? (int64_t)a.imin : 0;
        if (A.out.d_vmax) A.out.d_vmax[o] = a.n > 0 ? a.bmax : nan;
        if (A.out.d_vmin) A.out.d_vmin[o] = a.n > 0 ? a.bmin : nan;
        if (A.out.d_nvalid) A.out.d_nvalid[o] = a.n;
    }
}

bool pairs_ok(const spc_cube_f64* c, const MaskDev64& m) {
    const bool arr = (m.flags & SPC_MASK_ARRAY) != 0;
    return (c->nx % 2 == 0) && (c->row_stride % 2 == 0) && (c->plane_stride % 2 == 0) && ((((uintptr_t)c->d_data) & 15) == 0) &&
           (!arr || ((m.row_stride % 2 == 0) && (m.plane_stride % 2 == 0) && ((((uintptr_t)m.arr) & 1) == 0)));
}

template <int ORDER>
int launch64(Mom64Args& A, const spc_cube_f64* cube, bool ext, hipStream_t st) {
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const int vec = pairs_ok(cube, A.mask) ? 2 : 1;
    A.groups_per_row = cube->nx / vec;
    A.ngroups = cube->ny * A.groups_per_row;
    const int64_t nblk = (A.ngroups + kLanes - 1) / kLanes;
    SPC_REQUIRE(nblk < (1LL << 31), "map too large");
    dim3 grid((unsigned)nblk), block(kLanes, kZW);
#define SPC_L64(V_, A_, E_) hipLaunchKernelGGL((moments_f64_kernel<V_, A_, E_, ORDER>), grid, block, 0, st, A)
    if (vec == 2) {
        if (arr) { if (ext) SPC_L64(2, true, true); else SPC_L64(2, true, false); }
        else { if (ext) SPC_L64(2, false, true); else SPC_L64(2, false, false); }
    } else {
        if (arr) { if (ext) SPC_L64(1, true, true); else SPC_L64(1, true, false); }
        else { if (ext) SPC_L64(1, false, true); else SPC_L64(1, false, false); }
    }
#undef SPC_L64
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

}  // namespace

extern "C" {

int spc_moments_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, const double* d_cen,
                    double dv, double m1_add, const spc_moment_outputs_f64* out) {
    int rc = check_cube64(cube);
    if (rc) return rc;
    SPC_REQUIRE(out != nullptr, "outputs struct is NULL");
    SPC_REQUIRE(d_cen != nullptr, "d_cen is NULL");
    Mom64Args A{};
    rc = mask64_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.cen = d_cen; A.dv = dv; A.m1_add = m1_add;
    A.out = *out;
    A.out_row_stride = out->out_row_stride ? out->out_row_stride : cube->nx;
    const bool ext = out->d_argmax || out->d_argmin || out->d_vmax || out->d_vmin;
    return launch64<0>(A, cube, ext, (hipStream_t)stream);
}

int spc_moment_order_f64(int device, void* stream, const spc_cube_f64* cube, const spc_mask_f64* mask, const double* d_cen,
                         int order, const double* d_mu, const double* d_s0, double* d_out, int64_t out_row_stride) {
    int rc = check_cube64(cube);
    if (rc) return rc;
    SPC_REQUIRE(order >= 1 && order <= 64, "order must be in [1,64], got %d", order);
    SPC_REQUIRE(d_cen && d_mu && d_s0 && d_out, "NULL pointer argument");
    Mom64Args A{};
    rc = mask64_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.cen = d_cen; A.mu = d_mu; A.s0 = d_s0; A.ord_out = d_out; A.order = order;
    A.out_row_stride = out_row_stride ? out_row_stride : cube->nx;
    return launch64<1>(A, cube, false, (hipStream_t)stream);
}

}  // extern "C"
