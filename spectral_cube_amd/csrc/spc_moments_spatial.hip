// Masked moment 0/1/2 along a SPATIAL axis (axis 1 = y, axis 2 = x).
//
// Same arithmetic as spc_moments.hip (spectral_cube/dask_spectral_cube.py
// :1083-1104 with axis != 0; golden tables spectral_cube/tests/test_moments.py
// :19-49) but the pixel-centre offsets are a (ny,nx) map
// (spectral_cube/spectral_cube.py:1476-1503) and no world coordinate is added.
//   axis 1: output (nz,nx); lanes along x, each lane marches down y;
//   axis 2: output (nz,ny); one wavefront per (z,y) row, lanes stride over x
//           and the 64 partial sums are folded with wave shuffles.
#include "spc_common.h"

namespace {

struct SpMomArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    const double* cen;     // (ny,nx)
    double size;
    double *m0, *m1, *m2;
    // second pass for order N >= 3 (ORD kernels): out = sum v (c - mu)^N / sum v with mu = the
    // first-pass moment 1 of the same ray (dask_spectral_cube.py:1094-1099; _moments.py:185-193)
    int order;
    const double* mu;
    double* mN;
};

__device__ __forceinline__ double ipow(double b, int n) {      // b^n by squaring, n >= 0
    double r = 1.0;
    while (n) { if (n & 1) r *= b; b *= b; n >>= 1; }
    return r;
}

__device__ __forceinline__ void emit(const SpMomArgs& A, int64_t o, double s0, double s1, double s2, int n) {
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    if (A.order) { A.mN[o] = s1 / s0; return; }
    const double mu = s1 / s0;
    if (A.m0) A.m0[o] = n > 0 ? A.size * s0 : nan;
    if (A.m1) A.m1[o] = mu;
    if (A.m2) A.m2[o] = s2 / s0 - mu * mu;
}

typedef float f32x4m __attribute__((ext_vector_type(4)));
typedef double f64x2m __attribute__((ext_vector_type(2)));

struct Acc3 { double s0, s1, s2; int n; };

template <bool ORD>
__device__ __forceinline__ void acc3_add(Acc3& a, float v, bool ok, double c, int order, double mu) {
    const double d = ok ? (double)v : 0.0;
    a.s0 += d;
    if (ORD) {
        a.s1 = fma(d, ipow(c - mu, order), a.s1);
    } else {
        a.s1 = fma(d, c, a.s1);
        a.s2 = fma(d, c * c, a.s2);
    }
    a.n += ok ? 1 : 0;
}

// axis 1: a lane owns VEC adjacent x of one channel, the 4 waves of a block split y (U rows in flight),
// LDS combine - the access pattern of the spectral moment kernel with y in the role of z
template <int VEC, bool ARR, bool ORD>
__global__ __launch_bounds__(256) void moments_axis1_kernel(const SpMomArgs A) {
    constexpr int YW = 4, U = 4;
    __shared__ double sh[YW - 1][4][VEC][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t x = ((int64_t)blockIdx.x * 64 + lane) * VEC;
    const int64_t z = blockIdx.y;
    const bool live = x < A.nx;
    const int64_t xc = live ? x : 0;
    const float* p = A.cube + z * A.plane_stride + xc;
    const uint8_t* pm = ARR ? A.mask.arr + z * A.mask.plane_stride + xc : nullptr;
    const double* pc = A.cen + xc;
    Acc3 a[VEC];
    double muv[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        a[c] = Acc3{0.0, 0.0, 0.0, 0};
        muv[c] = (ORD && x + c < A.nx) ? A.mu[z * A.nx + x + c] : 0.0;
    }
    for (int64_t y0 = w; y0 < A.ny; y0 += YW * U) {
        float v[U][VEC];
        unsigned mk[U][VEC];
        double cc[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t y = min(y0 + (int64_t)u * YW, A.ny - 1);
            if (VEC == 4) {
                const f32x4m q = __builtin_nontemporal_load(reinterpret_cast<const f32x4m*>(p + y * A.row_stride));
                const uint32_t m = ARR ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(pm + y * A.mask.row_stride)) : 0x01010101u;
                const f64x2m c01 = *reinterpret_cast<const f64x2m*>(pc + y * A.nx);
                const f64x2m c23 = *reinterpret_cast<const f64x2m*>(pc + y * A.nx + 2);
#pragma unroll
                for (int c = 0; c < 4; ++c) { v[u][c] = q[c]; mk[u][c] = (m >> (8 * c)) & 0xffu; }
                cc[u][0] = c01.x; cc[u][1] = c01.y; cc[u][2] = c23.x; cc[u][3] = c23.y;
            } else {
                v[u][0] = p[y * A.row_stride];
                mk[u][0] = ARR ? pm[y * A.mask.row_stride] : 1u;
                cc[u][0] = pc[y * A.nx];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = y0 + (int64_t)u * YW < A.ny;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const float val = v[u][c];
                const bool ok = in & spc_pred_valid(A.mask, val) & (mk[u][c] != 0);
                acc3_add<ORD>(a[c], val, ok, cc[u][c], A.order, muv[c]);
            }
        }
    }
    if (w > 0) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            sh[w - 1][0][c][lane] = a[c].s0; sh[w - 1][1][c][lane] = a[c].s1;
            sh[w - 1][2][c][lane] = a[c].s2; sh[w - 1][3][c][lane] = (double)a[c].n;
        }
    }
    __syncthreads();
    if (w != 0 || !live) return;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
#pragma unroll
        for (int k = 0; k < YW - 1; ++k) {
            a[c].s0 += sh[k][0][c][lane]; a[c].s1 += sh[k][1][c][lane];
            a[c].s2 += sh[k][2][c][lane]; a[c].n += (int)sh[k][3][c][lane];
        }
        if (x + c < A.nx) emit(A, z * A.nx + x + c, a[c].s0, a[c].s1, a[c].s2, a[c].n);
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// axis 2: one wavefront per (z, y) row, 16-byte loads when the row is aligned, wave reduction
template <bool ARR, bool ORD>
__global__ __launch_bounds__(256) void moments_axis2_kernel(const SpMomArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (z,y) row index
    if (row >= A.nz * A.ny) return;
    const int64_t z = row / A.ny, y = row - z * A.ny;
    const float* p = A.cube + z * A.plane_stride + y * A.row_stride;
    const uint8_t* pm = ARR ? A.mask.arr + z * A.mask.plane_stride + y * A.mask.row_stride : nullptr;
    const double* cen = A.cen + y * A.nx;
    Acc3 a{0.0, 0.0, 0.0, 0};
    const double muv = ORD ? A.mu[row] : 0.0;
    const bool al = ((((uintptr_t)p) & 15) == 0) && ((((uintptr_t)cen) & 15) == 0) && (!ARR || ((((uintptr_t)pm) & 3) == 0));
    const int64_t n4 = al ? A.nx / 4 : 0;
    for (int64_t i = lane; i < n4; i += 64) {
        const f32x4m q = __builtin_nontemporal_load(reinterpret_cast<const f32x4m*>(p) + i);
        const uint32_t m = ARR ? reinterpret_cast<const uint32_t*>(pm)[i] : 0x01010101u;
        const f64x2m c01 = reinterpret_cast<const f64x2m*>(cen)[2 * i], c23 = reinterpret_cast<const f64x2m*>(cen)[2 * i + 1];
        const double cc[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float val = q[c];
            const bool ok = spc_pred_valid(A.mask, val) & (((m >> (8 * c)) & 0xffu) != 0);
            acc3_add<ORD>(a, val, ok, cc[c], A.order, muv);
        }
    }
    for (int64_t x = n4 * 4 + lane; x < A.nx; x += 64) {
        const float val = p[x];
        const bool ok = spc_pred_valid(A.mask, val) && (!ARR || pm[x] != 0);
        acc3_add<ORD>(a, val, ok, cen[x], A.order, muv);
    }
    a.s0 = wave_sum(a.s0); a.s1 = wave_sum(a.s1); a.s2 = wave_sum(a.s2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a.n += __shfl_down(a.n, off, 64);
    if (lane == 0) emit(A, row, a.s0, a.s1, a.s2, a.n);
}

}  // namespace

template <bool ORD>
static int launch_spatial(const SpMomArgs& A, const spc_cube_f32* cube, int axis, const double* d_cen, hipStream_t st) {
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    if (axis == 1) {
        SPC_REQUIRE(cube->nz <= 65535, "nz > 65535 not supported for axis-1 moments");
        const bool v4 = (cube->nx % 4 == 0) && (cube->row_stride % 4 == 0) && (cube->plane_stride % 4 == 0) &&
                        ((((uintptr_t)cube->d_data) & 15) == 0) && ((((uintptr_t)d_cen) & 15) == 0) &&
                        (!arr || ((A.mask.row_stride % 4 == 0) && (A.mask.plane_stride % 4 == 0) && ((((uintptr_t)A.mask.arr) & 3) == 0)));
        if (v4) {
            dim3 grid((unsigned)((cube->nx + 255) / 256), (unsigned)cube->nz);
            if (arr) hipLaunchKernelGGL((moments_axis1_kernel<4, true, ORD>), grid, dim3(256), 0, st, A);
            else hipLaunchKernelGGL((moments_axis1_kernel<4, false, ORD>), grid, dim3(256), 0, st, A);
        } else {
            dim3 grid((unsigned)((cube->nx + 63) / 64), (unsigned)cube->nz);
            if (arr) hipLaunchKernelGGL((moments_axis1_kernel<1, true, ORD>), grid, dim3(256), 0, st, A);
            else hipLaunchKernelGGL((moments_axis1_kernel<1, false, ORD>), grid, dim3(256), 0, st, A);
        }
    } else {
        const int64_t rows = cube->nz * cube->ny;
        if (arr) hipLaunchKernelGGL((moments_axis2_kernel<true, ORD>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, A);
        else hipLaunchKernelGGL((moments_axis2_kernel<false, ORD>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, A);
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

static int fill_spatial(SpMomArgs& A, const spc_cube_f32* cube, const spc_mask* mask, int axis, const double* d_cen) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(axis == 1 || axis == 2, "axis must be 1 or 2 (got %d)", axis);
    SPC_REQUIRE(d_cen != nullptr, "d_cen is NULL");
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.cen = d_cen;
    return SPC_OK;
}

extern "C" int spc_moments_spatial_f32(int device, void* stream, const spc_cube_f32* cube,
                                       const spc_mask* mask, int axis, const double* d_cen,
                                       double pix_size, double* d_m0, double* d_m1, double* d_m2) {
    SpMomArgs A{};
    int rc = fill_spatial(A, cube, mask, axis, d_cen);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.size = pix_size; A.m0 = d_m0; A.m1 = d_m1; A.m2 = d_m2;
    return launch_spatial<false>(A, cube, axis, d_cen, (hipStream_t)stream);
}

extern "C" int spc_moment_order_spatial_f32(int device, void* stream, const spc_cube_f32* cube,
                                            const spc_mask* mask, int axis, const double* d_cen, int order,
                                            const double* d_mu, double* d_out) {
    SpMomArgs A{};
    int rc = fill_spatial(A, cube, mask, axis, d_cen);
    if (rc) return rc;
    SPC_REQUIRE(order >= 2 && order <= 64, "order must be in 2..64 (got %d)", order);
    SPC_REQUIRE(d_mu != nullptr && d_out != nullptr, "NULL pointer argument");
    SPC_DEVICE(device);
    A.order = order; A.mu = d_mu; A.mN = d_out;
    return launch_spatial<true>(A, cube, axis, d_cen, (hipStream_t)stream);
}
