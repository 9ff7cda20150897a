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
};

__device__ __forceinline__ void emit(const SpMomArgs& A, int64_t o, double s0, double s1, double s2, int n) {
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    const double mu = s1 / s0;
    if (A.m0) A.m0[o] = n > 0 ? A.size * s0 : nan;
    if (A.m1) A.m1[o] = mu;
    if (A.m2) A.m2[o] = s2 / s0 - mu * mu;
}

__global__ __launch_bounds__(256) void moments_axis1_kernel(const SpMomArgs A) {
    const int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t z = blockIdx.y;
    if (x >= A.nx) return;
    const float* p = A.cube + z * A.plane_stride + x;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const uint8_t* pm = arr ? A.mask.arr + z * A.mask.plane_stride + x : nullptr;
    double s0 = 0, s1 = 0, s2 = 0;
    int n = 0;
    for (int64_t y = 0; y < A.ny; ++y) {
        const float v = p[y * A.row_stride];
        bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, v);
        if (arr) inc = inc && pm[y * A.mask.row_stride] != 0;
        if (inc && v == v) {
            const double c = A.cen[y * A.nx + x];
            s0 += (double)v;
            s1 = fma((double)v, c, s1);
            s2 = fma((double)v, c * c, s2);
            ++n;
        }
    }
    emit(A, z * A.nx + x, s0, s1, s2, n);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void moments_axis2_kernel(const SpMomArgs A) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (z,y) row index
    if (row >= A.nz * A.ny) return;
    const int64_t z = row / A.ny, y = row - z * A.ny;
    const float* p = A.cube + z * A.plane_stride + y * A.row_stride;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const uint8_t* pm = arr ? A.mask.arr + z * A.mask.plane_stride + y * A.mask.row_stride : nullptr;
    const double* cen = A.cen + y * A.nx;
    double s0 = 0, s1 = 0, s2 = 0;
    int n = 0;
    for (int64_t x = lane; x < A.nx; x += 64) {
        const float v = p[x];
        bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, v);
        if (arr) inc = inc && pm[x] != 0;
        if (inc && v == v) {
            const double c = cen[x];
            s0 += (double)v;
            s1 = fma((double)v, c, s1);
            s2 = fma((double)v, c * c, s2);
            ++n;
        }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
    if (lane == 0) emit(A, row, s0, s1, s2, n);
}

}  // namespace

extern "C" int spc_moments_spatial_f32(int device, void* stream, const spc_cube_f32* cube,
                                       const spc_mask* mask, int axis, const double* d_cen,
                                       double pix_size, double* d_m0, double* d_m1, double* d_m2) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    SPC_REQUIRE(axis == 1 || axis == 2, "axis must be 1 or 2 (got %d)", axis);
    SPC_REQUIRE(d_cen != nullptr, "d_cen is NULL");
    SpMomArgs A{};
    rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.cen = d_cen; A.size = pix_size; A.m0 = d_m0; A.m1 = d_m1; A.m2 = d_m2;
    hipStream_t st = (hipStream_t)stream;
    if (axis == 1) {
        SPC_REQUIRE(cube->nz <= 65535, "nz > 65535 not supported for axis-1 moments");
        hipLaunchKernelGGL(moments_axis1_kernel, dim3((unsigned)((cube->nx + 255) / 256), (unsigned)cube->nz),
                           dim3(256), 0, st, A);
    } else {
        const int64_t rows = cube->nz * cube->ny;
        hipLaunchKernelGGL(moments_axis2_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, A);
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}
