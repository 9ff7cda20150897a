// NaN-aware per-channel 2-D convolution (astropy.convolution.convolve
// semantics: boundary='fill' fill_value=0, nan_treatment='interpolate',
// normalize_kernel=True) for spatial_smooth.
//
// Replaces the chunk function of DaskSpectralCubeMixin.spatial_smooth
// (spectral_cube/dask_spectral_cube.py:962-993 + wrapper :540-547; NumPy twin
// spectral_cube/spectral_cube.py:2808-2842).
//
// Separable path (Gaussian2DKernel == outer(g, g)): the NaN renormalisation
// separates too, out = (Gy*(Gx*(d.ok))) / (Gy*(Gx*ok)).
//   y pass : one lane per INPUT column streams down the rows with the same
//            register ring as the spectral kernel (packed (num,den) fp32
//            accumulators, every input row loaded once, coalesced along x);
//   x pass : every RY finished rows ("one ring revolution") sit in LDS as
//            (num_y, den_y) pairs; each lane then produces a run of 8
//            consecutive x outputs from 8+2HX LDS reads (bank-conflict-free
//            9-float2 pitch), divides and stores 32 contiguous bytes.
// One block = one channel x one strip of TXO output columns (+2HX halo
// columns, 12 % overhead for 29 taps); the plane is read once from HBM.
#include "spc_common.h"
#include <algorithm>
#include <alloca.h>

namespace {

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
    float ky[kMaxTaps];                       // padded to RY, centred
    float kx[kMaxTaps];                       // padded to RX, centred
};

__device__ __forceinline__ int lds_phys(int t) { return t + (t >> 3); }   // 9-per-8 padding

template <int RY, int RX, bool ARR, bool SYM>
__global__ __launch_bounds__(kThreads) void spatial_sep_kernel(const SpArgs A) {
    constexpr int HY = RY / 2, HX = RX / 2;
    constexpr int kPitch = kThreads + kThreads / 8 + 2;   // float2 per LDS row
    __shared__ float2v yres[RY * kPitch];

    const int t = threadIdx.x;
    const int64_t z = blockIdx.y;
    const int64_t x0 = (int64_t)blockIdx.x * A.txo;       // first output column of the strip
    const int64_t yb = (int64_t)blockIdx.z * A.ychunk;
    const int64_t ye = min(A.ny, yb + A.ychunk);
    const int64_t xin = x0 - HX + t;                      // input column of this lane (y pass)
    const bool col_in = (xin >= 0) && (xin < A.nx) && (t < A.txo + 2 * HX);
    const int64_t xc = min(max(xin, (int64_t)0), A.nx - 1);
    const float* p = A.cube + z * A.plane_stride + xc;
    const uint8_t* pm = ARR ? A.mask.arr + z * A.mask.plane_stride + xc : nullptr;
    const uint32_t flags = A.mask.flags;
    const float tlo = A.mask.thr_lo, thi = A.mask.thr_hi;
    const int nrun = A.txo / kRun;

    float2v acc[RY];
#pragma unroll
    for (int m = 0; m < RY; ++m) acc[m] = float2v{0.f, 0.f};

    const int64_t T = (ye - yb) + 2 * HY;
    for (int64_t t0 = 0; t0 < T; t0 += RY) {
        const int64_t i0 = yb - HY + t0;                  // first input row of this revolution
        float v[RY];
        unsigned char mk[RY];
#pragma unroll
        for (int s = 0; s < RY; ++s) {
            const int64_t ic = min(max(i0 + s, (int64_t)0), A.ny - 1);
            v[s] = p[ic * A.row_stride];
            if (ARR) mk[s] = pm[ic * A.mask.row_stride];
        }
#pragma unroll
        for (int s = 0; s < RY; ++s) {
            const int64_t i = i0 + s;
            const bool inr = col_in && (i >= 0) && (i < A.ny);
            bool inc = spc_pred(flags, tlo, thi, v[s]);
            if (ARR) inc = inc && (mk[s] != 0);
            const bool ok = inr ? (inc && (v[s] == v[s])) : true;   // out of bounds = valid zero
            float2v x2;
            x2.x = (ok && inr) ? v[s] : 0.f;
            x2.y = ok ? 1.f : 0.f;
#pragma unroll
            for (int m = 0; m < RY; ++m) {
                const int a = (s - m + RY) % RY;
                const float wgt = A.ky[SYM ? (a <= HY ? a : 2 * HY - a) : 2 * HY - a];
                const float2v w2 = float2v{wgt, wgt};
                if (a == 0) acc[m] = w2 * x2;
                else acc[m] = __builtin_elementwise_fma(w2, x2, acc[m]);
                asm volatile("" : "+v"(acc[m]));            // see spc_spectral_conv.hip
            }
            // row o = i - HY is complete: park it in LDS slot s
            yres[s * kPitch + lds_phys(t)] = acc[(s + 1) % RY];
        }
        __syncthreads();
        // ---- x pass over the RY rows of this revolution
        for (int task = t; task < RY * nrun; task += kThreads) {
            const int s = task / nrun;
            const int j = task - s * nrun;
            const int64_t o = i0 + s - HY;                // output row
            if (o < yb || o >= ye) continue;
            float2v r[kRun];
#pragma unroll
            for (int k = 0; k < kRun; ++k) r[k] = float2v{0.f, 0.f};
            const float2v* src = yres + s * kPitch;
#pragma unroll
            for (int i = 0; i < kRun + 2 * HX; ++i) {
                const float2v in = src[lds_phys(kRun * j + i)];
#pragma unroll
                for (int k = 0; k < kRun; ++k) {
                    const int widx = k + 2 * HX - i;      // kx index for output k, input i
                    if (widx >= 0 && widx <= 2 * HX) {
                        const float wgt = A.kx[SYM ? (widx <= HX ? widx : 2 * HX - widx) : widx];
                        r[k] = __builtin_elementwise_fma(float2v{wgt, wgt}, in, r[k]);
                    }
                }
            }
            const int64_t xo = x0 + kRun * j;
            float* dst = A.out + z * A.out_plane_stride + o * A.out_row_stride + xo;
            float res[kRun];
#pragma unroll
            for (int k = 0; k < kRun; ++k) {
                if (r[k].y != 0.f) {
                    res[k] = r[k].x * __builtin_amdgcn_rcpf(r[k].y);
                } else {
                    // empty window -> (filled) centre sample, like astropy
                    res[k] = NAN;
                    if (xo + k < A.nx) {
                        const float c = A.cube[z * A.plane_stride + o * A.row_stride + xo + k];
                        bool inc = spc_pred(flags, tlo, thi, c);
                        if (ARR) inc = inc && A.mask.arr[z * A.mask.plane_stride + o * A.mask.row_stride + xo + k] != 0;
                        if (inc) res[k] = c;
                    }
                }
            }
            if (xo + kRun <= A.nx && ((((uintptr_t)dst) & 15) == 0)) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                *reinterpret_cast<f32x4*>(dst) = f32x4{res[0], res[1], res[2], res[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{res[4], res[5], res[6], res[7]};
            } else {
#pragma unroll
                for (int k = 0; k < kRun; ++k)
                    if (xo + k < A.nx) dst[k] = res[k];
            }
        }
        __syncthreads();
    }
}

// generic direct 2-D convolution (non-separable kernels such as Tophat2DKernel
// or rotated elliptical Gaussians): one thread per output pixel, taps in
// device memory, re-reads served by L1/L2.
__global__ __launch_bounds__(256) void spatial_conv2d_kernel(const SpArgs A, const float* kern, int nky, int nkx) {
    const int64_t x = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t y = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    const int64_t z = blockIdx.z;
    if (x >= A.nx || y >= A.ny) return;
    const int hy = nky / 2, hx = nkx / 2;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const float* p = A.cube + z * A.plane_stride;
    const uint8_t* pm = arr ? A.mask.arr + z * A.mask.plane_stride : nullptr;
    float num = 0.f, den = 0.f;
    for (int jy = 0; jy < nky; ++jy) {
        const int64_t iy = y + hy - jy;
        for (int jx = 0; jx < nkx; ++jx) {
            const float w = kern[jy * nkx + jx];
            if (w == 0.f) continue;
            const int64_t ix = x + hx - jx;
            float v = 0.f; bool ok = true;
            if (iy >= 0 && iy < A.ny && ix >= 0 && ix < A.nx) {
                v = p[iy * A.row_stride + ix];
                bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, v);
                if (arr) inc = inc && pm[iy * A.mask.row_stride + ix] != 0;
                ok = inc && (v == v);
                if (!ok) v = 0.f;
            }
            num = fmaf(w, v, num);
            den = fmaf(w, ok ? 1.f : 0.f, den);
        }
    }
    float res;
    if (den != 0.f) res = num / den;
    else {
        const float c = p[y * A.row_stride + x];
        bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, c);
        if (arr) inc = inc && pm[y * A.mask.row_stride + x] != 0;
        res = inc ? c : NAN;
    }
    A.out[z * A.out_plane_stride + y * A.out_row_stride + x] = res;
}

int pick_ring(int ntaps) {
    const int rings[] = {9, 17, 29, 33, 65};
    for (int r : rings) if (ntaps <= r) return r;
    return 0;
}

void pad_taps(float* dst, const double* k, int ntaps, int R) {
    for (int i = 0; i < kMaxTaps; ++i) dst[i] = 0.f;
    const int pad = (R - ntaps) / 2;
    for (int i = 0; i < ntaps; ++i) dst[pad + i] = (float)k[i];
}

bool is_sym(const float* k, int R) {
    for (int i = 0; i < R / 2; ++i) if (k[i] != k[R - 1 - i]) return false;
    return true;
}

template <int R>
int launch_sep(const SpArgs& A, hipStream_t st, dim3 grid, bool arr) {
    const bool sym = is_sym(A.ky, R) && is_sym(A.kx, R);
    dim3 block(kThreads);
    if (arr) {
        if (sym) hipLaunchKernelGGL((spatial_sep_kernel<R, R, true, true>), grid, block, 0, st, A);
        else hipLaunchKernelGGL((spatial_sep_kernel<R, R, true, false>), grid, block, 0, st, A);
    } else {
        if (sym) hipLaunchKernelGGL((spatial_sep_kernel<R, R, false, true>), grid, block, 0, st, A);
        else hipLaunchKernelGGL((spatial_sep_kernel<R, R, false, false>), grid, block, 0, st, A);
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int check_kernel(const double* k, int n, const char* what) {
    SPC_REQUIRE(k != nullptr, "%s kernel pointer is NULL", what);
    SPC_REQUIRE(n >= 1 && (n % 2) == 1, "%s kernel must have an odd number of taps (got %d)", what, n);
    return SPC_OK;
}

int fill_args(SpArgs& A, const spc_cube_f32* cube, const spc_mask* mask, float* d_out,
              int64_t out_row_stride, int64_t out_plane_stride) {
    int rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    A.out = d_out;
    A.out_row_stride = out_row_stride ? out_row_stride : cube->nx;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : cube->ny * A.out_row_stride;
    return SPC_OK;
}

}  // namespace

extern "C" {

int spc_spatial_conv2d_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                           const double* h_kernel, int nky, int nkx, float* d_out,
                           int64_t out_row_stride, int64_t out_plane_stride) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    if ((rc = check_kernel(h_kernel, nky, "y")) || (rc = check_kernel(h_kernel, nkx, "x"))) return rc;
    double sum = 0.0;
    for (int i = 0; i < nky * nkx; ++i) sum += h_kernel[i];
    SPC_REQUIRE(!(sum < 1e-8 && sum > -1e-8) && sum >= 1e-8,
                "The kernel can't be normalized, because its sum is close to zero");
    SpArgs A{};
    rc = fill_args(A, cube, mask, d_out, out_row_stride, out_plane_stride);
    if (rc) return rc;
    SPC_REQUIRE(cube->nz <= 65535, "nz > 65535 planes per call not supported by the 2-D kernel grid");
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    const int n = nky * nkx;
    float* d_k = nullptr;
    SPC_HIP(hipMallocAsync((void**)&d_k, sizeof(float) * n, st));
    float* hk = (float*)malloc(sizeof(float) * n);
    for (int i = 0; i < n; ++i) hk[i] = (float)h_kernel[i];
    hipError_t e = hipMemcpyAsync(d_k, hk, sizeof(float) * n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    free(hk);
    SPC_HIP(e);
    dim3 grid((unsigned)((cube->nx + 63) / 64), (unsigned)((cube->ny + 3) / 4), (unsigned)cube->nz);
    hipLaunchKernelGGL(spatial_conv2d_kernel, grid, dim3(256), 0, st, A, d_k, nky, nkx);
    SPC_LAUNCH_CHECK();
    SPC_HIP(hipFreeAsync(d_k, st));
    return SPC_OK;
}

int spc_spatial_conv_sep_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                             const double* h_ky, int nky, const double* h_kx, int nkx, float* d_out,
                             int64_t out_row_stride, int64_t out_plane_stride) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    if ((rc = check_kernel(h_ky, nky, "y")) || (rc = check_kernel(h_kx, nkx, "x"))) return rc;
    double sy = 0.0, sx = 0.0;
    for (int i = 0; i < nky; ++i) sy += h_ky[i];
    for (int i = 0; i < nkx; ++i) sx += h_kx[i];
    const double sum = sy * sx;
    SPC_REQUIRE(!(sum < 1e-8 && sum > -1e-8) && sum >= 1e-8,
                "The kernel can't be normalized, because its sum is close to zero");
    const int R = pick_ring(std::max(nky, nkx));
    if (!R) {
        // very wide kernels: materialise the outer product and use the direct kernel
        const size_t n = (size_t)nky * nkx;
        double* k2 = (double*)malloc(sizeof(double) * n);
        if (!k2) { spc_set_error("out of host memory"); return SPC_ERR_NOMEM; }
        for (int a = 0; a < nky; ++a)
            for (int b = 0; b < nkx; ++b) k2[(size_t)a * nkx + b] = h_ky[a] * h_kx[b];
        rc = spc_spatial_conv2d_f32(device, stream, cube, mask, k2, nky, nkx, d_out, out_row_stride, out_plane_stride);
        free(k2);
        return rc;
    }
    SpArgs A{};
    rc = fill_args(A, cube, mask, d_out, out_row_stride, out_plane_stride);
    if (rc) return rc;
    pad_taps(A.ky, h_ky, nky, R);
    pad_taps(A.kx, h_kx, nkx, R);
    SPC_REQUIRE(cube->nz <= 65535, "nz > 65535 planes per call not supported (split the call)");
    SPC_DEVICE(device);
    const int HX = R / 2;
    A.txo = ((kThreads - 2 * HX) / kRun) * kRun;
    const int64_t nstrips = (cube->nx + A.txo - 1) / A.txo;
    // y split only when channels x strips cannot fill the chip
    int nysplit = 1;
    const int64_t nblocks = nstrips * cube->nz;
    if (nblocks < 1024 && cube->ny >= 8 * R) {
        nysplit = (int)std::min<int64_t>((1024 + nblocks - 1) / nblocks, cube->ny / (4 * R));
        nysplit = std::max(nysplit, 1);
    }
    A.ychunk = (cube->ny + nysplit - 1) / nysplit;
    nysplit = (int)((cube->ny + A.ychunk - 1) / A.ychunk);
    dim3 grid((unsigned)nstrips, (unsigned)cube->nz, (unsigned)nysplit);
    hipStream_t st = (hipStream_t)stream;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    switch (R) {
        case 9: return launch_sep<9>(A, st, grid, arr);
        case 17: return launch_sep<17>(A, st, grid, arr);
        case 29: return launch_sep<29>(A, st, grid, arr);
        case 33: return launch_sep<33>(A, st, grid, arr);
        case 65: return launch_sep<65>(A, st, grid, arr);
    }
    spc_set_error("no ring kernel for R=%d", R);
    return SPC_ERR_UNSUPPORTED;
}

}  // extern "C"
