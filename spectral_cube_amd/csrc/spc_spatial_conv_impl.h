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


inline int pick_ring(int ntaps) {
    const int rings[] = {9, 17, 29, 33, 65};
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


}  // namespace spc_spconv
