// Kernel templates of the ring-streaming spectral stencil; included by the
// per-ring-size translation units spc_spectral_conv_r*.hip (one TU per R so
// `make -j` builds the fully unrolled kernels in parallel).
#pragma once
#include "spc_common.h"
#include <algorithm>

namespace spc_sconv {


constexpr int kMaxTaps = 65;
typedef float float2v __attribute__((ext_vector_type(2)));

struct ConvArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    MaskDev mask;
    float* out;
    int64_t out_row_stride, out_plane_stride;
    int64_t zchunk;            // outputs per gridDim.y slice
    // fused-moment part
    const double* cen;
    double dv, m1_add;
    spc_moment_outputs mo;
    int64_t mo_row_stride;
    float k[kMaxTaps];         // padded to R taps, centred
};

// fused-moment running state of one spaxel
struct MomState {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    int nvalid = 0;
    float bmax = -INFINITY, bmin = INFINITY;
    int imax = 0, imin = 0;
};

// One revolution of the ring: R consecutive input channels i0 .. i0+R-1.
// FAST = every input is inside [0,nz) and every completing output inside
// [zb,ze): no range predicates, which keeps the scalar register pressure low.
template <int R, bool ARR, bool FUSE, bool EXT, bool SYM, bool FAST>
__device__ __forceinline__ void ring_revolution(const ConvArgs& A, float2v (&acc)[R],
                                                unsigned long long& inc_hist, MomState& ms,
                                                const float* p, const uint8_t* pm, float* po,
                                                int64_t i0, int64_t zb, int64_t ze) {
    constexpr int H = R / 2;
    const uint32_t flags = A.mask.flags;
    const float tlo = A.mask.thr_lo, thi = A.mask.thr_hi;
    float v[R];
    unsigned char mk[R];
    if (FAST) {
        const float* q = p + i0 * A.plane_stride;
        const uint8_t* qm = ARR ? pm + i0 * A.mask.plane_stride : nullptr;
#pragma unroll
        for (int s = 0; s < R; ++s) {
            v[s] = __builtin_nontemporal_load(q);
            q += A.plane_stride;
            if (ARR) { mk[s] = __builtin_nontemporal_load(qm); qm += A.mask.plane_stride; }
        }
    } else {
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int64_t ic = min(max(i0 + s, (int64_t)0), A.nz - 1);
            v[s] = p[ic * A.plane_stride];
            if (ARR) mk[s] = pm[ic * A.mask.plane_stride];
        }
    }
#pragma unroll
    for (int s = 0; s < R; ++s) {
        const int64_t i = i0 + s;
        const bool inr = FAST ? true : ((i >= 0) && (i < A.nz));
        bool inc = spc_pred(flags, tlo, thi, v[s]);
        if (ARR) inc = inc && (mk[s] != 0);
        inc = inc && inr;
        // out-of-range samples are VALID ZEROS (boundary='fill', fill_value=0)
        const bool ok = inr ? (inc && (v[s] == v[s])) : true;
        float2v x2;
        x2.x = (ok && inr) ? v[s] : 0.f;
        x2.y = ok ? 1.f : 0.f;
        inc_hist = (inc_hist << 1) | (inc ? 1ull : 0ull);
#pragma unroll
        for (int m = 0; m < R; ++m) {
            const int a = (s - m + R) % R;      // age of the output living in slot m
            // symmetric kernels: half the distinct weights -> they all stay in SGPRs
            const float wgt = A.k[SYM ? (a <= H ? a : 2 * H - a) : 2 * H - a];
            const float2v w2 = float2v{wgt, wgt};
            if (a == 0) acc[m] = w2 * x2;
            else acc[m] = __builtin_elementwise_fma(w2, x2, acc[m]);
            // pin the update here: without it LLVM sinks each slot's whole FMA
            // chain down to its emission point -> R-long DEPENDENT chains with
            // all R inputs live (200 VGPRs) instead of R independent FMAs/step
            asm volatile("" : "+v"(acc[m]));
        }
        // the output that just received its last contribution
        const int64_t o = i - H;
        if (FAST || (o >= zb && o < ze)) {
            const float2v r = acc[(s + 1) % R];
            const bool inc_o = ((inc_hist >> H) & 1ull) != 0ull;
            float res;
            if (r.y != 0.f) res = r.x * __builtin_amdgcn_rcpf(r.y);  // 1 ulp; tolerance is 1e-5
            else res = inc_o ? p[o * A.plane_stride] : NAN;  // astropy: empty window -> (filled) centre sample
            if (!FUSE) {
                po[o * A.out_plane_stride] = res;
            } else {
                const bool okm = inc_o && (res == res);
                const double wd = okm ? (double)res : 0.0;
                const double c = A.cen[o];
                ms.s0 += wd;
                ms.s1 = fma(wd, c, ms.s1);
                ms.s2 = fma(wd, c * c, ms.s2);
                ms.nvalid += okm ? 1 : 0;
                if (EXT) {
                    const float hi = okm ? res : -INFINITY;
                    const float lo = okm ? res : INFINITY;
                    if (hi > ms.bmax) { ms.bmax = hi; ms.imax = (int)o; }
                    if (lo < ms.bmin) { ms.bmin = lo; ms.imin = (int)o; }
                }
            }
        }
    }
}

template <int R, bool ARR, bool FUSE, bool EXT, bool SYM>
__global__ __launch_bounds__(256) void spectral_conv_kernel(const ConvArgs A) {
    constexpr int H = R / 2;
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ny * A.nx) return;
    const int64_t y = col / A.nx, x = col - y * A.nx;
    const int64_t zb = (int64_t)blockIdx.y * A.zchunk;
    const int64_t ze = min(A.nz, zb + A.zchunk);
    const float* p = A.cube + y * A.row_stride + x;
    const uint8_t* pm = ARR ? A.mask.arr + y * A.mask.row_stride + x : nullptr;
    float* po = FUSE ? nullptr : A.out + y * A.out_row_stride + x;

    float2v acc[R];
#pragma unroll
    for (int m = 0; m < R; ++m) acc[m] = float2v{0.f, 0.f};
    unsigned long long inc_hist = 0ull;  // include bit of the last 64 inputs (bit 0 = newest)
    MomState ms;

    const int64_t T = (ze - zb) + 2 * H;  // number of input steps
    for (int64_t t0 = 0; t0 < T; t0 += R) {
        const int64_t i0 = zb - H + t0;
        // all inputs in range and all R completing outputs (i0-H .. i0+R-1-H) wanted?
        const bool fast = (i0 - H >= zb) && (i0 + R - 1 < A.nz) && (i0 + R - 1 - H < ze);
        if (fast) ring_revolution<R, ARR, FUSE, EXT, SYM, true>(A, acc, inc_hist, ms, p, pm, po, i0, zb, ze);
        else ring_revolution<R, ARR, FUSE, EXT, SYM, false>(A, acc, inc_hist, ms, p, pm, po, i0, zb, ze);
    }

    if (FUSE) {
        const int64_t o = y * A.mo_row_stride + x;
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        const double mu = ms.s1 / ms.s0;
        if (A.mo.d_m0) A.mo.d_m0[o] = ms.nvalid > 0 ? A.dv * ms.s0 : nan;
        if (A.mo.d_m1) A.mo.d_m1[o] = mu + A.m1_add;
        if (A.mo.d_m2) A.mo.d_m2[o] = ms.s2 / ms.s0 - mu * mu;
        if (A.mo.d_mu) A.mo.d_mu[o] = mu;
        if (A.mo.d_s0) A.mo.d_s0[o] = ms.s0;
        if (A.mo.d_argmax) A.mo.d_argmax[o] = ms.nvalid > 0 ? (int64_t)ms.imax : 0;
        if (A.mo.d_argmin) A.mo.d_argmin[o] = ms.nvalid > 0 ? (int64_t)ms.imin : 0;
        if (A.mo.d_vmax) A.mo.d_vmax[o] = ms.nvalid > 0 ? ms.bmax : NAN;
        if (A.mo.d_vmin) A.mo.d_vmin[o] = ms.nvalid > 0 ? ms.bmin : NAN;
        if (A.mo.d_nvalid) A.mo.d_nvalid[o] = ms.nvalid;
    }
}


template <int R, bool FUSE, bool SYM>
int launch_rs(const ConvArgs& A, hipStream_t st, dim3 grid, bool arr, bool ext) {
    dim3 block(256);
    if (arr) {
        if (FUSE && ext) hipLaunchKernelGGL((spectral_conv_kernel<R, true, FUSE, FUSE, SYM>), grid, block, 0, st, A);
        else hipLaunchKernelGGL((spectral_conv_kernel<R, true, FUSE, false, SYM>), grid, block, 0, st, A);
    } else {
        if (FUSE && ext) hipLaunchKernelGGL((spectral_conv_kernel<R, false, FUSE, FUSE, SYM>), grid, block, 0, st, A);
        else hipLaunchKernelGGL((spectral_conv_kernel<R, false, FUSE, false, SYM>), grid, block, 0, st, A);
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

template <int R, bool FUSE>
int launch_r(const ConvArgs& A, hipStream_t st, dim3 grid, bool arr, bool ext) {
    bool sym = true;
    for (int i = 0; i < R / 2; ++i) sym = sym && (A.k[i] == A.k[R - 1 - i]);
    return sym ? launch_rs<R, FUSE, true>(A, st, grid, arr, ext) : launch_rs<R, FUSE, false>(A, st, grid, arr, ext);
}


// entry point instantiated once per ring size in spc_spectral_conv_r<R>.hip
template <int R>
int launch(const ConvArgs& A, hipStream_t st, dim3 grid, bool arr, bool fuse, bool ext) {
    return fuse ? launch_r<R, true>(A, st, grid, arr, ext) : launch_r<R, false>(A, st, grid, arr, ext);
}

}  // namespace spc_sconv
