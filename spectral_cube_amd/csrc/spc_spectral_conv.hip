// NaN-aware 1-D convolution along the spectral axis (astropy.convolution.
// convolve semantics: boundary='fill' fill_value=0, nan_treatment=
// 'interpolate', normalize_kernel=True), optionally fused with the masked
// moment reduction so the smoothed cube is never written.
//
// Replaces the chunk function of DaskSpectralCubeMixin.spectral_smooth
// (spectral_cube/dask_spectral_cube.py:880-917; NumPy twin
// spectral_cube/spectral_cube.py:3186-3222) and, fused, the following
// DaskSpectralCubeMixin.moment (dask_spectral_cube.py:1083-1104).
//
// Streaming ring: one lane owns one spaxel and marches over z.  The R = 2H+1
// outputs currently "in flight" (those the newest input still contributes to)
// live in R packed (num, den) fp32 accumulators; each new input is loaded ONCE,
// turned into (v*ok, ok) and FMA-ed into all R accumulators with the R kernel
// weights (v_pk_fma_f32: numerator and NaN-renormalising denominator ride in
// one instruction); the oldest output then completes and is emitted.  The
// loop is unrolled by R so every ring slot and weight index is static.  No
// halo re-reads, no LDS: the cube is read exactly once (4 B/voxel) and, when
// materialised, written once.
#include "spc_common.h"
#include <algorithm>
#include <alloca.h>

namespace {

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

// generic fallback for kernels wider than kMaxTaps taps (runtime tap loop,
// re-reads served by L1/L2)
__global__ __launch_bounds__(256) void spectral_conv_generic_kernel(const ConvArgs A, const float* kern, int ntaps) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ny * A.nx) return;
    const int64_t y = col / A.nx, x = col - y * A.nx;
    const int H = ntaps / 2;
    const float* p = A.cube + y * A.row_stride + x;
    const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
    const uint8_t* pm = arr ? A.mask.arr + y * A.mask.row_stride + x : nullptr;
    const int64_t zb = (int64_t)blockIdx.y * A.zchunk;
    const int64_t ze = min(A.nz, zb + A.zchunk);
    for (int64_t o = zb; o < ze; ++o) {
        float num = 0.f, den = 0.f;
        for (int j = 0; j < ntaps; ++j) {
            const int64_t i = o + H - j;  // weight kern[o - i + H] = kern[j]
            float v = 0.f; bool ok = true;
            if (i >= 0 && i < A.nz) {
                v = p[i * A.plane_stride];
                bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, v);
                if (arr) inc = inc && pm[i * A.mask.plane_stride] != 0;
                ok = inc && (v == v);
                if (!ok) v = 0.f;
            }
            num = fmaf(kern[j], v, num);
            den = fmaf(kern[j], ok ? 1.f : 0.f, den);
        }
        float res;
        if (den != 0.f) {
            res = num / den;
        } else {  // empty window -> (filled) centre sample
            const float c = p[o * A.plane_stride];
            bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, c);
            if (arr) inc = inc && pm[o * A.mask.plane_stride] != 0;
            res = inc ? c : NAN;
        }
        A.out[y * A.out_row_stride + x + o * A.out_plane_stride] = res;
    }
}

int pick_ring(int ntaps) {
    const int rings[] = {9, 17, 33, 65};
    for (int r : rings) if (ntaps <= r) return r;
    return 0;
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

template <bool FUSE>
int launch_ring(int R, const ConvArgs& A, hipStream_t st, dim3 grid, bool arr, bool ext) {
    switch (R) {
        case 9: return launch_r<9, FUSE>(A, st, grid, arr, ext);
        case 17: return launch_r<17, FUSE>(A, st, grid, arr, ext);
        case 33: return launch_r<33, FUSE>(A, st, grid, arr, ext);
        case 65: return launch_r<65, FUSE>(A, st, grid, arr, ext);
    }
    spc_set_error("no ring kernel for R=%d", R);
    return SPC_ERR_UNSUPPORTED;
}

int fill_common(ConvArgs& A, const spc_cube_f32* cube, const spc_mask* mask, const double* h_kernel,
                int ntaps, int R) {
    int rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    for (int i = 0; i < kMaxTaps; ++i) A.k[i] = 0.f;
    if (R) {
        const int pad = (R - ntaps) / 2;
        for (int i = 0; i < ntaps; ++i) A.k[pad + i] = (float)h_kernel[i];
    }
    return SPC_OK;
}

int check_kernel(const double* h_kernel, int ntaps) {
    SPC_REQUIRE(h_kernel != nullptr, "kernel pointer is NULL");
    SPC_REQUIRE(ntaps >= 1 && (ntaps % 2) == 1, "kernel must have an odd number of taps (got %d)", ntaps);
    double sum = 0.0;
    for (int i = 0; i < ntaps; ++i) sum += h_kernel[i];
    // astropy: "The kernel can't be normalized, because its sum is close to zero"
    SPC_REQUIRE(!(sum < 1e-8 && sum > -1e-8) && sum >= 1.0 / 1e8,
                "The kernel can't be normalized, because its sum is close to zero");
    return SPC_OK;
}

}  // namespace

extern "C" {

int spc_spectral_conv_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                          const double* h_kernel, int ntaps, float* d_out, int64_t out_row_stride,
                          int64_t out_plane_stride) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    rc = check_kernel(h_kernel, ntaps);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    const int R = pick_ring(ntaps);
    ConvArgs A{};
    rc = fill_common(A, cube, mask, h_kernel, ntaps, R);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.out = d_out;
    A.out_row_stride = out_row_stride ? out_row_stride : cube->nx;
    A.out_plane_stride = out_plane_stride ? out_plane_stride : cube->ny * A.out_row_stride;
    const int64_t ncols = cube->ny * cube->nx;
    const int64_t nblocks = (ncols + 255) / 256;
    // z split for small maps (each slice re-reads a 2H halo)
    int nsplit = 1;
    if (nblocks < 1024 && cube->nz >= 8 * (R ? R : ntaps)) {
        nsplit = (int)std::min<int64_t>((1024 + nblocks - 1) / nblocks, cube->nz / (4 * (R ? R : ntaps)));
        nsplit = std::max(nsplit, 1);
    }
    A.zchunk = (cube->nz + nsplit - 1) / nsplit;
    nsplit = (int)((cube->nz + A.zchunk - 1) / A.zchunk);
    dim3 grid((unsigned)nblocks, (unsigned)nsplit);
    hipStream_t st = (hipStream_t)stream;
    if (R) return launch_ring<false>(R, A, st, grid, (A.mask.flags & SPC_MASK_ARRAY) != 0, false);
    // wide kernels: generic path with the taps in device memory
    float* d_k = nullptr;
    SPC_HIP(hipMallocAsync((void**)&d_k, sizeof(float) * ntaps, st));
    float* hk = (float*)alloca(sizeof(float) * ntaps);
    for (int i = 0; i < ntaps; ++i) hk[i] = (float)h_kernel[i];
    SPC_HIP(hipMemcpyAsync(d_k, hk, sizeof(float) * ntaps, hipMemcpyHostToDevice, st));
    SPC_HIP(hipStreamSynchronize(st));  // hk lives on this stack frame
    hipLaunchKernelGGL(spectral_conv_generic_kernel, grid, dim3(256), 0, st, A, d_k, ntaps);
    SPC_LAUNCH_CHECK();
    SPC_HIP(hipFreeAsync(d_k, st));
    return SPC_OK;
}

int spc_spectral_conv_moments_f32(int device, void* stream, const spc_cube_f32* cube,
                                  const spc_mask* mask, const double* h_kernel, int ntaps,
                                  const double* d_cen, double dv, double m1_add,
                                  const spc_moment_outputs* out) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    rc = check_kernel(h_kernel, ntaps);
    if (rc) return rc;
    SPC_REQUIRE(out != nullptr && d_cen != nullptr, "NULL pointer argument");
    const int R = pick_ring(ntaps);
    if (!R) {
        spc_set_error("fused smooth->moments supports up to %d taps (got %d); "
                      "materialise with spc_spectral_conv_f32 instead", kMaxTaps, ntaps);
        return SPC_ERR_UNSUPPORTED;
    }
    ConvArgs A{};
    rc = fill_common(A, cube, mask, h_kernel, ntaps, R);
    if (rc) return rc;
    SPC_DEVICE(device);
    A.cen = d_cen; A.dv = dv; A.m1_add = m1_add;
    A.mo = *out;
    A.mo_row_stride = out->out_row_stride ? out->out_row_stride : cube->nx;
    A.zchunk = cube->nz;
    const int64_t ncols = cube->ny * cube->nx;
    dim3 grid((unsigned)((ncols + 255) / 256), 1);
    const bool ext = out->d_argmax || out->d_argmin || out->d_vmax || out->d_vmin;
    return launch_ring<true>(R, A, (hipStream_t)stream, grid, (A.mask.flags & SPC_MASK_ARRAY) != 0, ext);
}

}  // extern "C"
