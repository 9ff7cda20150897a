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

#include "spc_spectral_conv_impl.h"
#include <cmath>
#include <cstdlib>
#include <vector>

namespace spc_sconv {
extern template int launch<9>(const ConvArgs&, hipStream_t, int, bool);
extern template int launch<17>(const ConvArgs&, hipStream_t, int, bool);
extern template int launch<33>(const ConvArgs&, hipStream_t, int, bool);
}
using namespace spc_sconv;

namespace {

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

// ring kernels need a non-zero centre tap (see the empty-window note in
// spc_spectral_conv_impl.h); anything else goes to the generic kernel
int pick_ring(const double* k, int ntaps) {
    if (k[ntaps / 2] == 0.0) return 0;
    // (a 63-tap ring compiles for > 10 minutes; wider kernels use the generic kernel)
    const int rings[] = {9, 17, 33};
    for (int r : rings) if (ntaps <= r) return r;
    return 0;
}

int launch_ring_raw(int R, const ConvArgs& A, hipStream_t st, int fast, bool fuse) {
    switch (R) {
        case 9: return spc_sconv::launch<9>(A, st, fast, fuse);
        case 17: return spc_sconv::launch<17>(A, st, fast, fuse);
        case 33: return spc_sconv::launch<33>(A, st, fast, fuse);
    }
    spc_set_error("no ring kernel for R=%d", R);
    return SPC_ERR_UNSUPPORTED;
}

// vec == 2 (8-byte aligned rows), no mask array and a single z slice: speculate that the
// data has no invalid samples (all-valid fast kernel), then redo only the dirty tiles
int launch_ring(int R, ConvArgs& A, hipStream_t st, int vec, bool fuse) {
    const char* env = getenv("SPC_CONV_FAST");
    const bool want = env ? atoi(env) != 0 : true;
    // (the fast kernel addresses R+1 planes through one descriptor + 32-bit scalar offsets)
    const bool fits = (A.plane_stride * 4 * (R + 1) < (1LL << 31)) && (A.out_plane_stride * 4 * (R + 1) < (1LL << 31));
    const bool fast = want && fits && vec == 2 && (A.mask.flags & ~(uint32_t)SPC_MASK_FINITE) == 0 && A.zchunk >= A.nz;
    A.status = nullptr;
    if (!fast) return launch_ring_raw(R, A, st, 0, fuse);
    const size_t ntiles = (size_t)((A.ny * A.nx / 2 + 63) / 64);
    unsigned char* d_status = nullptr;
    SPC_HIP(hipMallocAsync((void**)&d_status, ntiles, st));
    SPC_HIP(hipMemsetAsync(d_status, 0, ntiles, st));
    A.status = d_status;
    const int rc = launch_ring_raw(R, A, st, 1, fuse);
    SPC_HIP(hipFreeAsync(d_status, st));
    return rc;
}

// two spaxels per lane need 8-byte aligned rows everywhere
int pick_vec(const spc_cube_f32* c, const MaskDev& m, const float* out, int64_t out_row, int64_t out_plane) {
    const char* env = getenv("SPC_CONV_VEC");
    if (env && atoi(env) == 1) return 1;
    bool ok = (c->nx % 2 == 0) && (c->row_stride % 2 == 0) && (c->plane_stride % 2 == 0) &&
              (((uintptr_t)c->d_data) % 8 == 0);
    if (m.flags & SPC_MASK_ARRAY)
        ok = ok && (m.row_stride % 2 == 0) && (m.plane_stride % 2 == 0) && (((uintptr_t)m.arr) % 2 == 0);
    if (out) ok = ok && (out_row % 2 == 0) && (out_plane % 2 == 0) && (((uintptr_t)out) % 8 == 0);
    return ok ? 2 : 1;
}

int fill_common(ConvArgs& A, const spc_cube_f32* cube, const spc_mask* mask, const double* h_kernel,
                int ntaps, int R) {
    int rc = spc_mask_to_dev(mask, cube, &A.mask);
    if (rc) return rc;
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    for (int i = 0; i < 64; ++i) A.k[i] = 0.f;
    for (int i = 0; i < 130; ++i) A.kPS[i] = 0.f;
    if (R) {
        const int pad = (R - ntaps) / 2;
        for (int i = 0; i < ntaps; ++i) A.k[pad + i] = (float)h_kernel[i];
        { double ks = 0.0; for (int i = 0; i < R; ++i) ks += (double)A.k[i]; A.inv_ksum = (float)(1.0 / ks); }
        double acc = 0.0;                      // prefix / suffix sums of the float32 taps
        for (int i = 0; i < R; ++i) { acc += (double)A.k[i]; A.kPS[2 * i] = (float)acc; }
        acc = 0.0;
        for (int i = R - 1; i >= 0; --i) { A.kPS[2 * i + 1] = (float)acc; acc += (double)A.k[i]; }
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
    const int R = pick_ring(h_kernel, ntaps);
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
    if (R) {
        SPC_REQUIRE(cube->ny * cube->row_stride < (1LL << 29) && cube->ny * A.out_row_stride < (1LL << 29),
                    "image plane too large for 32-bit buffer offsets");
        const int vec = pick_vec(cube, A.mask, d_out, A.out_row_stride, A.out_plane_stride);
        return launch_ring(R, A, st, vec, false);
    }
    // wide kernels: generic path with the taps in device memory (rare fallback:
    // plain synchronous allocation/copy, released after the kernel has drained)
    float* d_k = nullptr;
    SPC_HIP(hipMalloc((void**)&d_k, sizeof(float) * ntaps));
    std::vector<float> hk(ntaps);
    for (int i = 0; i < ntaps; ++i) hk[i] = (float)h_kernel[i];
    hipError_t e = hipMemcpy(d_k, hk.data(), sizeof(float) * ntaps, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(spectral_conv_generic_kernel, grid, dim3(256), 0, st, A, d_k, ntaps);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    (void)hipFree(d_k);
    SPC_HIP(e);
    return SPC_OK;
}

int spc_spectral_conv_moments_f32(int device, void* stream, const spc_cube_f32* cube,
                                  const spc_mask* mask, const double* h_kernel, int ntaps,
                                  const double* d_cen, const double* h_cen, double dv, double m1_add,
                                  const spc_moment_outputs* out) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    rc = check_kernel(h_kernel, ntaps);
    if (rc) return rc;
    SPC_REQUIRE(out != nullptr && d_cen != nullptr, "NULL pointer argument");
    const int R = pick_ring(h_kernel, ntaps);
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
    // linear spectral axis (every FITS axis is): c[z] = c0 + z*dc -> no per-channel loads
    A.cen_linear = 0;
    if (h_cen && cube->nz >= 2) {
        const double c0 = h_cen[0], dc = (h_cen[cube->nz - 1] - h_cen[0]) / (double)(cube->nz - 1);
        double span = std::abs(dc) * (double)cube->nz, worst = 0.0;
        for (int64_t z = 0; z < cube->nz; ++z) worst = std::max(worst, std::abs(h_cen[z] - (c0 + dc * (double)z)));
        if (worst <= 1e-13 * std::max(span, 1e-300)) { A.cen_linear = 1; A.cen_c0 = c0; A.cen_dc = dc; }
    }
    A.mo = *out;
    A.mo_row_stride = out->out_row_stride ? out->out_row_stride : cube->nx;
    A.zchunk = cube->nz;
    SPC_REQUIRE(cube->ny * cube->row_stride < (1LL << 29), "image plane too large for 32-bit buffer offsets");
    const int vec = pick_vec(cube, A.mask, nullptr, 0, 0);
    return launch_ring(R, A, (hipStream_t)stream, vec, true);
}

}  // extern "C"
