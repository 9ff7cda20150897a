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
// live in R (num, den) float64 accumulators (astropy accumulates in float64 and
// rounds once to float32; so does every kernel here); each new input is loaded
// ONCE, turned into (v*ok, ok) and FMA-ed into all R accumulators with the R kernel
// weights; the oldest output then completes and is emitted.  The
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
extern template int launch_fast_only<49>(const ConvArgs&, hipStream_t);
extern template int launch_fast_only<65>(const ConvArgs&, hipStream_t);
extern template int launch_ring_wide<49>(const ConvArgs&, hipStream_t);
extern template int launch_ring_wide<65>(const ConvArgs&, hipStream_t);
}
using namespace spc_sconv;

namespace {

// generic fallback for kernels wider than kMaxTaps taps (runtime tap loop,
// re-reads served by L1/L2)
__global__ __launch_bounds__(256) void spectral_conv_generic_kernel(const ConvArgs A, const double* kern, int ntaps) {
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
        double num = 0.0, den = 0.0;
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
            num = fma(kern[j], (double)v, num);
            den = fma(kern[j], ok ? 1.0 : 0.0, den);
        }
        float res;
        if (den != 0.0) {
            res = (float)(num / den);
        } else {  // empty window -> (filled) centre sample
            const float c = p[o * A.plane_stride];
            bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, c);
            if (arr) inc = inc && pm[o * A.mask.plane_stride] != 0;
            res = inc ? c : NAN;
        }
        A.out[y * A.out_row_stride + x + o * A.out_plane_stride] = res;
    }
}

// ---- wide kernels (more taps than the largest ring): runs of 16 outputs along z -------------------
// A lane owns one spaxel (lanes along x: coalesced) and produces 16 consecutive output channels at a
// time: it walks the ntaps + 15 input planes of the run once (one load + classification each) and
// feeds every sample into up to 16 float64 (num, den) accumulator pairs, the weights coming in as
// wave-uniform scalars - 23 per chunk of 8 planes: kpad[15 + j] = k[j] with 15 zeros of padding on
// both sides that are never multiplied (which (plane, output) pairs exist is static in the two head
// and the two tail chunks).  Consecutive runs re-read their (ntaps - 1)-plane halo from L2.  Replaces
// the per-output tap loop (81 taps at 1024^3: 130 ms) for every kernel of 17 taps or more that has
// no ring; same astropy semantics (zero fill, NaN renormalisation, empty window -> centre sample).
struct double2w { double x, y; };

// ALLV: speculative all-valid form - numerators only (the denominator of an all-valid window is the kernel sum: samples
// outside the cube are valid zeros), half the FMAs; a wavefront that meets an excluded sample flags its tile and quits,
// the (num, den) form then redoes the flagged tiles.
template <bool ARR, bool ALLV>
__global__ __launch_bounds__(256, 3) void spectral_conv_wide_kernel(const ConvArgs A, const double* kpad, int ntaps) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= A.ny * A.nx) return;
    // tiles (128 columns) already finished by an all-valid pass
    if (!ALLV && A.status && spc_flag_get(A.status + __builtin_amdgcn_readfirstlane((int)(col >> 7))) == 0) return;
    bool bad = false;
    const int64_t y = col / A.nx, x = col - y * A.nx;
    const int H = ntaps / 2;
    const float* p = A.cube + y * A.row_stride + x;
    const uint8_t* pm = ARR ? A.mask.arr + y * A.mask.row_stride + x : nullptr;
    const int64_t zb = (int64_t)blockIdx.y * A.zchunk;
    const int64_t ze = min(A.nz, zb + A.zchunk);
    for (int64_t o0 = zb; o0 < ze; o0 += 16) {
        double2w acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = double2w{0.0, 0.0};
        // output o0 + q, tap j reads plane o0 + q + H - j = (o0 - H) + r with r = q + ntaps - 1 - j
        auto sample = [&](int r) -> double2w {
            const int64_t i = o0 - H + r;
            if (i < 0 || i >= A.nz) return double2w{0.0, 1.0};       // outside the cube: a valid zero
            const float v = p[i * A.plane_stride];
            bool ok = spc_pred_valid(A.mask, v);
            if (ARR) ok = ok && pm[i * A.mask.plane_stride] != 0;
            if (ALLV) bad = bad || !ok;
            return ok ? double2w{(double)v, 1.0} : double2w{0.0, 0.0};
        };
        // chunk of 8 planes r0 .. r0+7; weight of (plane r0 + i, output q) = k[ntaps-1-(r0+i)+q] = wp[7 - i + q],
        // wp = kpad + 15 + (ntaps - 1 - r0) - 7.  mode 0: q <= i (+ off), mode 2: q >= i (+ off), mode 1: all.
        auto chunk = [&](int r0, int mode, int off, int nrows) {
            // (read through the constant address space: wave-uniform scalar loads, the taps stay out of the VGPRs)
            typedef const double __attribute__((address_space(4))) cdouble;
            cdouble* wp = (cdouble*)(kpad + 15 + (ntaps - 1 - r0) - 7);
            double w[23];
#pragma unroll
            for (int t = 0; t < 23; ++t) w[t] = wp[t];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (mode == 1 && i >= nrows) break;                   // wave-uniform
                const double2w in = sample(r0 + i);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool on = (mode == 0) ? (q <= i + off) : (mode == 2) ? (q >= i + off) : true;
                    if (on) { acc[q].x = fma(w[7 - i + q], in.x, acc[q].x); if (!ALLV) acc[q].y = fma(w[7 - i + q], in.y, acc[q].y); }
                }
            }
        };
        chunk(0, 0, 0, 8);                                  // planes 0..7:   outputs q <= r
        chunk(8, 0, 8, 8);                                  // planes 8..15:  outputs q <= r (ntaps >= 17: r - q <= ntaps - 1 holds)
        int r0 = 16;
        for (; r0 + 8 <= ntaps - 1; r0 += 8) chunk(r0, 1, 0, 8);
        if (r0 < ntaps - 1) chunk(r0, 1, 0, ntaps - 1 - r0);
        chunk(ntaps - 1, 2, 0, 8);                          // planes ntaps-1 .. ntaps+6: outputs q >= r - (ntaps-1)
        chunk(ntaps + 7, 2, 8, 8);                          // planes ntaps+7 .. ntaps+14
        if (ALLV) {
            if (__any(bad)) {                              // wave-uniform: the (num, den) kernel redoes this tile
                if ((threadIdx.x & 63) == 0) spc_flag_set(A.status + (col >> 7));
                return;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t o = o0 + q;
                if (o >= ze) break;
                A.out[y * A.out_row_stride + x + o * A.out_plane_stride] = (float)(acc[q].x / A.ksum);
            }
            continue;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t o = o0 + q;
            if (o >= ze) break;
            float res;
            if (acc[q].y != 0.0) res = (float)(acc[q].x / acc[q].y);
            else {                                          // empty window -> (filled) centre sample
                const float c = p[o * A.plane_stride];
                bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, c);
                if (ARR) inc = inc && pm[o * A.mask.plane_stride] != 0;
                res = inc ? c : NAN;
            }
            A.out[y * A.out_row_stride + x + o * A.out_plane_stride] = res;
        }
    }
}

// ring kernels need a non-zero centre tap (see the empty-window note in
// spc_spectral_conv_impl.h); anything else goes to the generic kernel
// The ring kernels address the R + 1 planes of a revolution through ONE buffer descriptor (4 GiB
// range) plus 32-bit offsets: lane offset inside a plane + plane delta.  Larger planes (beyond
// ~5600^2 for 33 taps) take the kernels without a ring.
bool ring_fits(int R, const spc_cube_f32* c, const spc_mask* m, int64_t out_row_stride, int64_t out_plane_stride) {
    const int64_t lim = 1LL << 32;
    bool ok = (R + 1) * c->plane_stride * 4 + c->ny * c->row_stride * 4 < lim && c->ny * c->row_stride < (1LL << 29);
    if (out_plane_stride) ok = ok && (R + 1) * out_plane_stride * 4 + c->ny * out_row_stride * 4 < lim && c->ny * out_row_stride < (1LL << 29);
    if (m && (m->flags & SPC_MASK_ARRAY)) {
        const int64_t mr = m->row_stride ? m->row_stride : c->row_stride, mp = m->plane_stride ? m->plane_stride : c->plane_stride;
        ok = ok && (R + 1) * mp + c->ny * mr < lim;
    }
    return ok;
}

int pick_ring(const double* k, int ntaps) {
    if (k[ntaps / 2] == 0.0) return 0;
    // (a 63-tap ring compiles for > 10 minutes; wider kernels use the generic kernel)
    const int rings[] = {9, 17, 33};
    for (int r : rings) if (ntaps <= r) return r;
    return 0;
}

// ---- fused smooth -> moments without the stencil ------------------------------------------
// With every sample valid the smoothed spectrum is LINEAR in the data (astropy's NaN-free
// branch: s_o = sum_i k[o + H - i] d_i / sum(k), out-of-range samples are zeros), so
//   S_n' = sum_o c_o^n s_o = sum_i d_i W_n(i),   W_n(i) = sum_o k[o + H - i] c_o^n / sum(k):
// the three moment sums of the smoothed cube are weighted sums over the UNSMOOTHED data with
// per-channel weights computed on the host in float64 (nz x ntaps operations).  The 33-tap
// stencil (33 packed FMAs per voxel pair, VALU-bound at 49 % of the HBM roofline) becomes
// three fp64 FMAs per voxel - the plain moment kernel's access pattern and speed.  A spaxel
// that holds a NaN / Inf is not linear (NaN renormalisation): its 128-column tile is flagged
// and redone by the general fused kernel.  (The reference rounds the smoothed cube to float32
// before summing; this path does not - a 1e-7-level difference, inside the 1e-5 tolerance.)
constexpr int kMaxAlgebraicTaps = 255;            // taps of the algebraic path travel as kernel arguments

struct WmArgs {
    const float* cube;
    int64_t nz, ny, nx, row_stride, plane_stride;
    const double* w;                  // [nz][3]
    double dv, m1_add;
    spc_moment_outputs mo;
    int64_t mo_row_stride;
    unsigned char* status;            // one byte per 128 columns (linear spaxel index >> 7)
};

typedef float f32x4w __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void weighted_moments_kernel(const WmArgs A) {
    constexpr int ZW = 4;
    __shared__ double sh[ZW - 1][4][4][64];            // wave, {s0, s1, s2, chk}, column, lane
    // (the wave index through readfirstlane: a plane's address offset and its three weights are then provably uniform -
    //  scalar loads instead of three per-lane loads per plane, round 4)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t x = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const int64_t y = blockIdx.y;
    const bool live = x < A.nx;
    const float* p = A.cube + y * A.row_stride + (live ? x : 0);
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    float chk[4] = {0.f, 0.f, 0.f, 0.f};
    auto add = [&](const f32x4w& v, int64_t k) {
        const double w0 = A.w[3 * k], w1 = A.w[3 * k + 1], w2 = A.w[3 * k + 2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double d = (double)v[c];
            s0[c] = fma(d, w0, s0[c]);
            s1[c] = fma(d, w1, s1[c]);
            s2[c] = fma(d, w2, s2[c]);
            chk[c] = fmaf(v[c], 0.f, chk[c]);               // NaN iff a NaN / Inf was read
        }
    };
    int64_t k0 = w;
    for (; k0 + (int64_t)(U - 1) * ZW < A.nz; k0 += ZW * U) {   // U planes (stride ZW) in flight per lane
        f32x4w v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4w*>(p + (k0 + (int64_t)u * ZW) * A.plane_stride));
#pragma unroll
        for (int u = 0; u < U; ++u) add(v[u], k0 + (int64_t)u * ZW);
    }
    for (; k0 < A.nz; k0 += ZW) add(*reinterpret_cast<const f32x4w*>(p + k0 * A.plane_stride), k0);
    if (w > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            sh[w - 1][0][c][lane] = s0[c]; sh[w - 1][1][c][lane] = s1[c];
            sh[w - 1][2][c][lane] = s2[c]; sh[w - 1][3][c][lane] = (double)chk[c];
        }
    }
    __syncthreads();
    if (w != 0) return;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int k = 0; k < ZW - 1; ++k) {
            s0[c] += sh[k][0][c][lane]; s1[c] += sh[k][1][c][lane]; s2[c] += sh[k][2][c][lane];
            chk[c] += (float)sh[k][3][c][lane];
        }
        bad = bad || !(chk[c] == chk[c]);
    }
    // 32 lanes = 128 columns of this ROW: if ANY of them is bad, all of them leave their spaxels to the general kernel.
    // The tile flags are indexed by the LINEAR spaxel number (128 spaxels per flag), so a half-wave of a row whose
    // length is no multiple of 128 can lie in two flag tiles: every lane that quits flags the tile of ITS spaxels (only
    // the bad lane did: the rest of a half-wave in the neighbouring, clean flag tile was then written by nobody).
    const unsigned long long bm = __ballot(bad && live);
    const bool tile_bad = ((lane < 32) ? (bm & 0xffffffffull) : (bm >> 32)) != 0;
    if (!live) return;
    if (tile_bad) { spc_flag_set(A.status + ((y * A.nx + x) >> 7)); return; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t o = y * A.mo_row_stride + x + c;
        const double mu = s1[c] / s0[c];
        if (A.mo.d_m0) A.mo.d_m0[o] = A.dv * s0[c];
        if (A.mo.d_m1) A.mo.d_m1[o] = mu + A.m1_add;
        if (A.mo.d_m2) A.mo.d_m2[o] = s2[c] / s0[c] - mu * mu;
        if (A.mo.d_mu) A.mo.d_mu[o] = mu;
        if (A.mo.d_s0) A.mo.d_s0[o] = s0[c];
        if (A.mo.d_nvalid) A.mo.d_nvalid[o] = (int32_t)A.nz;
    }
}

// per-channel weights of the algebraic path, on the device: W_n(i) = sum_o k[o + H - i] c_o^n / sum(k)
// over the outputs o = i - H + j that exist.  One thread per channel; the taps ride in as kernel arguments.
struct WeightTaps { double k[kMaxAlgebraicTaps]; };

__global__ __launch_bounds__(256) void moment_weights_kernel(const double* cen, int64_t nz, const WeightTaps T, int ntaps,
                                                             double ksum, double* w) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nz) return;
    const int H = ntaps / 2;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int j = 0; j < ntaps; ++j) {
        const int64_t o = i - H + j;                   // the output that reads input i through tap j
        if (o < 0 || o >= nz) continue;
        const double kj = T.k[j], c = cen[o];
        a0 += kj; a1 = fma(kj, c, a1); a2 = fma(kj * c, c, a2);
    }
    w[3 * i] = a0 / ksum; w[3 * i + 1] = a1 / ksum; w[3 * i + 2] = a2 / ksum;
}

// returns 1 when the algebraic pass ran (A.status then marks the tiles left to the general kernel);
// d_status (one byte per 128 columns) and d_w (3 nz doubles) come out of the caller's workspace
int try_weighted_moments(ConvArgs& A, const spc_cube_f32* cube, const double* h_kernel, int ntaps,
                         hipStream_t st, unsigned char* d_status, double* d_w) {
    const char* env = getenv("SPC_FUSE_ALGEBRAIC");
    if (env && atoi(env) == 0) return 0;
    const bool ext = A.mo.d_argmax || A.mo.d_argmin || A.mo.d_vmax || A.mo.d_vmin;
    const bool al = (cube->nx % 4 == 0) && (cube->row_stride % 4 == 0) && (cube->plane_stride % 4 == 0) &&
                    ((((uintptr_t)cube->d_data) & 15) == 0);
    if (ext || !al || (A.mask.flags & ~(uint32_t)SPC_MASK_FINITE) != 0 || cube->ny > 65535 || ntaps > kMaxAlgebraicTaps) return 0;
    const int64_t nz = cube->nz;
    WeightTaps T{};
    double ksum = 0.0;
    for (int j = 0; j < ntaps; ++j) { T.k[j] = h_kernel[j]; ksum += h_kernel[j]; }
    const size_t ntiles = (size_t)((A.ny * A.nx + 127) / 128);
    (void)spc_flags_clear(d_status, ntiles, st);
    hipLaunchKernelGGL(moment_weights_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, st, A.cen, nz, T, ntaps, ksum, d_w);
    WmArgs W{};
    W.cube = A.cube; W.nz = A.nz; W.ny = A.ny; W.nx = A.nx; W.row_stride = A.row_stride; W.plane_stride = A.plane_stride;
    W.w = d_w; W.dv = A.dv; W.m1_add = A.m1_add; W.mo = A.mo; W.mo_row_stride = A.mo_row_stride; W.status = d_status;
    static const int wm_u = [] { const char* e = getenv("SPC_WMOM_U"); return e ? atoi(e) : 8; }();      // tuning hook
    if (wm_u == 4) hipLaunchKernelGGL(weighted_moments_kernel<4>, dim3((unsigned)((A.nx + 255) / 256), (unsigned)A.ny), dim3(256), 0, st, W);
    else hipLaunchKernelGGL(weighted_moments_kernel<8>, dim3((unsigned)((A.nx + 255) / 256), (unsigned)A.ny), dim3(256), 0, st, W);
    A.status = d_status;
    return 1;
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

// no mask array and a single z slice: speculate that the data has no invalid samples
// (all-valid fast kernel, vec spaxels per lane), then redo only the dirty tiles.  d_status: one byte
// per 128 columns from the caller's workspace.
int launch_ring(int R, ConvArgs& A, hipStream_t st, int vec, bool fuse, unsigned char* d_status) {
    const char* env = getenv("SPC_CONV_FAST");
    const bool want = env ? atoi(env) != 0 : true;
    const bool fast = want && (A.mask.flags & ~(uint32_t)SPC_MASK_FINITE) == 0 && A.zchunk >= A.nz;
    A.status = nullptr;
    if (!fast) return launch_ring_raw(R, A, st, 0, fuse);
    const size_t ntiles = (size_t)((A.ny * A.nx + 127) / 128);
    SPC_HIP(spc_flags_clear(d_status, ntiles, st));
    A.status = d_status;
    return launch_ring_raw(R, A, st, vec, fuse);
}

// two spaxels per lane need 8-byte aligned rows everywhere (SPC_CONV_VEC=1 forces one)
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
    spc_canonical_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, &A.pred_lim, &A.pred_lo, &A.pred_hi);
    A.cube = cube->d_data;
    A.nz = cube->nz; A.ny = cube->ny; A.nx = cube->nx;
    A.row_stride = cube->row_stride; A.plane_stride = cube->plane_stride;
    for (int i = 0; i < 72; ++i) A.k[i] = 0.0;
    if (R) {
        const int pad = (R - ntaps) / 2;
        for (int i = 0; i < ntaps; ++i) A.k[pad + i] = h_kernel[i];
        { double ks = 0.0; for (int i = 0; i < R; ++i) ks += A.k[i]; A.ksum = ks; A.inv_ksum = 1.0 / ks; }
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
                          int64_t out_plane_stride, void* d_workspace, size_t workspace_bytes) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    rc = check_kernel(h_kernel, ntaps);
    if (rc) return rc;
    SPC_REQUIRE(d_out != nullptr, "d_out is NULL");
    int R = pick_ring(h_kernel, ntaps);
    {
        const int64_t ors = out_row_stride ? out_row_stride : cube->nx;
        if (R && !ring_fits(R, cube, mask, ors, out_plane_stride ? out_plane_stride : cube->ny * ors)) R = 0;
    }
    ConvArgs A{};
    rc = fill_common(A, cube, mask, h_kernel, ntaps, R);
    if (rc) return rc;
    SPC_DEVICE(device);
    SpcWorkspace ws(d_workspace, workspace_bytes);
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
    { const char* sd = getenv("SPC_SPECTRAL_SKIP_DEAD"); A.skip_dead = sd ? atoi(sd) : 1; }
    nsplit = (int)((cube->nz + A.zchunk - 1) / A.zchunk);
    dim3 grid((unsigned)nblocks, (unsigned)nsplit);
    hipStream_t st = (hipStream_t)stream;
    if (R) {
        SPC_WS_TAKE(d_status, ws, unsigned char, (ncols + 127) / 128);
        const int vec = pick_vec(cube, A.mask, d_out, A.out_row_stride, A.out_plane_stride);
        return launch_ring(R, A, st, vec, false, d_status);
    }
    // no ring for this kernel: runs-of-16 kernel for 17 taps or more, per-output tap loop below that
    // (taps in the workspace, written there by kernel-argument uploads: nothing waits)
    const char* wenv = getenv("SPC_CONV_WIDE");
    const bool wide = (wenv ? atoi(wenv) != 0 : true) && ntaps >= 17;
    const int npad = wide ? ntaps + 30 : ntaps;
    std::vector<double> hk((size_t)npad, 0.0);
    for (int i = 0; i < ntaps; ++i) hk[(wide ? 15 : 0) + i] = h_kernel[i];
    SPC_WS_TAKE(d_k, ws, double, ntaps + 30);
    SPC_HIP(spc_table_upload(d_k, hk.data(), sizeof(double) * npad, st));
    A.status = nullptr;
    // 35 - 65 symmetric taps on data the mask of which only rejects non-finite samples: the all-valid ring pass first
    // (spc_spectral_conv_r49/65), then the kernels below redo the tiles that hold an invalid sample
    {
        const int Rf = ntaps <= 49 ? 49 : 65;
        bool sym = (ntaps & 1) && ntaps <= 65 && h_kernel[ntaps / 2] != 0.0;
        for (int i = 0; sym && i < ntaps / 2; ++i) sym = h_kernel[i] == h_kernel[ntaps - 1 - i];
        const bool fits = wide && sym && ring_fits(Rf, cube, mask, A.out_row_stride, A.out_plane_stride);
        const char* fenv = getenv("SPC_CONV_FAST");
        if (fits && (fenv ? atoi(fenv) != 0 : true) && (A.mask.flags & ~(uint32_t)SPC_MASK_FINITE) == 0) {
            rc = fill_common(A, cube, mask, h_kernel, ntaps, Rf);            // taps centred in the ring, kernel sum
            if (rc) return rc;
            SPC_WS_TAKE(d_status, ws, unsigned char, (ncols + 127) / 128);
            SPC_HIP(spc_flags_clear(d_status, (size_t)((ncols + 127) / 128), st));
            A.status = d_status;
            const int64_t keep = A.zchunk;
            A.zchunk = cube->nz;                                             // the ring pass marches over the whole ray
            rc = Rf == 49 ? spc_sconv::launch_fast_only<49>(A, st) : spc_sconv::launch_fast_only<65>(A, st);
            if (rc) return rc;
            A.zchunk = keep;
        }
        // tiles with invalid samples / cubes that bring a mask: the general form of the same rings (round 5), for
        // non-negative taps (an empty window then means an invalid centre sample: NaN, which is what 0 / 0 gives)
        bool nonneg = true;
        for (int i = 0; i < ntaps; ++i) nonneg = nonneg && h_kernel[i] >= 0.0;
        const char* renv = getenv("SPC_SPECTRAL_RING_WIDE");
        if (fits && nonneg && (renv ? atoi(renv) != 0 : true)) {
            unsigned char* keep_status = A.status;
            rc = fill_common(A, cube, mask, h_kernel, ntaps, Rf);
            if (rc) return rc;
            A.status = keep_status;
            return Rf == 49 ? spc_sconv::launch_ring_wide<49>(A, st) : spc_sconv::launch_ring_wide<65>(A, st);
        }
    }
    if (wide) {
        // whole runs of 16 per z slice
        A.zchunk = ((A.zchunk + 15) / 16) * 16;
        dim3 wgrid((unsigned)nblocks, (unsigned)((cube->nz + A.zchunk - 1) / A.zchunk));
        // no ring pass ran (asymmetric kernel or more than 65 taps) and the mask only rejects non-finite samples: the
        // numerator-only form first
        {
            const char* fenv = getenv("SPC_CONV_FAST");
            if (!A.status && (fenv ? atoi(fenv) != 0 : true) && (A.mask.flags & ~(uint32_t)SPC_MASK_FINITE) == 0) {
                SPC_WS_TAKE(d_status, ws, unsigned char, (ncols + 127) / 128);
                SPC_HIP(spc_flags_clear(d_status, (size_t)((ncols + 127) / 128), st));
                A.status = d_status;
                double ks = 0.0;
                for (int i = 0; i < ntaps; ++i) ks += h_kernel[i];
                A.ksum = ks;
                hipLaunchKernelGGL((spectral_conv_wide_kernel<false, true>), wgrid, dim3(256), 0, st, A, d_k, ntaps);
                SPC_LAUNCH_CHECK();
            }
        }
        if (A.mask.flags & SPC_MASK_ARRAY) hipLaunchKernelGGL((spectral_conv_wide_kernel<true, false>), wgrid, dim3(256), 0, st, A, d_k, ntaps);
        else hipLaunchKernelGGL((spectral_conv_wide_kernel<false, false>), wgrid, dim3(256), 0, st, A, d_k, ntaps);
    } else {
        hipLaunchKernelGGL(spectral_conv_generic_kernel, grid, dim3(256), 0, st, A, d_k, ntaps);
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_spectral_conv_moments_f32(int device, void* stream, const spc_cube_f32* cube,
                                  const spc_mask* mask, const double* h_kernel, int ntaps,
                                  const double* d_cen, const double* h_cen, double dv, double m1_add,
                                  const spc_moment_outputs* out, void* d_workspace, size_t workspace_bytes) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    rc = check_kernel(h_kernel, ntaps);
    if (rc) return rc;
    SPC_REQUIRE(out != nullptr && d_cen != nullptr, "NULL pointer argument");
    int R = pick_ring(h_kernel, ntaps);
    if (R && !ring_fits(R, cube, mask, 0, 0)) R = 0;
    ConvArgs A{};
    rc = fill_common(A, cube, mask, h_kernel, ntaps, R);
    if (rc) return rc;
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    SpcWorkspace ws(d_workspace, workspace_bytes);
    const size_t ntiles = (size_t)((cube->ny * cube->nx + 127) / 128);
    SPC_WS_TAKE(d_status, ws, unsigned char, ntiles);
    SPC_WS_TAKE(d_w, ws, double, 3 * cube->nz);
    A.cen = d_cen; A.dv = dv; A.m1_add = m1_add; A.mo = *out;
    A.mo_row_stride = out->out_row_stride ? out->out_row_stride : cube->nx;
    if (!R) {
        // no ring (wide kernel): only the algebraic all-valid path can fuse; a flagged tile or an
        // extremum request sends the caller to the materialised route.  Whether a tile was flagged is a
        // result the HOST needs before it can answer: this branch waits for the stream.
        bool ok = false;
        if (try_weighted_moments(A, cube, h_kernel, ntaps, st, d_status, d_w)) {
            std::vector<unsigned char> hs(ntiles);
            ok = hipMemcpyAsync(hs.data(), d_status, ntiles, hipMemcpyDeviceToHost, st) == hipSuccess &&
                 hipStreamSynchronize(st) == hipSuccess;
            for (size_t i = 0; ok && i < ntiles; ++i) ok = hs[i] == 0;
        }
        if (ok) return SPC_OK;
        spc_set_error("fused smooth->moments with %d taps needs an all-valid cube and no extremum outputs "
                      "(ring kernels go up to %d taps); materialise with spc_spectral_conv_f32 instead", ntaps, 33);
        return SPC_ERR_UNSUPPORTED;
    }
    // linear spectral axis (every FITS axis is): c[z] = c0 + z*dc -> no per-channel loads
    A.cen_linear = 0;
    if (h_cen && cube->nz >= 2) {
        const double c0 = h_cen[0], dc = (h_cen[cube->nz - 1] - h_cen[0]) / (double)(cube->nz - 1);
        double span = std::abs(dc) * (double)cube->nz, worst = 0.0;
        for (int64_t z = 0; z < cube->nz; ++z) worst = std::max(worst, std::abs(h_cen[z] - (c0 + dc * (double)z)));
        if (worst <= 1e-13 * std::max(span, 1e-300)) { A.cen_linear = 1; A.cen_c0 = c0; A.cen_dc = dc; }
    }
    A.zchunk = cube->nz;
    { const char* sd = getenv("SPC_SPECTRAL_SKIP_DEAD"); A.skip_dead = sd ? atoi(sd) : 1; }
    const int vec = pick_vec(cube, A.mask, nullptr, 0, 0);
    if (try_weighted_moments(A, cube, h_kernel, ntaps, st, d_status, d_w)) {
        SPC_LAUNCH_CHECK();
        return launch_ring_raw(R, A, st, 0, true);      // general kernel: flagged tiles only
    }
    return launch_ring(R, A, st, vec, true, d_status);
}

}  // extern "C"

size_t spc_ws_spectral_conv(int64_t nz, int64_t ny, int64_t nx, int64_t ntaps, bool fused) {
    size_t n = spc_ws_round((size_t)((ny * nx + 127) / 128));              // tile flags
    n += fused ? spc_ws_round(sizeof(double) * 3 * (size_t)nz)            // per-channel weights of the algebraic path
               : spc_ws_round(sizeof(double) * (size_t)(ntaps + 30));     // taps of the kernels without a ring
    return n + 256;
}
