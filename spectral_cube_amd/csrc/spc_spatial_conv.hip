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

#include "spc_spatial_conv_impl.h"
#include <cstdlib>
#include <vector>

namespace spc_spconv {
extern template int launch_sep<9>(const SpArgs&, hipStream_t, dim3, bool);
extern template int launch_sep<17>(const SpArgs&, hipStream_t, dim3, bool);
extern template int launch_sep<29>(const SpArgs&, hipStream_t, dim3, bool);
extern template int launch_sep<33>(const SpArgs&, hipStream_t, dim3, bool);
extern template int launch_sep<49>(const SpArgs&, hipStream_t, dim3, bool);
extern template int launch_sep<65>(const SpArgs&, hipStream_t, dim3, bool);
}
using namespace spc_spconv;

namespace {

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

// mask predicate -> |v| <= lim && !(v <= lo) && !(v >= hi); an absent bound is NaN (the negated
// compare is then true for every v), >= / <= become strict compares against the neighbouring
// float, a NaN threshold rejects everything (numpy: x > nan is False)
void canonical_pred(SpArgs& A) {
    const uint32_t f = A.mask.flags;
    A.pred_lim = (f & SPC_MASK_FINITE) ? 3.402823466e+38f : INFINITY;
    A.pred_lo = NAN;
    A.pred_hi = NAN;
    if (f & (SPC_MASK_GT | SPC_MASK_GE)) {
        const float t = A.mask.thr_lo;
        if (t != t) A.pred_lim = -1.f;
        else if (f & SPC_MASK_GT) A.pred_lo = t;
        else A.pred_lo = (t == -INFINITY) ? NAN : nextafterf(t, -INFINITY);
    }
    if (f & (SPC_MASK_LT | SPC_MASK_LE)) {
        const float t = A.mask.thr_hi;
        if (t != t) A.pred_lim = -1.f;
        else if (f & SPC_MASK_LT) A.pred_hi = t;
        else A.pred_hi = (t == INFINITY) ? NAN : nextafterf(t, INFINITY);
    }
}

// ---- non-separable kernels up to 33 x 33: LDS-tiled direct convolution --------------------------
// (Tophat2DKernel, rotated elliptical Gaussians = what convolve_to builds.)  A block owns a
// 64 x 32 output tile of one plane: the input tile with its halo is classified once into LDS as
// packed (value * valid, valid) pairs, a lane then produces a vertical run of 8 outputs, so each
// LDS read feeds up to 8 packed FMAs (num and den together); weights are wave-uniform scalar
// loads.  The run's first and last 7 input rows reach only some of the 8 outputs and are
// predicated (a zero-padded weight table would turn an Inf sample into NaN for outputs whose
// window does not contain it).  Measured (15 x 15, 1024^3): 184 ms (per-pixel global loads) -> see
// DESIGN.md 3.3.
constexpr int kT2X = 64, kT2Y = 32, kT2Run = 8, kT2MaxK = 65;       // 65 x 65: 99 KB of LDS, one block per CU

template <bool ARR>
__global__ __launch_bounds__(256) void spatial_conv2d_tiled_kernel(const SpArgs A, const float* kern, int nky, int nkx) {
    extern __shared__ float2v tile[];                     // (kT2Y + nky - 1) x pitch
    const int hy = nky / 2, hx = nkx / 2;
    const int trows = kT2Y + nky - 1, tcols = kT2X + nkx - 1;
    const int pitch = tcols | 1;                           // odd pitch in float2 units
    const int64_t z = blockIdx.z;
    const int64_t X0 = (int64_t)blockIdx.x * kT2X, Y0 = (int64_t)blockIdx.y * kT2Y;
    // tiles already finished by the all-valid pass (its tiles are two of these wide)
    if (A.status && spc_flag_get(A.status + (z * gridDim.y + blockIdx.y) * A.fast_nstrips + (blockIdx.x >> 1)) == 0) return;
    const float* p = A.cube + z * A.plane_stride;
    const uint8_t* pm = ARR ? A.mask.arr + z * A.mask.plane_stride : nullptr;
    // true convolution: out[y][x] = sum k[jy][jx] in[y + hy - jy][x + hx - jx]; tile row r holds input
    // row Y0 - hy + r, tile column c holds input column X0 - hx + c
    for (int e = threadIdx.x; e < trows * tcols; e += 256) {
        const int r = e / tcols, c = e - r * tcols;
        const int64_t iy = Y0 - hy + r, ix = X0 - hx + c;
        float2v val = float2v{0.f, 1.f};                   // outside the image: a valid zero
        if (iy >= 0 && iy < A.ny && ix >= 0 && ix < A.nx) {
            const float v = p[iy * A.row_stride + ix];
            bool ok = spc_pred_valid(A.mask, v);
            if (ARR) ok = ok && pm[iy * A.mask.row_stride + ix] != 0;
            val = ok ? float2v{v, 1.f} : float2v{0.f, 0.f};
        }
        tile[r * pitch + c] = val;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = (threadIdx.x >> 6) * kT2Run;
    float2v acc[kT2Run];
#pragma unroll
    for (int o = 0; o < kT2Run; ++o) acc[o] = float2v{0.f, 0.f};
    // Output (ty + o, tx) with tap (jy, jx) reads tile[ty + r][tx + nkx - 1 - jx], r = o + nky - 1 - jy in
    // [0, nky + 7).  Input row r is walked in chunks of 8: one LDS read per row, and the 15 weights
    // a chunk can touch (kernT[jx][nky - 1 - r0 + 0..14], 7 zeros of padding on both sides, never
    // multiplied: masks below) come in as wave-uniform scalar loads, so every FMA takes its weight
    // from an SGPR.  Which (row, output) pairs exist is STATIC inside the first chunk (o <= i) and
    // the last one (i <= o); in between every pair exists.
    const int wpitch = nky + 14;
    auto chunk = [&](const float2v* col, const float* wp, int r0, int mode, int nrows) {
        float w[15];
#pragma unroll
        for (int q = 0; q < 15; ++q) w[q] = wp[q];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (mode == 1 && i >= nrows) break;              // partial middle chunk (wave-uniform)
            const float2v in = col[(r0 + i) * pitch];
#pragma unroll
            for (int o = 0; o < kT2Run; ++o) {
                const bool on = (mode == 0) ? (o <= i) : (mode == 2) ? (i <= o) : true;
                if (on) acc[o] = __builtin_elementwise_fma(float2v{w[7 - i + o], w[7 - i + o]}, in, acc[o]);
            }
        }
    };
    for (int jx = 0; jx < nkx; ++jx) {
        const float2v* col = tile + ty * pitch + tx + (nkx - 1 - jx);
        const float* wcol = kern + (size_t)jx * wpitch;     // padded column jx: wcol[7 + jy] = k[jy][jx]
        // weight of (row r, output o) = k[nky - 1 - r + o][jx] = wcol[nky + 6 - r + o]; chunk at r0, row i:
        // wcol[(nky - 1 - r0) + 7 - i + o]
        chunk(col, wcol + (nky - 1), 0, 0, 8);                              // rows 0..7
        int r0 = 8;
        for (; r0 + 8 <= nky - 1; r0 += 8) chunk(col, wcol + (nky - 1 - r0), r0, 1, 8);
        if (r0 < nky - 1) chunk(col, wcol + (nky - 1 - r0), r0, 1, nky - 1 - r0);   // rows r0 .. nky-2
        chunk(col, wcol + 0, nky - 1, 2, 8);                                // rows nky-1 .. nky+6
    }
    const int64_t x = X0 + tx;
    if (x >= A.nx) return;
#pragma unroll
    for (int o = 0; o < kT2Run; ++o) {
        const int64_t y = Y0 + ty + o;
        if (y >= A.ny) break;
        float res;
        if (acc[o].y != 0.f) res = acc[o].x / acc[o].y;
        else {                                                          // empty window -> filled centre sample
            const float c = p[y * A.row_stride + x];
            bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, c);
            if (ARR) inc = inc && pm[y * A.mask.row_stride + x] != 0;
            res = inc ? c : NAN;
        }
        A.out[z * A.out_plane_stride + y * A.out_row_stride + x] = res;
    }
}

// ---- all-valid pass of the tiled 2-D stencil ------------------------------------------------------------
// The kernel above carries (value * valid, valid) pairs: with every sample valid half of each packed FMA is spent on a
// denominator that is the kernel sum.  This pass owns a 128 x 32 output tile and packs TWO COLUMNS 64 apart into the
// pair instead - (in[r][c], in[r][c + 64]) share every weight - so the same instruction stream produces twice the
// outputs, out = num * (1 / sum k).  A block that meets an excluded sample in its tile or halo flags the tile and
// quits; the kernel above then redoes the flagged tiles (speculation as in the separable stencils).
template <bool SCAL>
__global__ __launch_bounds__(256) void spatial_conv2d_allvalid_kernel(const SpArgs A, const float* kern, int nky, int nkx) {
    extern __shared__ float2v tile[];                     // (kT2Y + nky - 1) x pitch pairs (column c, column c + 64)
    const int hy = nky / 2, hx = nkx / 2;
    const int trows = kT2Y + nky - 1, tcols = kT2X + nkx - 1;
    const int pitch = tcols | 1;
    const int64_t z = blockIdx.z;
    const int64_t X0 = (int64_t)blockIdx.x * (2 * kT2X), Y0 = (int64_t)blockIdx.y * kT2Y;
    const float* p = A.cube + z * A.plane_stride;
    bool bad = false;
    for (int e = threadIdx.x; e < trows * tcols; e += 256) {
        const int r = e / tcols, c = e - r * tcols;
        const int64_t iy = Y0 - hy + r, ix = X0 - hx + c;
        float2v val = float2v{0.f, 0.f};                   // outside the image: a valid zero
        if (iy >= 0 && iy < A.ny) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t ixh = ix + h * kT2X;
                if (ixh >= 0 && ixh < A.nx) {
                    const float v = p[iy * A.row_stride + ixh];
                    bad = bad || !spc_pred_valid(A.mask, v);
                    if (h == 0) val.x = v; else val.y = v;
                }
            }
        }
        tile[r * pitch + c] = val;
    }
    if (__syncthreads_or(bad ? 1 : 0)) {
        if (threadIdx.x == 0) spc_flag_set(A.status + (z * gridDim.y + blockIdx.y) * A.fast_nstrips + blockIdx.x);
        return;
    }
    const int tx = threadIdx.x & 63, ty = (threadIdx.x >> 6) * kT2Run;
    float2v acc[kT2Run];
#pragma unroll
    for (int o = 0; o < kT2Run; ++o) acc[o] = float2v{0.f, 0.f};
    const int wpitch = nky + 14;                            // (the weight table and the chunk walk of the kernel above)
    auto chunk = [&](const float2v* col, const float* wp, int r0, int mode, int nrows) {
        // SCAL (weight table <= 4 KB): read through the constant address space - scalar loads that stay in the scalar
        // cache (13 x 13 taps: 9.1 -> 6.7 ms); larger tables thrash it (41 x 41: 49 -> 59 ms) and use plain loads
        typedef const float __attribute__((address_space(4))) cfloat;
        float w[15];
#pragma unroll
        for (int q = 0; q < 15; ++q) w[q] = SCAL ? ((cfloat*)wp)[q] : wp[q];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (mode == 1 && i >= nrows) break;
            const float2v in = col[(r0 + i) * pitch];
#pragma unroll
            for (int o = 0; o < kT2Run; ++o) {
                const bool on = (mode == 0) ? (o <= i) : (mode == 2) ? (i <= o) : true;
                if (on) acc[o] = __builtin_elementwise_fma(float2v{w[7 - i + o], w[7 - i + o]}, in, acc[o]);
            }
        }
    };
    for (int jx = 0; jx < nkx; ++jx) {
        const float2v* col = tile + ty * pitch + tx + (nkx - 1 - jx);
        const float* wcol = kern + (size_t)jx * wpitch;
        chunk(col, wcol + (nky - 1), 0, 0, 8);
        int r0 = 8;
        for (; r0 + 8 <= nky - 1; r0 += 8) chunk(col, wcol + (nky - 1 - r0), r0, 1, 8);
        if (r0 < nky - 1) chunk(col, wcol + (nky - 1 - r0), r0, 1, nky - 1 - r0);
        chunk(col, wcol + 0, nky - 1, 2, 8);
    }
    const int64_t x = X0 + tx;
#pragma unroll
    for (int o = 0; o < kT2Run; ++o) {
        const int64_t y = Y0 + ty + o;
        if (y >= A.ny) break;
        float* q = A.out + z * A.out_plane_stride + y * A.out_row_stride;
        if (x < A.nx) q[x] = acc[o].x * A.inv_ksum;
        if (x + kT2X < A.nx) q[x + kT2X] = acc[o].y * A.inv_ksum;
    }
}

// ---- very wide separable kernels (more taps than the largest ring, e.g. Gaussian2DKernel(sigma > 8)) ----
// Two passes through an intermediate (num, den) buffer, a chunk of planes at a time:
//   y pass: the runs-of-16 scheme of the wide spectral kernel with y as the marched axis - a lane owns
//           one (z, x) column and produces 16 consecutive rows per run, weights as wave-uniform scalars;
//   x pass: a block stages one row (+ halo) of the intermediate in LDS, a lane produces one output with
//           one LDS read per tap (consecutive lanes read consecutive words: conflict free); columns
//           outside the image are valid zeros of EVERY row: (num, den) = (0, sum(ky)).
// Was: outer product + per-pixel 2-D loop (81 x 81 taps: 6561 global loads per output).
template <bool ARR>
__global__ __launch_bounds__(256) void spatial_wide_ypass_kernel(const SpArgs A, const float* kpad, int ntaps, float2v* inter,
                                                                   int64_t z0, int64_t nzc) {
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (z - z0) * nx + x
    if (col >= nzc * A.nx) return;
    const int64_t zl = col / A.nx, x = col - zl * A.nx, z = z0 + zl;
    const int H = ntaps / 2;
    const float* p = A.cube + z * A.plane_stride + x;
    const uint8_t* pm = ARR ? A.mask.arr + z * A.mask.plane_stride + x : nullptr;
    float2v* q = inter + zl * A.ny * A.nx + x;
    const int64_t yb = (int64_t)blockIdx.y * A.ychunk;
    const int64_t ye = min(A.ny, yb + A.ychunk);
    for (int64_t o0 = yb; o0 < ye; o0 += 16) {
        float2v acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = float2v{0.f, 0.f};
        auto sample = [&](int r) -> float2v {
            const int64_t i = o0 - H + r;
            if (i < 0 || i >= A.ny) return float2v{0.f, 1.f};
            const float v = p[i * A.row_stride];
            bool ok = (__builtin_fabsf(v) <= A.pred_lim) && !(v <= A.pred_lo) && !(v >= A.pred_hi);
            if (ARR) ok = ok && pm[i * A.mask.row_stride] != 0;
            return ok ? float2v{v, 1.f} : float2v{0.f, 0.f};
        };
        auto chunk = [&](int r0, int mode, int off, int nrows) {
            const float* wp = kpad + 15 + (ntaps - 1 - r0) - 7;
            float w[23];
#pragma unroll
            for (int t = 0; t < 23; ++t) w[t] = wp[t];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (mode == 1 && i >= nrows) break;
                const float2v in = sample(r0 + i);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const bool on = (mode == 0) ? (k <= i + off) : (mode == 2) ? (k >= i + off) : true;
                    if (on) acc[k] = __builtin_elementwise_fma(float2v{w[7 - i + k], w[7 - i + k]}, in, acc[k]);
                }
            }
        };
        chunk(0, 0, 0, 8);
        chunk(8, 0, 8, 8);
        int r0 = 16;
        for (; r0 + 8 <= ntaps - 1; r0 += 8) chunk(r0, 1, 0, 8);
        if (r0 < ntaps - 1) chunk(r0, 1, 0, ntaps - 1 - r0);
        chunk(ntaps - 1, 2, 0, 8);
        chunk(ntaps + 7, 2, 8, 8);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (o0 + k < ye) q[(o0 + k) * A.nx] = acc[k];
    }
}

template <bool ARR>
__global__ __launch_bounds__(256) void spatial_wide_xpass_kernel(const SpArgs A, const float* kx, int ntaps, const float2v* inter,
                                                                   int64_t z0, float sum_ky) {
    extern __shared__ float2v rowbuf[];                    // 256 + ntaps - 1
    const int H = ntaps / 2;
    const int64_t zl = blockIdx.z, y = blockIdx.y, z = z0 + zl;
    const int64_t X0 = (int64_t)blockIdx.x * 256;
    const float2v* src = inter + (zl * A.ny + y) * A.nx;
    for (int c = threadIdx.x; c < 256 + ntaps - 1; c += 256) {
        const int64_t ix = X0 - H + c;
        rowbuf[c] = (ix >= 0 && ix < A.nx) ? src[ix] : float2v{0.f, sum_ky};
    }
    __syncthreads();
    const int64_t x = X0 + threadIdx.x;
    if (x >= A.nx) return;
    float2v acc = float2v{0.f, 0.f};
    // out[x] = sum_j kx[j] in[x + H - j]: buffer index threadIdx.x + (ntaps - 1 - j)
    for (int j = 0; j < ntaps; ++j) {
        const float w = kx[j];
        acc = __builtin_elementwise_fma(float2v{w, w}, rowbuf[threadIdx.x + (ntaps - 1 - j)], acc);
    }
    float res;
    if (acc.y != 0.f) res = acc.x / acc.y;
    else {
        const float c = A.cube[z * A.plane_stride + y * A.row_stride + x];
        bool inc = spc_pred(A.mask.flags, A.mask.thr_lo, A.mask.thr_hi, c);
        if (ARR) inc = inc && A.mask.arr[z * A.mask.plane_stride + y * A.mask.row_stride + x] != 0;
        res = inc ? c : NAN;
    }
    A.out[z * A.out_plane_stride + y * A.out_row_stride + x] = res;
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

// planes per chunk of the two-pass wide path: <= 256 MiB of (num, den) per chunk
int64_t wide_chunk_planes(int64_t nz, int64_t plane) {
    return std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nz, 65535), (int64_t)(1ll << 28) / (plane * 8)));
}

}  // namespace

size_t spc_ws_spatial_conv_sep(int64_t nz, int64_t ny, int64_t nx, int64_t nky, int64_t nkx) {
    const int R = pick_ring((int)std::max(nky, nkx));
    if (R) {                                                    // tile flags of the speculative pass: strips x channels
        const int64_t nstrips = (nx + fast_txo(65) - 1) / fast_txo(65);      // the narrowest strips of any ring
        return spc_ws_round((size_t)(nstrips * nz)) + 256;
    }
    const int64_t nyp = std::max<int64_t>(nky, 17);
    return spc_ws_round(sizeof(float) * (size_t)(nyp + 30 + nkx)) +
           spc_ws_round(sizeof(float2v) * (size_t)(wide_chunk_planes(nz, ny * nx) * ny * nx)) + 512;
}

size_t spc_ws_spatial_conv2d(int64_t nz, int64_t ny, int64_t nx, int64_t nky, int64_t nkx) {
    const int64_t ntiles = nz * ((ny + kT2Y - 1) / kT2Y) * ((nx + 2 * kT2X - 1) / (2 * kT2X));   // flags of the all-valid pass
    return spc_ws_round(sizeof(float) * (size_t)(nkx * (nky + 14))) + spc_ws_round((size_t)ntiles) + 256;
}

extern "C" {

int spc_spatial_conv2d_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                           const double* h_kernel, int nky, int nkx, float* d_out,
                           int64_t out_row_stride, int64_t out_plane_stride, void* d_workspace, size_t workspace_bytes) {
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
    const char* env = getenv("SPC_CONV2D_TILED");
    // (the tiled kernel's static row/output masks assume at least 9 rows of taps)
    const bool tiled = (env ? atoi(env) != 0 : true) && nky >= 9 && nky <= kT2MaxK && nkx <= kT2MaxK &&
                       (cube->ny + kT2Y - 1) / kT2Y <= 65535;
    // tiled: transposed columns padded with 7 zeros on both sides, kT[jx][7 + jy] = k[jy][jx]
    const int n = tiled ? nkx * (nky + 14) : nky * nkx;
    std::vector<float> hk((size_t)n, 0.f);
    if (tiled) {
        for (int jy = 0; jy < nky; ++jy)
            for (int jx = 0; jx < nkx; ++jx) hk[(size_t)jx * (nky + 14) + 7 + jy] = (float)h_kernel[jy * nkx + jx];
    } else {
        for (int i = 0; i < n; ++i) hk[i] = (float)h_kernel[i];
    }
    SpcWorkspace ws(d_workspace, workspace_bytes);
    SPC_WS_TAKE(d_k, ws, float, (size_t)nkx * (nky + 14));
    SPC_HIP(spc_table_upload(d_k, hk.data(), sizeof(float) * n, st));
    {
        if (tiled) {
            const int pitch = (kT2X + nkx - 1) | 1;
            const size_t lds = (size_t)(kT2Y + nky - 1) * pitch * sizeof(float2v);
            dim3 grid((unsigned)((cube->nx + kT2X - 1) / kT2X), (unsigned)((cube->ny + kT2Y - 1) / kT2Y), (unsigned)cube->nz);
            if (lds > 48 * 1024) {      // dynamic LDS beyond the default limit has to be requested
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_conv2d_tiled_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_conv2d_tiled_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            }
            // masks that only reject non-finite samples: the all-valid pass first, then the flagged tiles
            const char* fenv = getenv("SPC_CONV_FAST");
            A.status = nullptr;
            if ((fenv ? atoi(fenv) != 0 : true) && (A.mask.flags & ~(uint32_t)SPC_MASK_FINITE) == 0) {
                double sumf = 0.0;                      // the denominator the general kernel accumulates: float32 taps
                for (int i = 0; i < nky * nkx; ++i) sumf += (double)(float)h_kernel[i];
                A.inv_ksum = (float)(1.0 / sumf);
                A.fast_nstrips = (int)((cube->nx + 2 * kT2X - 1) / (2 * kT2X));
                const size_t nt = (size_t)cube->nz * grid.y * (size_t)A.fast_nstrips;
                SPC_WS_TAKE(d_status, ws, unsigned char, nt);
                SPC_HIP(spc_flags_clear(d_status, nt, st));
                A.status = d_status;
                const bool scal = (size_t)nkx * (nky + 14) * sizeof(float) <= 4096;
                if (lds > 48 * 1024) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_conv2d_allvalid_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_conv2d_allvalid_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                }
                const dim3 fgrid((unsigned)A.fast_nstrips, grid.y, grid.z);
                if (scal) hipLaunchKernelGGL(spatial_conv2d_allvalid_kernel<true>, fgrid, dim3(256), lds, st, A, d_k, nky, nkx);
                else hipLaunchKernelGGL(spatial_conv2d_allvalid_kernel<false>, fgrid, dim3(256), lds, st, A, d_k, nky, nkx);
                SPC_LAUNCH_CHECK();
            }
            if (A.mask.flags & SPC_MASK_ARRAY) hipLaunchKernelGGL(spatial_conv2d_tiled_kernel<true>, grid, dim3(256), lds, st, A, d_k, nky, nkx);
            else hipLaunchKernelGGL(spatial_conv2d_tiled_kernel<false>, grid, dim3(256), lds, st, A, d_k, nky, nkx);
        } else {
            dim3 grid((unsigned)((cube->nx + 63) / 64), (unsigned)((cube->ny + 3) / 4), (unsigned)cube->nz);
            hipLaunchKernelGGL(spatial_conv2d_kernel, grid, dim3(256), 0, st, A, d_k, nky, nkx);
        }
    }
    SPC_LAUNCH_CHECK();
    return SPC_OK;
}

int spc_spatial_conv_sep_f32(int device, void* stream, const spc_cube_f32* cube, const spc_mask* mask,
                             const double* h_ky, int nky, const double* h_kx, int nkx, float* d_out,
                             int64_t out_row_stride, int64_t out_plane_stride, void* d_workspace, size_t workspace_bytes) {
    int rc = spc_check_cube(cube);
    if (rc) return rc;
    if ((rc = check_kernel(h_ky, nky, "y")) || (rc = check_kernel(h_kx, nkx, "x"))) return rc;
    double sy = 0.0, sx = 0.0;
    for (int i = 0; i < nky; ++i) sy += h_ky[i];
    for (int i = 0; i < nkx; ++i) sx += h_kx[i];
    const double sum = sy * sx;
    SPC_REQUIRE(!(sum < 1e-8 && sum > -1e-8) && sum >= 1e-8,
                "The kernel can't be normalized, because its sum is close to zero");
    // a mask ARRAY (and nothing but the array / isfinite) with a kernel of up to 29 non-negative taps: the split form on the
    // matrix cores first (spc_spatial_split.hip); whatever it does not take stays with the ring kernels below
    if (mask && (mask->flags & SPC_MASK_ARRAY) && !(mask->flags & ~(uint32_t)(SPC_MASK_ARRAY | SPC_MASK_FINITE)) && std::max(nky, nkx) <= 65 && d_out) {
        const char* ring = getenv("SPC_SPATIAL_RING");
        if (!(ring && atoi(ring) == 1)) {
            rc = spc_spatial_conv_split_store(device, stream, cube, mask, h_ky, nky, h_kx, nkx, d_out, out_row_stride, out_plane_stride);
            if (rc != SPC_ERR_UNSUPPORTED) return rc;
        }
    }
    const int R = pick_ring(std::max(nky, nkx));
    if (!R) {
        // very wide kernels: two passes through a (num, den) buffer, a chunk of planes at a time
        SpArgs A{};
        rc = fill_args(A, cube, mask, d_out, out_row_stride, out_plane_stride);
        if (rc) return rc;
        SPC_REQUIRE(cube->ny <= 65535, "more than 65535 rows per call not supported (split the call)");
        SPC_DEVICE(device);
        canonical_pred(A);
        hipStream_t st = (hipStream_t)stream;
        const bool arr = (A.mask.flags & SPC_MASK_ARRAY) != 0;
        // y weights padded by 15 zeros (runs-of-16 scheme needs >= 17 taps: true here, the rings cover less), x plain
        SPC_REQUIRE(nky >= 17 || nkx >= 17, "internal: wide path called with a short kernel");
        const int nyp = std::max(nky, 17);                  // a short y kernel is centred in 17 taps of zeros
        std::vector<float> hk((size_t)(nyp + 30) + nkx, 0.f);
        for (int j = 0; j < nky; ++j) hk[15 + (nyp - nky) / 2 + j] = (float)h_ky[j];
        for (int j = 0; j < nkx; ++j) hk[(size_t)(nyp + 30) + j] = (float)h_kx[j];
        const int64_t plane = cube->ny * cube->nx;
        const int64_t nzc_max = wide_chunk_planes(cube->nz, plane);
        SpcWorkspace ws(d_workspace, workspace_bytes);
        SPC_WS_TAKE(d_k, ws, float, hk.size());
        SPC_WS_TAKE(d_inter, ws, float2v, (size_t)(nzc_max * plane));
        SPC_HIP(spc_table_upload(d_k, hk.data(), sizeof(float) * hk.size(), st));
        A.ychunk = ((std::max<int64_t>(16, (cube->ny + 3) / 4) + 15) / 16) * 16;      // a few y slices of whole runs
        const unsigned nys = (unsigned)((cube->ny + A.ychunk - 1) / A.ychunk);
        for (int64_t z0 = 0; z0 < cube->nz; z0 += nzc_max) {
            const int64_t nzc = std::min<int64_t>(nzc_max, cube->nz - z0);
            dim3 gy((unsigned)((nzc * cube->nx + 255) / 256), nys);
            dim3 gx((unsigned)((cube->nx + 255) / 256), (unsigned)cube->ny, (unsigned)nzc);
            const size_t lds = sizeof(float2v) * (size_t)(256 + nkx - 1);
            if (arr) {
                hipLaunchKernelGGL(spatial_wide_ypass_kernel<true>, gy, dim3(256), 0, st, A, d_k, nyp, d_inter, z0, nzc);
                hipLaunchKernelGGL(spatial_wide_xpass_kernel<true>, gx, dim3(256), lds, st, A, d_k + (nyp + 30), nkx, d_inter, z0, (float)sy);
            } else {
                hipLaunchKernelGGL(spatial_wide_ypass_kernel<false>, gy, dim3(256), 0, st, A, d_k, nyp, d_inter, z0, nzc);
                hipLaunchKernelGGL(spatial_wide_xpass_kernel<false>, gx, dim3(256), lds, st, A, d_k + (nyp + 30), nkx, d_inter, z0, (float)sy);
            }
            SPC_LAUNCH_CHECK();
        }
        return SPC_OK;
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
    // speculative all-valid fast pass: 8-byte aligned rows, a mask that rejects exactly the
    // non-finite samples (none / isfinite), whole columns per block
    const char* env = getenv("SPC_CONV_FAST");
    const bool want = env ? atoi(env) != 0 : true;
    const bool al = (cube->nx % 2 == 0) && (cube->row_stride % 2 == 0) && (cube->plane_stride % 2 == 0) &&
                    (((uintptr_t)cube->d_data) % 8 == 0) && (A.out_row_stride % 4 == 0) &&
                    (A.out_plane_stride % 4 == 0) && (((uintptr_t)d_out) % 16 == 0);
    A.status = nullptr;
    A.inv_ksum = (float)(1.0 / sum);
    // den == 0 means "centre excluded" (NaN from 0 * 1/0) only for non-negative taps with a positive centre: signed taps can
    // cancel to den == 0 over a valid centre, and then astropy's filled-centre rule needs the lookup
    A.centre_zero = (A.ky[R / 2] * A.kx[R / 2] == 0.f) ? 1 : 0;
    for (int i = 0; i < R; ++i) if (A.ky[i] < 0.f || A.kx[i] < 0.f) A.centre_zero = 1;
    { const char* xe = getenv("SPC_XCD_SWIZZLE"); A.xcd_swizzle = xe ? (atoi(xe) != 0) : 1; }
    canonical_pred(A);
    // (the 49- and 65-tap rings - 35 to 65 taps - have an all-valid kernel for isotropic kernels only: two sets of
    // that many weights do not fit the SGPR file)
    bool iso65 = R == 49 || R == 65;
    for (int i = 0; iso65 && i < R; ++i) iso65 = A.ky[i] == A.kx[i];
    if (want && al && (A.mask.flags & ~(uint32_t)SPC_MASK_FINITE) == 0 && nysplit == 1 && (R <= 33 || iso65) && cube->nx >= 64) {
        A.fast_nstrips = (int)((cube->nx + fast_txo(R) - 1) / fast_txo(R));
        const size_t nt = (size_t)A.fast_nstrips * (size_t)cube->nz;
        SpcWorkspace ws(d_workspace, workspace_bytes);
        SPC_WS_TAKE(d_status, ws, unsigned char, nt);
        SPC_HIP(spc_flags_clear(d_status, nt, st));
        A.status = d_status;
    }
    switch (R) {
        case 9: rc = launch_sep<9>(A, st, grid, arr); break;
        case 17: rc = launch_sep<17>(A, st, grid, arr); break;
        case 29: rc = launch_sep<29>(A, st, grid, arr); break;
        case 33: rc = launch_sep<33>(A, st, grid, arr); break;
        case 49: rc = launch_sep<49>(A, st, grid, arr); break;
        case 65: rc = launch_sep<65>(A, st, grid, arr); break;
        default: spc_set_error("no ring kernel for R=%d", R); rc = SPC_ERR_UNSUPPORTED;
    }
    return rc;
}

}  // extern "C"
