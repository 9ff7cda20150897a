// build + run: hipcc --offload-arch=gfx950 -O3 -o tools/micro/read_patterns tools/micro/read_patterns.hip && gpurun -- tools/micro/read_patterns
// Microbenchmark: why does the moment kernel's z-march read 6.0 - 6.5 TB/s when a linear float4 read of the same
// bytes reaches 7.1 - 7.2 TB/s (tools/micro/copy_patterns.hip)?  Read-only variants of the march:
//   zm<LB,ZW,U> : a block of 64 x ZW lanes; ZW waves take interleaved planes of the same 64 x LB/4 columns
//                 (LB = bytes per lane: 4 / 8 / 16), U planes in flight per lane
//   remap       : 0 = blockIdx as dispatched, 1 = the blocks of one XCD take neighbouring column groups,
//                 2 = reversed bit order (spread)
//   pad         : plane stride = ny * nx + pad floats (is a power-of-two plane stride a DRAM bank conflict?)
//   slabs       : gridDim.y z slabs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int LB> struct V;
template <> struct V<16> { using T = f4; };
template <> struct V<8> { using T = f2; };
template <> struct V<4> { using T = float; };
__device__ __forceinline__ float hsum(f4 v) { return v.x + v.y + v.z + v.w; }
__device__ __forceinline__ float hsum(f2 v) { return v.x + v.y; }
__device__ __forceinline__ float hsum(float v) { return v; }

template <int LB, int ZW, int U>
__global__ __launch_bounds__(64 * ZW) void zm(const float* __restrict__ in, float* sink, long nz, long ncols, long plane_stride,
                                              int remap) {
    using T = typename V<LB>::T;
    constexpr int VEC = LB / 4;
    long b = blockIdx.x;
    const long nb = gridDim.x;
    if (remap == 1) b = (b & 7) * (nb >> 3) + (b >> 3);
    const long g = (b * 64 + threadIdx.x) * VEC;
    if (g >= ncols) return;
    const long zs = (nz + gridDim.y - 1) / gridDim.y;
    const long z0 = blockIdx.y * zs, z1 = (z0 + zs < nz) ? z0 + zs : nz;
    T acc{};
    for (long z = z0 + threadIdx.y; z + (long)(U - 1) * ZW < z1; z += (long)U * ZW) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const T*)(in + (z + (long)u * ZW) * plane_stride + g));
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (hsum(acc) == 12345.f) sink[0] = 1;
}

// zm with a per-block rotation of the march: block b starts at plane (b * rot) % nz and wraps (nz % (U * ZW) == 0)
template <int LB, int ZW, int U>
__global__ __launch_bounds__(64 * ZW) void zr(const float* __restrict__ in, float* sink, long nz, long ncols, long plane_stride,
                                              int rot, int remap) {
    using T = typename V<LB>::T;
    constexpr int VEC = LB / 4;
    long b = blockIdx.x;
    const long nb = gridDim.x;
    if (remap == 1) b = (b & 7) * (nb >> 3) + (b >> 3);
    const long g = (b * 64 + threadIdx.x) * VEC;
    if (g >= ncols) return;
    const long step = (long)U * ZW;
    long z = (((remap == 2 ? (long)blockIdx.x : b) * rot) % (nz / step)) * step + threadIdx.y;
    T acc{};
    for (long it = 0; it < nz / step; ++it) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const T*)(in + (z + (long)u * ZW) * plane_stride + g));
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
        z += step;
        if (z >= nz) z -= nz;
    }
    if (hsum(acc) == 12345.f) sink[0] = 1;
}

// all 64 * ZW lanes on the SAME plane (a block reads 64 * ZW * LB contiguous bytes per plane)
template <int LB, int ZW, int U>
__global__ __launch_bounds__(64 * ZW) void zw(const float* __restrict__ in, float* sink, long nz, long ncols, long plane_stride,
                                              int remap) {
    using T = typename V<LB>::T;
    constexpr int VEC = LB / 4;
    long b = blockIdx.x;
    const long nb = gridDim.x;
    if (remap == 1) b = (b & 7) * (nb >> 3) + (b >> 3);
    const long g = (b * 64 * ZW + threadIdx.y * 64 + threadIdx.x) * VEC;
    if (g >= ncols) return;
    const long zs = (nz + gridDim.y - 1) / gridDim.y;
    const long z0 = blockIdx.y * zs, z1 = (z0 + zs < nz) ? z0 + zs : nz;
    T acc{};
    for (long z = z0; z + U <= z1; z += U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const T*)(in + (z + u) * plane_stride + g));
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (hsum(acc) == 12345.f) sink[0] = 1;
}

template <int U>
__global__ __launch_bounds__(256) void lin_read(const f4* __restrict__ in, float* sink, long n4) {
    const long stride = (long)gridDim.x * 256 * U;
    f4 acc{};
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += __builtin_nontemporal_load(in + base + u * 256);
    }
    if (hsum(acc) == 12345.f) sink[0] = 1;
}

static hipEvent_t e0, e1;
template <typename F>
static void timeit(const char* name, double bytes, F f) {
    float best = 1e9, ms = 0, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-64s best %.3f ms  mean %.3f ms  %.2f TB/s\n", name, best, sum / 4, bytes / best / 1e9);
    fflush(stdout);
}

template <int LB, int ZW, int U>
static void run_zm(const float* in, float* sink, long nz, long ncols, long pad, int slabs, int remap) {
    char name[160];
    snprintf(name, 160, "zm  %2d B/lane, %d waves on interleaved planes, U%d, pad %ld, slabs %d, remap %d", LB, ZW, U, pad, slabs, remap);
    dim3 grid((unsigned)(ncols / (LB / 4) / 64), slabs), block(64, ZW);
    timeit(name, (double)nz * ncols * 4, [&] { zm<LB, ZW, U><<<grid, block>>>(in, sink, nz, ncols, ncols + pad, remap); });
}
template <int LB, int ZW, int U>
static void run_zw(const float* in, float* sink, long nz, long ncols, long pad, int slabs, int remap) {
    char name[160];
    snprintf(name, 160, "zw  %2d B/lane, %d waves on one plane,         U%d, pad %ld, slabs %d, remap %d", LB, ZW, U, pad, slabs, remap);
    dim3 grid((unsigned)(ncols / (LB / 4) / 64 / ZW), slabs), block(64, ZW);
    timeit(name, (double)nz * ncols * 4, [&] { zw<LB, ZW, U><<<grid, block>>>(in, sink, nz, ncols, ncols + pad, remap); });
}

int main() {
    const long nz = 1024, ncols = 1024 * 1024;
    const long maxpad = 1 << 16;
    float *in, *sink;
    hipMalloc(&in, nz * (ncols + maxpad) * 4); hipMalloc(&sink, 64);
    hipMemset(in, 0, nz * (ncols + maxpad) * 4);
    hipEventCreate(&e0); hipEventCreate(&e1);
    timeit("lin read U8, 8 blocks/CU", (double)nz * ncols * 4, [&] { lin_read<8><<<2048, 256>>>((const f4*)in, sink, nz * ncols / 4); });
    // the moment kernel's pattern and its neighbours
    run_zm<16, 4, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zm<16, 4, 8>(in, sink, nz, ncols, 0, 1, 1);
    run_zm<16, 4, 4>(in, sink, nz, ncols, 0, 1, 0);
    run_zm<16, 4, 16>(in, sink, nz, ncols, 0, 1, 0);
    run_zm<16, 8, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zm<16, 2, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zm<16, 1, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zm<8, 4, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zm<4, 4, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zm<4, 1, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zm<4, 1, 16>(in, sink, nz, ncols, 0, 1, 0);
    for (long pad : {64L, 1024L, 1024L + 64, 16384L, 16384L + 1024 + 64, 65536L - 1024}) {
        run_zm<16, 4, 8>(in, sink, nz, ncols, pad, 1, 0);
    }
    for (int slabs : {2, 4, 8}) run_zm<16, 4, 8>(in, sink, nz, ncols, 0, slabs, 0);
    run_zw<16, 4, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zw<16, 4, 8>(in, sink, nz, ncols, 0, 1, 1);
    run_zw<16, 4, 8>(in, sink, nz, ncols, 0, 2, 0);
    run_zw<16, 4, 8>(in, sink, nz, ncols, 0, 4, 0);
    run_zw<16, 4, 8>(in, sink, nz, ncols, 1024 + 64, 1, 0);
    run_zw<16, 4, 4>(in, sink, nz, ncols, 0, 4, 0);
    run_zw<16, 4, 16>(in, sink, nz, ncols, 0, 1, 0);
    run_zw<16, 1, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zw<16, 1, 8>(in, sink, nz, ncols, 0, 1, 1);
    run_zw<8, 4, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zw<4, 4, 8>(in, sink, nz, ncols, 0, 1, 0);
    run_zw<4, 4, 16>(in, sink, nz, ncols, 0, 1, 0);
    run_zw<4, 4, 8>(in, sink, nz, ncols, 0, 1, 1);
    run_zw<4, 1, 8>(in, sink, nz, ncols, 0, 1, 0);
    for (int rot : {0, 1, 3, 7, 13}) for (int remap : {0, 1, 2}) {
        char name[160];
        snprintf(name, 160, "zr  16 B/lane, 4 waves interleaved, U8, start plane (blk * %d) %% nz, remap %d", rot, remap);
        dim3 grid((unsigned)(ncols / 4 / 64), 1), block(64, 4);
        timeit(name, (double)nz * ncols * 4, [&] { zr<16, 4, 8><<<grid, block>>>(in, sink, nz, ncols, ncols, rot, remap); });
    }
    for (int rot : {0, 1, 7}) {
        char name[160];
        snprintf(name, 160, "zr  16 B/lane, 4 waves interleaved, U4, start plane (blk * %d) %% nz", rot);
        dim3 grid((unsigned)(ncols / 4 / 64), 1), block(64, 4);
        timeit(name, (double)nz * ncols * 4, [&] { zr<16, 4, 4><<<grid, block>>>(in, sink, nz, ncols, ncols, rot, 0); });
        snprintf(name, 160, "zr  16 B/lane, 8 waves interleaved, U8, start plane (blk * %d) %% nz", rot);
        dim3 block8(64, 8);
        timeit(name, (double)nz * ncols * 4, [&] { zr<16, 8, 8><<<grid, block8>>>(in, sink, nz, ncols, ncols, rot, 0); });
    }
    // the north-star plane (2048 x 2048) at 256 planes: the same bytes
    run_zm<16, 4, 8>(in, sink, 256, 4 * ncols, 0, 1, 0);
    run_zw<16, 4, 8>(in, sink, 256, 4 * ncols, 0, 1, 0);
    run_zw<4, 4, 8>(in, sink, 256, 4 * ncols, 0, 1, 0);
    return 0;
}
