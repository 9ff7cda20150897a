// build + run: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mask_patterns tools/micro/mask_patterns.hip && gpurun -- tools/micro/mask_patterns
// Microbenchmark: the moment kernel's two streams (float32 cube + uint8 mask), read-only, no arithmetic.
//   A<ZW,U>    : the kernel's pattern - a wave reads 1 KiB of data + 256 B of mask per plane, ZW waves on interleaved planes
//   B<ZW,U,K>  : a wave reads K consecutive KiB of data + K x 256 B of mask per plane (lane l owns columns 4l + 256k .. + 3)
//   C<ZW,U>    : ZW waves side by side on ONE plane (ZW KiB of data + ZW x 256 B of mask contiguous per block and plane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long xcd_group(long b, long nb) { const long q = nb >> 3, r = nb & 7, k = b & 7; return k * q + (k < r ? k : r) + (b >> 3); }

template <int ZW, int U, int K, bool SIDE>
__global__ __launch_bounds__(64 * ZW) void rd(const float* __restrict__ in, const uint8_t* __restrict__ mk, float* sink, long nz, long ncols,
                                             int remap) {
    long b = blockIdx.x;
    if (remap) b = xcd_group(b, gridDim.x);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    // SIDE: the ZW waves of a block sit side by side on one plane; otherwise they take interleaved planes
    const long col0 = SIDE ? ((b * ZW + w) * 64 * K + threadIdx.x) * 4 : (b * 64 * K + threadIdx.x) * 4;
    if (col0 >= ncols) return;
    const long zstep = SIDE ? 1 : ZW;
    f4 acc{};
    unsigned macc = 0;
    for (long z = SIDE ? 0 : w; z + (long)(U - 1) * zstep < nz; z += (long)U * zstep) {
        f4 v[U][K];
        unsigned m[U][K];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                v[u][k] = __builtin_nontemporal_load((const f4*)(in + (z + u * zstep) * ncols + col0 + k * 256));
                m[u][k] = __builtin_nontemporal_load((const unsigned*)(mk + (z + u * zstep) * ncols + col0 + k * 256));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < K; ++k) { acc += v[u][k]; macc += m[u][k]; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f || macc == 0x12345u) sink[0] = 1;
}

// W<ZW,U>: the data as in A (16 bytes per lane and plane); the mask in 16-byte requests - lane l of a quad of lanes reads the quad's
// 16 mask bytes of plane (u & ~3) + (l & 3): U / 4 mask requests per lane for U planes instead of U (the moment kernel would then
// hand the four dwords around inside the quad)
template <int ZW, int U>
__global__ __launch_bounds__(64 * ZW) void rdw(const float* __restrict__ in, const uint8_t* __restrict__ mk, float* sink, long nz, long ncols, int remap) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    long b = blockIdx.x;
    if (remap) b = xcd_group(b, gridDim.x);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const long col0 = (b * 64 + threadIdx.x) * 4, colq = (b * 64 + (threadIdx.x & ~3)) * 4;
    if (col0 >= ncols) return;
    const long zstep = ZW;
    f4 acc{};
    u4 macc{};
    for (long z = w; z + (long)(U - 1) * zstep < nz; z += (long)U * zstep) {
        f4 v[U];
        u4 m[U / 4];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const f4*)(in + (z + u * zstep) * ncols + col0));
#pragma unroll
        for (int u = 0; u < U / 4; ++u) m[u] = __builtin_nontemporal_load((const u4*)(mk + (z + (4 * u + (threadIdx.x & 3)) * zstep) * ncols + colq));
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
#pragma unroll
        for (int u = 0; u < U / 4; ++u) macc += m[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f || (macc.x ^ macc.y ^ macc.z ^ macc.w) == 0x12345u) sink[0] = 1;
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
    printf("%-72s best %.3f ms  mean %.3f ms  %.2f TB/s\n", name, best, sum / 4, bytes / best / 1e9);
    fflush(stdout);
}
template <int ZW, int U, int K, bool SIDE>
static void run(const float* in, const uint8_t* mk, float* sink, long nz, long ncols) {
    for (int remap : {0, 1}) {
        char name[160];
        snprintf(name, 160, "%s ZW %d, U %d, %d KiB + %d B per wave and plane, xcd grouping %d", SIDE ? "side by side" : "interleaved ", ZW, U, K, 256 * K, remap);
        dim3 grid((unsigned)(ncols / 4 / 64 / K / (SIDE ? ZW : 1))), block(64, ZW);
        timeit(name, (double)nz * ncols * 5, [&] { rd<ZW, U, K, SIDE><<<grid, block>>>(in, mk, sink, nz, ncols, remap); });
    }
}
int main() {
    for (long shape = 0; shape < 2; ++shape) {
        const long nz = shape ? 256 : 1024, ncols = shape ? 4096L * 1024 : 1024 * 1024;
        printf("# %ld planes of %ld columns\n", nz, ncols);
        float *in, *sink; uint8_t* mk;
        hipMalloc(&in, nz * ncols * 4); hipMalloc(&mk, nz * ncols); hipMalloc(&sink, 64);
        hipMemset(in, 0, nz * ncols * 4); hipMemset(mk, 1, nz * ncols);
        hipEventCreate(&e0); hipEventCreate(&e1);
        run<4, 8, 1, false>(in, mk, sink, nz, ncols);
        for (int remap : {0, 1}) {
            char name[160];
            snprintf(name, 160, "16-byte mask requests ZW 4, U 8, 1 KiB + 4 x 1 KiB / 4 planes per wave, xcd grouping %d", remap);
            timeit(name, (double)nz * ncols * 5, [&] { rdw<4, 8><<<dim3((unsigned)(ncols / 4 / 64)), dim3(64, 4)>>>(in, mk, sink, nz, ncols, remap); });
            snprintf(name, 160, "16-byte mask requests ZW 8, U 8, xcd grouping %d", remap);
            timeit(name, (double)nz * ncols * 5, [&] { rdw<8, 8><<<dim3((unsigned)(ncols / 4 / 64)), dim3(64, 8)>>>(in, mk, sink, nz, ncols, remap); });
        }
        run<8, 8, 1, false>(in, mk, sink, nz, ncols);
        run<4, 4, 1, false>(in, mk, sink, nz, ncols);
        run<4, 4, 2, false>(in, mk, sink, nz, ncols);
        run<4, 2, 4, false>(in, mk, sink, nz, ncols);
        run<4, 4, 4, false>(in, mk, sink, nz, ncols);
        run<2, 4, 4, false>(in, mk, sink, nz, ncols);
        run<1, 4, 4, false>(in, mk, sink, nz, ncols);
        run<8, 2, 4, false>(in, mk, sink, nz, ncols);
        run<4, 8, 1, true>(in, mk, sink, nz, ncols);
        run<4, 4, 1, true>(in, mk, sink, nz, ncols);
        run<4, 4, 2, true>(in, mk, sink, nz, ncols);
        run<4, 2, 4, true>(in, mk, sink, nz, ncols);
        hipFree(in); hipFree(mk); hipFree(sink);
    }
    return 0;
}
