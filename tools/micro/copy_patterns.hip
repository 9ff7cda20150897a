// build + run: hipcc --offload-arch=gfx950 -O3 -o tools/micro/copy_patterns tools/micro/copy_patterns.hip && gpurun -- tools/micro/copy_patterns
// Microbenchmark: what bounds a cube -> cube stream on MI355X?  The z-march pattern of the stencil kernels
// copies at ~5.0 TB/s (tools/micro/zmarch_copy.hip); the guide quotes 6.29 TB/s for a float4 copy.  This file
// measures the copy rate as a function of the ORDER in which a launch touches memory:
//   lin      : grid-stride float4 copy, U loads in flight, then U stores (persistent grid of G blocks per CU)
//   oneshot  : one block per contiguous chunk of CH bytes
//   zmarch   : every lane marches over z (the stencil kernels' pattern), optional z slabs (grid.y)
//   zmarch-w : as zmarch, a block covers W consecutive 1-KiB row segments (wider footprint per plane)
//   memcpy   : hipMemcpyDtoD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(256) void lin(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
    const long stride = (long)gridDim.x * 256 * U;
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(in + base + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u] * 1.5f, out + base + u * 256);
            else out[base + u * 256] = v[u] * 1.5f;
        }
    }
}

// every lane owns 4 consecutive x and walks z in [z0, z1); grid.y = number of z slabs
template <int U, int NT>
__global__ __launch_bounds__(256) void zmarch(const float* __restrict__ in, float* __restrict__ out, long nz, long plane) {
    const long g = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (g >= plane) return;
    const long zs = (nz + gridDim.y - 1) / gridDim.y;
    const long z0 = blockIdx.y * zs, z1 = (z0 + zs < nz) ? z0 + zs : nz;
    for (long z = z0; z + U <= z1; z += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const f4*)(in + (z + u) * plane + g));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u] * 1.5f, (f4*)(out + (z + u) * plane + g));
            else *(f4*)(out + (z + u) * plane + g) = v[u] * 1.5f;
        }
    }
}

// z-march copy with separate plane strides (is a power-of-two plane stride what holds the march at 5.0 TB/s?)
template <int U>
__global__ __launch_bounds__(256) void zmarch_pad(const float* __restrict__ in, float* __restrict__ out, long nz, long plane, long ps_in, long ps_out) {
    const long g = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (g >= plane) return;
    for (long z = 0; z + U <= nz; z += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const f4*)(in + (z + u) * ps_in + g));
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u] * 1.5f, (f4*)(out + (z + u) * ps_out + g));
    }
}

// read-only forms for the ceiling
template <int U>
__global__ __launch_bounds__(256) void lin_read(const f4* __restrict__ in, float* sink, long n4) {
    const long stride = (long)gridDim.x * 256 * U;
    f4 acc{};
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += __builtin_nontemporal_load(in + base + u * 256);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) sink[0] = 1;
}
// write-only
template <int U, int NT>
__global__ __launch_bounds__(256) void lin_write(f4* __restrict__ out, long n4) {
    const long stride = (long)gridDim.x * 256 * U;
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v, out + base + u * 256);
            else out[base + u * 256] = v;
        }
    }
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
    printf("%-44s best %.3f ms  mean %.3f ms  %.2f TB/s\n", name, best, sum / 4, bytes / best / 1e9);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long nz = 1024, plane = 1024 * 1024;
    const long n = nz * plane, n4 = n / 4;
    float *in, *out, *sink;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4 + (64 << 20)); hipMalloc(&sink, 64);
    hipMemset(in, 0, n * 4); hipMemset(out, 0, n * 4);
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double rw = 2.0 * n * 4, ro = 1.0 * n * 4;
    char name[128];
    timeit("hipMemcpyDtoD", rw, [&] { hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0); });
    for (int bpc : {2, 4, 8, 16}) {
        const int G = 256 * bpc;
        snprintf(name, 128, "lin read  U8, %d blocks/CU", bpc);
        timeit(name, ro, [&] { lin_read<8><<<G, 256>>>((const f4*)in, sink, n4); });
    }
    for (int bpc : {4, 8}) {
        const int G = 256 * bpc;
        snprintf(name, 128, "lin write U8 plain, %d blocks/CU", bpc);
        timeit(name, ro, [&] { lin_write<8, 0><<<G, 256>>>((f4*)out, n4); });
        snprintf(name, 128, "lin write U8 nt, %d blocks/CU", bpc);
        timeit(name, ro, [&] { lin_write<8, 1><<<G, 256>>>((f4*)out, n4); });
    }
    for (int bpc : {2, 4, 8, 16}) {
        const int G = 256 * bpc;
        snprintf(name, 128, "lin copy U4 plain, %d blocks/CU", bpc);
        timeit(name, rw, [&] { lin<4, 0><<<G, 256>>>((const f4*)in, (f4*)out, n4); });
        snprintf(name, 128, "lin copy U4 nt, %d blocks/CU", bpc);
        timeit(name, rw, [&] { lin<4, 1><<<G, 256>>>((const f4*)in, (f4*)out, n4); });
        snprintf(name, 128, "lin copy U8 nt, %d blocks/CU", bpc);
        timeit(name, rw, [&] { lin<8, 1><<<G, 256>>>((const f4*)in, (f4*)out, n4); });
        snprintf(name, 128, "lin copy U16 nt, %d blocks/CU", bpc);
        timeit(name, rw, [&] { lin<16, 1><<<G, 256>>>((const f4*)in, (f4*)out, n4); });
    }
    {   // one block per chunk: the whole grid, hardware dispatch order
        const int G = (int)(n4 / (256 * 8));
        timeit("oneshot copy U8 nt (1 chunk of 32 KiB per block)", rw, [&] { lin<8, 1><<<G, 256>>>((const f4*)in, (f4*)out, n4); });
        timeit("oneshot copy U8 plain", rw, [&] { lin<8, 0><<<G, 256>>>((const f4*)in, (f4*)out, n4); });
    }
    for (int slabs : {1, 2, 4, 8, 32}) {
        dim3 grid((unsigned)(plane / 4 / 256), slabs);
        snprintf(name, 128, "zmarch copy U8 nt, %d z slabs", slabs);
        timeit(name, rw, [&] { zmarch<8, 1><<<grid, 256>>>(in, out, nz, plane); });
        snprintf(name, 128, "zmarch copy U8 plain, %d z slabs", slabs);
        timeit(name, rw, [&] { zmarch<8, 0><<<grid, 256>>>(in, out, nz, plane); });
    }
    {
        dim3 grid((unsigned)(plane / 4 / 256), 1);
        timeit("zmarch copy U2 nt", rw, [&] { zmarch<2, 1><<<grid, 256>>>(in, out, nz, plane); });
        timeit("zmarch copy U4 nt", rw, [&] { zmarch<4, 1><<<grid, 256>>>(in, out, nz, plane); });
        timeit("zmarch copy U16 nt", rw, [&] { zmarch<16, 1><<<grid, 256>>>(in, out, nz, plane); });
    }
    for (long off : {0L, 256L, 4096L, 65536L, 1L << 20, (1L << 20) + 4096, (16L << 20) + 65536 + 4096}) {
        dim3 grid((unsigned)(plane / 4 / 256), 1);
        float* o2 = out + off / 4;
        snprintf(name, 128, "zmarch copy U8 nt, out + %ld B", off);
        timeit(name, rw, [&] { zmarch<8, 1><<<grid, 256>>>(in, o2, nz, plane); });
        snprintf(name, 128, "lin copy U8 nt 8/CU, out + %ld B", off);
        timeit(name, rw, [&] { lin<8, 1><<<2048, 256>>>((const f4*)in, (f4*)o2, n4); });
    }
    {   // padded plane strides: 1000 planes so that the padded cubes fit the same allocations
        const long nzp = 1000;
        dim3 grid((unsigned)(plane / 4 / 256), 1);
        for (long pi : {0L, 1088L}) for (long po : {0L, 1088L, 16384L + 1088}) {
            snprintf(name, 128, "zmarch copy U8 nt, plane stride + %ld in / + %ld out", pi, po);
            timeit(name, 2.0 * nzp * plane * 4, [&] { zmarch_pad<8><<<grid, 256>>>(in, out, nzp, plane, plane + pi, plane + po); });
        }
        timeit("zmarch copy U4 nt, + 1088 / + 1088", 2.0 * nzp * plane * 4, [&] { zmarch_pad<4><<<grid, 256>>>(in, out, nzp, plane, plane + 1088, plane + 1088); });
    }
    return 0;
}
