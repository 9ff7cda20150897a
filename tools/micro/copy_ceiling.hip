// build + run: hipcc --offload-arch=gfx950 -O3 -o tools/micro/copy_ceiling tools/micro/copy_ceiling.hip && gpurun -- tools/micro/copy_ceiling
// Round 5, verdict item 3: MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; the best of tools/micro/copy_patterns.hip
// (persistent grids, 4 - 16 loads in flight per lane, 4 GiB in + 4 GiB out) is 5.5 TB/s.  This file looks for the difference:
//   simple<BT>   : the textbook copy - one float4 per thread, no loop, grid = n4 / BT      (BT = 256 / 512 / 1024)
//   stride<BT,U> : persistent grid, U float4 per lane per iteration, G blocks per CU
//   xcd<BT>      : as simple, but block b works on chunk (b % 8) * (nblocks / 8) + b / 8: every XCD streams ONE contiguous
//                  eighth of the buffers
//   sizes 64 MiB ... 4 GiB per buffer (below 256 MiB the Infinity Cache holds a whole buffer), zero and random contents
//   (the chip clocks to its power budget: zero-filled buffers run faster), plain / nt loads and stores, output offsets.
// Rates are (bytes read + bytes written) / time, like the guide's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int BT, int NTL, int NTS>
__global__ __launch_bounds__(BT) void simple(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
    const long i = (long)blockIdx.x * BT + threadIdx.x;
    if (i < n4) {
        const f4 v = NTL ? __builtin_nontemporal_load(in + i) : in[i];
        if (NTS) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

template <int BT, int NTL, int NTS>
__global__ __launch_bounds__(BT) void xcd(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
    const long nb = gridDim.x, b = blockIdx.x;
    const long chunk = (b % 8) * (nb / 8) + b / 8;
    const long i = chunk * BT + threadIdx.x;
    if (i < n4) {
        const f4 v = NTL ? __builtin_nontemporal_load(in + i) : in[i];
        if (NTS) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}

template <int BT, int U, int NTL, int NTS>
__global__ __launch_bounds__(BT) void stride(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
    const long step = (long)gridDim.x * BT * U;
    for (long base = (long)blockIdx.x * BT * U + threadIdx.x; base < n4; base += step) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(in + base + u * BT) : in[base + u * BT];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NTS) __builtin_nontemporal_store(v[u], out + base + u * BT); else out[base + u * BT] = v[u]; }
    }
}

template <int BT>
__global__ __launch_bounds__(BT) void read_only(const f4* __restrict__ in, float* sink, long n4) {
    const long i = (long)blockIdx.x * BT + threadIdx.x;
    if (i < n4) { const f4 v = in[i]; if (v.x + v.y + v.z + v.w == 12345.678f) sink[0] = 1.f; }
}
template <int BT>
__global__ __launch_bounds__(BT) void write_only(f4* __restrict__ out, long n4) {
    const long i = (long)blockIdx.x * BT + threadIdx.x;
    if (i < n4) out[i] = f4{1.f, 2.f, 3.f, 4.f};
}

__global__ void fill_random(float* p, long n, unsigned seed) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long step = (long)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        unsigned x = (unsigned)i * 2654435761u ^ seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f;
    }
}

static hipEvent_t e0, e1;
template <typename F>
static double timeit(const char* name, double bytes, F f, int reps = 12) {
    std::vector<float> t;
    float ms = 0;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const float best = t[0], med = t[t.size() / 2];
    printf("%-64s best %.4f ms  median %.4f ms  %.2f TB/s (median %.2f)\n", name, best, med, bytes / best / 1e9, bytes / med / 1e9);
    fflush(stdout);
    return bytes / med / 1e9;
}

int main() {
    const long nmax = 1L << 30;          // floats per buffer (4 GiB)
    float *in, *out, *sink;
    if (hipMalloc(&in, nmax * 4) != hipSuccess || hipMalloc(&out, nmax * 4 + (64 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 64);
    hipEventCreate(&e0); hipEventCreate(&e1);
    char name[160];
    for (int random = 0; random < 2; ++random) {
        if (random) { fill_random<<<4096, 256>>>(in, nmax, 0x1234u); fill_random<<<4096, 256>>>(out, nmax, 0x9876u); }
        else { hipMemset(in, 0, nmax * 4); hipMemset(out, 0, nmax * 4); }
        hipDeviceSynchronize();
        const char* tag = random ? "random" : "zeros ";
        for (long mib : {64L, 256L, 1024L, 4096L}) {
            const long n = mib << 18, n4 = n / 4;
            const double rw = 2.0 * n * 4;
            snprintf(name, 160, "%s %4ld MiB  hipMemcpyDtoD", tag, mib);
            timeit(name, rw, [&] { hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0); });
#define SIMPLE(BT, NL, NS, label) snprintf(name, 160, "%s %4ld MiB  simple BT=%d %s", tag, mib, BT, label); \
            timeit(name, rw, [&] { simple<BT, NL, NS><<<(unsigned)((n4 + BT - 1) / BT), BT>>>((const f4*)in, (f4*)out, n4); })
            SIMPLE(256, 0, 0, "plain/plain");
            SIMPLE(512, 0, 0, "plain/plain");
            SIMPLE(1024, 0, 0, "plain/plain");
            SIMPLE(256, 1, 1, "nt/nt");
            SIMPLE(512, 1, 1, "nt/nt");
            SIMPLE(1024, 1, 1, "nt/nt");
            SIMPLE(512, 0, 1, "plain/nt");
            SIMPLE(512, 1, 0, "nt/plain");
#define XCD(BT, NL, NS, label) snprintf(name, 160, "%s %4ld MiB  xcd-contiguous BT=%d %s", tag, mib, BT, label); \
            timeit(name, rw, [&] { xcd<BT, NL, NS><<<(unsigned)((n4 + BT - 1) / BT), BT>>>((const f4*)in, (f4*)out, n4); })
            XCD(256, 0, 0, "plain/plain");
            XCD(512, 1, 1, "nt/nt");
            XCD(1024, 1, 1, "nt/nt");
#define STRIDE(BT, U, NL, NS, G, label) snprintf(name, 160, "%s %4ld MiB  stride BT=%d U=%d %d blocks/CU %s", tag, mib, BT, U, G, label); \
            timeit(name, rw, [&] { stride<BT, U, NL, NS><<<256 * G, BT>>>((const f4*)in, (f4*)out, n4); })
            STRIDE(512, 1, 1, 1, 4, "nt/nt");
            STRIDE(1024, 1, 1, 1, 2, "nt/nt");
            STRIDE(1024, 1, 0, 0, 2, "plain/plain");
            STRIDE(512, 2, 1, 1, 4, "nt/nt");
            STRIDE(1024, 2, 1, 1, 2, "nt/nt");
            STRIDE(256, 4, 1, 1, 8, "nt/nt");
            STRIDE(256, 8, 1, 1, 8, "nt/nt");
            if (mib >= 1024) {
                snprintf(name, 160, "%s %4ld MiB  read only  BT=512", tag, mib);
                timeit(name, rw / 2, [&] { read_only<512><<<(unsigned)((n4 + 511) / 512), 512>>>((const f4*)in, sink, n4); });
                snprintf(name, 160, "%s %4ld MiB  write only BT=512", tag, mib);
                timeit(name, rw / 2, [&] { write_only<512><<<(unsigned)((n4 + 511) / 512), 512>>>((f4*)out, n4); });
                if (random) fill_random<<<4096, 256>>>(out, nmax, 0x9876u);
            }
        }
        // output offsets (half of candidate channel periods) at 4 GiB
        {
            const long n = nmax, n4 = n / 4;
            const double rw = 2.0 * n * 4;
            for (long off : {128L, 2048L, 32768L, 1L << 19, 1L << 21}) {
                f4* o2 = (f4*)(out + off / 4);
                snprintf(name, 160, "%s 4096 MiB  simple BT=512 nt/nt, out + %ld B", tag, off);
                timeit(name, rw, [&] { simple<512, 1, 1><<<(unsigned)((n4 + 511) / 512), 512>>>((const f4*)in, o2, n4); });
            }
        }
    }
    return 0;
}
