// build + run: hipcc --offload-arch=gfx950 -O3 -o tools/micro/vmcnt_pipeline tools/micro/vmcnt_pipeline.hip && gpurun -- tools/micro/vmcnt_pipeline
// Round 5: WHY do persistent copies stay at 4.6 - 5.5 TB/s where the one-shot copy reaches 6.3 - 6.6 (copy_ceiling.hip)?
// Hypothesis: gfx9's ONE vector-memory counter counts loads and stores in issue order.  A loop `v = load; store v` waits for
// its load with vmcnt(0) - behind the previous iteration's store, whose acknowledgement (a write to HBM) is then on every
// iteration's critical path; a wave that ends never waits for its stores.
//   naive<U>     : for (...) { U loads; U stores }                                  (= copy_ceiling's stride)
//   piped<U>     : the NEXT group's loads are issued before this group's stores; the wait for them leaves the U stores
//                  issued after them outstanding (vmcnt(U)): store acknowledgements off the critical path
//   simple       : the one-shot copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int BT>
__global__ __launch_bounds__(BT) void simple(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
    const long i = (long)blockIdx.x * BT + threadIdx.x;
    if (i < n4) out[i] = in[i];
}
template <int BT, int U, int NT>
__global__ __launch_bounds__(BT) void naive(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
    const long step = (long)gridDim.x * BT * U;
    for (long base = (long)blockIdx.x * BT * U + threadIdx.x; base < n4; base += step) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(in + base + u * BT) : in[base + u * BT];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], out + base + u * BT); else out[base + u * BT] = v[u]; }
    }
}
template <int BT, int U, int NT>
__global__ __launch_bounds__(BT) void piped(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
    const long step = (long)gridDim.x * BT * U;
    long base = (long)blockIdx.x * BT * U + threadIdx.x;
    if (base >= n4) return;                              // (n4 is a multiple of step in this test)
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(in + base + u * BT) : in[base + u * BT];
    for (;;) {
        const long nb = base + step;
        const bool more = nb < n4;                       // wave-uniform
        f4 w[U];
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = NT ? __builtin_nontemporal_load(in + nb + u * BT) : in[nb + u * BT];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], out + base + u * BT); else out[base + u * BT] = v[u]; }
        __builtin_amdgcn_sched_barrier(0);
        if (!more) break;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = w[u];
        base = nb;
    }
}
// The compiler's own s_waitcnt placement defeats piped<> (ISA: vmcnt(0) behind the conditional load - it waits for the load it
// has just issued).  By hand: loads and stores as inline asm, the waits written out.  Two register sets A / B; in the steady state
// the queue at a wait is [awaited load, store of the other set, next load]: vmcnt(2) leaves the store and the next load out.
__device__ __forceinline__ f4 ld(const f4* p) { f4 v; asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st(f4* p, const f4& v) { asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory"); }
template <int BT>
__global__ __launch_bounds__(BT) void piped_asm(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
    const long step = (long)gridDim.x * BT;
    long base = (long)blockIdx.x * BT + threadIdx.x;
    const long iters = n4 / step;                        // (n4 is a multiple of step, iters is even and >= 4)
    f4 a = ld(in + base);
    f4 b = ld(in + base + step);
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    asm volatile("" : "+v"(a));
    st(out + base, a);
    for (long k = 2; k < iters; k += 2) {
        a = ld(in + base + k * step);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          // queue: b, store a, a'  ->  b is here
        asm volatile("" : "+v"(b));
        st(out + base + (k - 1) * step, b);
        b = ld(in + base + (k + 1) * step);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          // queue: a', store b, b'  ->  a' is here
        asm volatile("" : "+v"(a));
        st(out + base + k * step, a);
    }
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    asm volatile("" : "+v"(b));
    st(out + base + (iters - 1) * step, b);
}
// the same loop waiting the way the compiler does: everything, every time
template <int BT>
__global__ __launch_bounds__(BT) void piped_asm_wait0(const f4* __restrict__ in, f4* __restrict__ out, long n4) {
    const long step = (long)gridDim.x * BT;
    long base = (long)blockIdx.x * BT + threadIdx.x;
    const long iters = n4 / step;
    for (long k = 0; k < iters; ++k) {
        f4 a = ld(in + base + k * step);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(a));
        st(out + base + k * step, a);
    }
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
static void timeit(const char* name, double bytes, F f, int reps = 12) {
    std::vector<float> t;
    float ms = 0;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-56s best %.4f ms  median %.4f ms  %.2f TB/s (median %.2f)\n", name, t[0], t[t.size() / 2], bytes / t[0] / 1e9, bytes / t[t.size() / 2] / 1e9);
    fflush(stdout);
}
int main() {
    const long n = 1L << 30, n4 = n / 4;                 // 4 GiB per buffer
    float *in, *out;
    if (hipMalloc(&in, n * 4) != hipSuccess || hipMalloc(&out, n * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEventCreate(&e0); hipEventCreate(&e1);
    fill_random<<<4096, 256>>>(in, n, 0x1234u); fill_random<<<4096, 256>>>(out, n, 0x9876u);
    hipDeviceSynchronize();
    const double rw = 2.0 * n * 4;
    timeit("simple BT=512 (one shot)", rw, [&] { simple<512><<<(unsigned)(n4 / 512), 512>>>((const f4*)in, (f4*)out, n4); });
#define RUN(K, BT, U, NT, G) timeit(#K " BT=" #BT " U=" #U " nt=" #NT " blocks/CU=" #G, rw, [&] { K<BT, U, NT><<<256 * G, BT>>>((const f4*)in, (f4*)out, n4); })
#define RUNA(K, BT, G) timeit(#K " BT=" #BT " blocks/CU=" #G, rw, [&] { K<BT><<<256 * G, BT>>>((const f4*)in, (f4*)out, n4); })
    RUNA(piped_asm_wait0, 256, 8); RUNA(piped_asm, 256, 8);
    RUNA(piped_asm_wait0, 512, 4); RUNA(piped_asm, 512, 4);
    RUNA(piped_asm_wait0, 1024, 2); RUNA(piped_asm, 1024, 2);
    RUNA(piped_asm_wait0, 256, 4); RUNA(piped_asm, 256, 4);
    {   // the result of the hand-written loop is the input
        std::vector<float> h0(1 << 16), h1(1 << 16);
        hipMemcpy(h0.data(), in + (n - (1 << 16)), 4 << 16, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), out + (n - (1 << 16)), 4 << 16, hipMemcpyDeviceToHost);
        printf("piped_asm copy %s\n", h0 == h1 ? "verified (last 64 Ki floats)" : "WRONG");
    }
    // waves in flight against bytes in flight: the persistent loop at 1 - 8 blocks of 256 per CU, 1 - 8 loads per lane
#define SWEEP(U) RUN(naive, 256, U, 1, 1); RUN(naive, 256, U, 1, 2); RUN(naive, 256, U, 1, 3); RUN(naive, 256, U, 1, 4); RUN(naive, 256, U, 1, 6); RUN(naive, 256, U, 1, 8)
    SWEEP(1); SWEEP(2); SWEEP(4); SWEEP(8);
    return 0;
    RUN(naive, 256, 1, 1, 8); RUN(piped, 256, 1, 1, 8);
    RUN(naive, 256, 2, 1, 8); RUN(piped, 256, 2, 1, 8);
    RUN(naive, 256, 4, 1, 8); RUN(piped, 256, 4, 1, 8);
    RUN(naive, 512, 1, 1, 4); RUN(piped, 512, 1, 1, 4);
    RUN(naive, 512, 2, 1, 4); RUN(piped, 512, 2, 1, 4);
    RUN(naive, 256, 1, 0, 8); RUN(piped, 256, 1, 0, 8);
    RUN(naive, 256, 2, 0, 8); RUN(piped, 256, 2, 0, 8);
    RUN(naive, 256, 4, 0, 8); RUN(piped, 256, 4, 0, 8);
    RUN(naive, 1024, 1, 1, 2); RUN(piped, 1024, 1, 1, 2);
    RUN(naive, 1024, 2, 1, 2); RUN(piped, 1024, 2, 1, 2);
    return 0;
}
