// Microbenchmark: the clock a dense VALU loop actually gets on MI355X (power management), and the issue
// cost of the FMA flavours on random (not constant) data.  Effective clock = s_memtime ticks of a wave's
// lifetime / the kernel's wall time by HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* seed, float* out, unsigned long long* ticks, int iters, float w0, float w1) {
    float2v b[16];
    const float s0 = seed[(blockIdx.x * 256 + threadIdx.x) & 0xffff];
    for (int i = 0; i < 16; ++i) b[i] = float2v{s0 * (i + 1), s0 - i};
    float2v x2 = float2v{s0 * 0.37f + 0.1f, s0 * 1.7f - 0.3f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(b[i]) : "s"(float2v{w0, w1}), "v"(x2));
                else if (MODE == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(*(double*)&b[i]) : "s"((double)w0), "v"(*(double*)&x2));
                else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(b[i].x) : "s"(w0), "v"(x2.x));
            }
        }
        x2 = x2 * 0.999f + b[it & 15] * 1e-30f;       // keep the data moving
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 16; ++i) s += b[i].x + b[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float *seed, *d; unsigned long long* tk;
    hipMalloc(&seed, 65536 * 4); hipMalloc(&d, 1 << 26); hipMalloc(&tk, 8 * 4096);
    float* h = (float*)malloc(65536 * 4);
    srand(1); for (int i = 0; i < 65536; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(seed, h, 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"v_pk_fma_f32 (SGPR weight)", "v_fma_f64 (SGPR weight)", "v_fmac_f32 (SGPR weight)"};
    for (int iters : {20000, 200000}) {
        for (int wpS : {2, 4}) {
            const int blocks = 256 * wpS;
            for (int mode = 0; mode < 3; ++mode) {
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) k<0><<<blocks, 256>>>(seed, d, tk, iters, 0.99991f, 1.00003f);
                    else if (mode == 1) k<1><<<blocks, 256>>>(seed, d, tk, iters, 0.99991f, 1.00003f);
                    else k<2><<<blocks, 256>>>(seed, d, tk, iters, 0.99991f, 1.00003f);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                unsigned long long t[1024]; hipMemcpy(t, tk, 8 * blocks, hipMemcpyDeviceToHost);
                double mean = 0; for (int i = 0; i < blocks; ++i) mean += (double)t[i]; mean /= blocks;
                const double instr = (double)iters * 65;
                printf("%-28s iters %6d waves/SIMD %d: %8.3f ms  s_memtime ticks/wave %.3e -> %.3f GHz if ticks are shader cycles; %.2f ticks per wave-instr per SIMD, %.3f ns\n",
                       names[mode], iters, wpS, ms, mean, mean / (ms * 1e6), mean / (instr * wpS), ms * 1e6 / (instr * wpS));
            }
        }
    }
    return 0;
}
