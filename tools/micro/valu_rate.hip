// Microbenchmark: issue rate of v_fmac_f32 vs v_pk_fma_f32 on gfx950 (wave64).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float w) {
    float a[16]; float2v b[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = float2v{a[i], a[i] + 1.f}; }
    float x = out[threadIdx.x & 63];
    float2v x2 = float2v{x, x + 1.f};
    float2v w2 = float2v{w, w};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(w), "v"(x));
                else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(w2), "v"(x2));
                else asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(*(double*)&b[i]) : "v"(*(double*)&w2), "v"(*(double*)&x2));
            }
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + b[i].x + b[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 1 << 26);
    hipMemset(d, 0, 1 << 26);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wpS = 1; wpS <= 4; wpS *= 2) {   // waves per SIMD
        const int blocks = 256 * wpS;           // 256 threads = 4 waves = 1 per SIMD
        for (int mode = 0; mode < 3; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<blocks, 256>>>(d, iters, 1.0001f);
                else if (mode == 1) k<1><<<blocks, 256>>>(d, iters, 1.0001f);
                else k<2><<<blocks, 256>>>(d, iters, 1.0001f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double instr_per_wave = (double)iters * 64;
            double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * wpS);
            printf("mode %d (%s) waves/SIMD %d: %.3f ms  -> %.3f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n",
                   mode, mode == 0 ? "v_fmac_f32" : mode == 1 ? "v_pk_fma_f32" : "v_fma_f64", wpS, ms,
                   ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
        }
    }
    return 0;
}
