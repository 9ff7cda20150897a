// build + run: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_valu_rates tools/micro/mfma_valu_rates.hip && gpurun -- tools/micro/mfma_valu_rates
// Round 5: issue cost (shader cycles per wave-instruction, s_memtime) of what the fp16 split stencil is made of, at one and
// at two waves per SIMD, alone and mixed:
//   matrix: v_mfma_f32_16x16x16_f16 (the K = 16 form carried forward), v_mfma_f32_16x16x32_f16, v_mfma_f32_16x16x4_f32
//   vector: v_fma_f32, v_pk_fma_f32, v_cvt_pk_f16_f32, v_fma_mix_f32, v_fma_mixlo_f16, v_cndmask_b32, v_cmp_le_f32,
//           v_cmp_ne_u32_sdwa, v_rcp_f32, v_max3_f32, v_pk_mul_f32
//   mixed : one K = 32 (or two K = 16) matrix instruction(s) followed by k independent vector instructions, k = 0 .. 8:
//           where does the vector work stop hiding behind the matrix pipe (one wave alone / two waves per SIMD)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

struct Out { long cycles; };

template <int KIND>
__global__ void rate(Out* out, int iters, float seed) {
    f32x4 acc0 = {seed, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    half4 a4 = {(_Float16)seed, 1, 2, 3}, b4 = {1, (_Float16)seed, 0, 1};
    half8 a8 = {(_Float16)seed, 1, 2, 3, 0, 1, 2, 3}, b8 = {1, (_Float16)seed, 0, 1, 1, 0, 1, 0};
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, v7 = seed + 7;
    float2v p0 = {seed, 1}, p1 = {seed, 2}, p2 = {seed, 3}, p3 = {seed, 4};
    unsigned u0 = __builtin_bit_cast(unsigned, seed), u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {        // 16 x K16 mfma, 4 chains
            REP4(asm volatile("v_mfma_f32_16x16x16_f16 %0, %4, %5, %0\n v_mfma_f32_16x16x16_f16 %1, %4, %5, %1\n v_mfma_f32_16x16x16_f16 %2, %4, %5, %2\n v_mfma_f32_16x16x16_f16 %3, %4, %5, %3"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a4), "v"(b4));)
        } else if (KIND == 1) { // 16 x K32 mfma
            REP4(asm volatile("v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_f16 %1, %4, %5, %1\n v_mfma_f32_16x16x32_f16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_f16 %3, %4, %5, %3"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a8), "v"(b8));)
        } else if (KIND == 2) { // 16 x f32 mfma
            REP4(asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(v0), "v"(v1));)
        } else if (KIND == 3) { // v_fma_f32 x 16
            REP4(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5));)
        } else if (KIND == 4) { // v_pk_fma_f32
            REP4(asm volatile("v_pk_fma_f32 %0, %4, %4, %0\n v_pk_fma_f32 %1, %4, %4, %1\n v_pk_fma_f32 %2, %4, %4, %2\n v_pk_fma_f32 %3, %4, %4, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));)
        } else if (KIND == 5) { // v_cvt_pk_f16_f32
            REP4(asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %5, %4\n v_cvt_pk_f16_f32 %2, %4, %4\n v_cvt_pk_f16_f32 %3, %5, %5" : "=v"(u0), "=v"(u1), "=v"(u2), "=v"(u3) : "v"(v4), "v"(v5));)
        } else if (KIND == 6) { // v_fma_mix_f32 (f16 source)
            REP4(asm volatile("v_fma_mix_f32 %0, %4, %5, %6 op_sel_hi:[0,0,1]\n v_fma_mix_f32 %1, %4, %5, %6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n v_fma_mix_f32 %2, %5, %4, %6 op_sel_hi:[0,0,1]\n v_fma_mix_f32 %3, %5, %4, %6 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                              : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4), "v"(v5), "v"(u0));)
        } else if (KIND == 7) { // v_fma_mixlo_f16
            REP4(asm volatile("v_fma_mixlo_f16 %0, %4, %5, %6\n v_fma_mixhi_f16 %0, %4, %5, %6\n v_fma_mixlo_f16 %1, %4, %5, %6\n v_fma_mixhi_f16 %1, %4, %5, %6" : "+v"(u0), "+v"(u1) , "+v"(u2), "+v"(u3): "v"(v4), "v"(v5), "v"(v6));)
        } else if (KIND == 8) { // v_cmp + v_cndmask pairs
            REP4(asm volatile("v_cmp_le_f32 vcc, |%4|, %5\n v_cndmask_b32 %0, 0, %4, vcc\n v_cmp_le_f32 vcc, |%5|, %4\n v_cndmask_b32 %1, 0, %5, vcc" : "=v"(v0), "=v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5) : "vcc");)
        } else if (KIND == 9) { // sdwa compare
            REP4(asm volatile("v_cmp_ne_u32_sdwa vcc, %4, %5 src0_sel:BYTE_0 src1_sel:DWORD\n v_cmp_ne_u32_sdwa vcc, %4, %5 src0_sel:BYTE_1 src1_sel:DWORD\n v_cmp_ne_u32_sdwa vcc, %4, %5 src0_sel:BYTE_2 src1_sel:DWORD\n v_cmp_ne_u32_sdwa vcc, %4, %5 src0_sel:BYTE_3 src1_sel:DWORD"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0), "v"(u1) : "vcc");)
        } else if (KIND == 10) { // v_rcp_f32
            REP4(asm volatile("v_rcp_f32 %0, %4\n v_rcp_f32 %1, %5\n v_rcp_f32 %2, %4\n v_rcp_f32 %3, %5" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4), "v"(v5));)
        } else if (KIND == 11) { // v_max3_f32 with abs
            REP4(asm volatile("v_max3_f32 %0, |%4|, |%5|, %0\n v_max3_f32 %1, |%4|, |%5|, %1\n v_max3_f32 %2, |%4|, |%5|, %2\n v_max3_f32 %3, |%4|, |%5|, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "v"(v5));)
        } else if (KIND == 12) { // v_pk_mul_f32
            REP4(asm volatile("v_pk_mul_f32 %0, %4, %4\n v_pk_mul_f32 %1, %4, %4\n v_pk_mul_f32 %2, %4, %4\n v_pk_mul_f32 %3, %4, %4" : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(p0));)
        } else if (KIND == 13) { // sdwa cndmask into the high half
            REP4(asm volatile("v_cndmask_b32_sdwa %0, %4, %5, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n v_cndmask_b32_sdwa %1, %4, %5, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"
                              "v_cndmask_b32_sdwa %2, %4, %5, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n v_cndmask_b32_sdwa %3, %4, %5, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0), "v"(u1) : "vcc");)
        }
    }
    const long t1 = clock64();
    float s = acc0.x + acc1.x + acc2.x + acc3.x + v0 + v1 + v2 + v3 + p0.x + p1.x + p2.x + p3.x + (float)(u0 + u1 + u2 + u3);
    if (s == 12345.678f) out[0].cycles = 0;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6].cycles = t1 - t0;
}

// mixed: per group: M matrix instructions (K32 if M32 else K16 x 2) on rotating accumulators + KV independent v_fma_f32
template <int KV, bool M32>
__global__ void mixed(Out* out, int iters, float seed) {
    f32x4 acc0 = {seed, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    half4 a4 = {(_Float16)seed, 1, 2, 3}, b4 = {1, (_Float16)seed, 0, 1};
    half8 a8 = {(_Float16)seed, 1, 2, 3, 0, 1, 2, 3}, b8 = {1, (_Float16)seed, 0, 1, 1, 0, 1, 0};
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    const float c0 = seed * 0.5f, c1 = seed * 0.25f;
    const long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4& acc = (g & 3) == 0 ? acc0 : (g & 3) == 1 ? acc1 : (g & 3) == 2 ? acc2 : acc3;
            if (M32) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a8), "v"(b8));
            else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a4), "v"(b4));
#pragma unroll
            for (int k = 0; k < KV; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k]) : "v"(c0), "v"(c1));
        }
    }
    const long t1 = clock64();
    float s = acc0.x + acc1.x + acc2.x + acc3.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[0].cycles = 0;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6].cycles = t1 - t0;
}

template <typename L>
static void run(const char* name, int bt, int per_iter, L launch) {
    static Out* d = nullptr;
    if (!d) hipMalloc(&d, sizeof(Out) * 256 * 16);
    const int iters = 2000;
    std::vector<Out> h(256 * (bt / 64));
    double med = 0;
    for (int rep = 0; rep < 3; ++rep) {
        launch(d, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, sizeof(Out) * h.size(), hipMemcpyDeviceToHost);
        std::vector<double> c;
        for (auto& o : h) c.push_back((double)o.cycles / iters / per_iter);
        std::sort(c.begin(), c.end());
        med = c[c.size() / 2];
    }
    printf("%-58s %d waves/SIMD: %7.2f cycles per instruction (wave view) -> %6.2f per SIMD\n", name, bt / 256, med, med / (bt / 256));
    fflush(stdout);
}

#define RATE(KIND, label) for (int bt : {256, 512}) run(label, bt, 16, [&](Out* d, int it) { rate<KIND><<<256, bt>>>(d, it, 1.0f); })
#define MIXED(KV, M32, label) for (int bt : {256, 512}) run(label, bt, 8, [&](Out* d, int it) { mixed<KV, M32><<<256, bt>>>(d, it, 1.0f); })

int main() {
    RATE(0, "v_mfma_f32_16x16x16_f16");
    RATE(1, "v_mfma_f32_16x16x32_f16");
    RATE(2, "v_mfma_f32_16x16x4_f32");
    RATE(3, "v_fma_f32");
    RATE(4, "v_pk_fma_f32");
    RATE(5, "v_cvt_pk_f16_f32");
    RATE(6, "v_fma_mix_f32");
    RATE(7, "v_fma_mixlo/hi_f16");
    RATE(8, "v_cmp_le_f32 + v_cndmask_b32 (per instruction)");
    RATE(9, "v_cmp_ne_u32_sdwa");
    RATE(10, "v_rcp_f32");
    RATE(11, "v_max3_f32 |.|");
    RATE(12, "v_pk_mul_f32");
    RATE(13, "v_cndmask_b32_sdwa WORD_1");
    printf("mixed: cycles per GROUP of one K=32 matrix instruction + k x v_fma_f32\n");
    MIXED(0, true, "K32 + 0 valu");
    MIXED(1, true, "K32 + 1 valu");
    MIXED(2, true, "K32 + 2 valu");
    MIXED(3, true, "K32 + 3 valu");
    MIXED(4, true, "K32 + 4 valu");
    MIXED(6, true, "K32 + 6 valu");
    MIXED(8, true, "K32 + 8 valu");
    printf("mixed: cycles per GROUP of two K=16 matrix instructions + k x v_fma_f32\n");
    MIXED(0, false, "2 x K16 + 0 valu");
    MIXED(2, false, "2 x K16 + 2 valu");
    MIXED(4, false, "2 x K16 + 4 valu");
    MIXED(8, false, "2 x K16 + 8 valu");
    return 0;
}
