// checks the inline-asm building blocks of spc_spatial_split.hip on the device: fp16 hi / lo split, the wave maximum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ void split_pair(float a, float b, float s, unsigned& hi, unsigned& lo) {
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(b), "s"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "s"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "s"(s), "v"(h));
    hi = h; lo = l;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__global__ void k(const float* in, unsigned* out, float s, int hot) {
    const int t = threadIdx.x;
    unsigned hi, lo;
    split_pair(in[2 * t], in[2 * t + 1], s, hi, lo);
    out[3 * t] = hi; out[3 * t + 1] = lo;
    unsigned v = (unsigned)(t == hot ? 1000 + t : t);
    out[3 * t + 2] = wave_max_u32(v);
}
static float h2f(unsigned short h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? ldexpf((float)m, -24) : e == 31 ? (m ? NAN : INFINITY) : ldexpf((float)(m | 1024), e - 25);
    return s ? -v : v;
}
int main() {
    std::vector<float> h(128);
    for (int i = 0; i < 128; ++i) h[i] = (float)(sin(i * 1.7) * 3.0 + (i % 5 == 0 ? 1e-3 : 0.0));
    float* d; unsigned* o;
    hipMalloc(&d, 512); hipMalloc(&o, 64 * 12);
    hipMemcpy(d, h.data(), 512, hipMemcpyHostToDevice);
    const float s = 4096.f;
    int bad = 0;
    for (int hot : {0, 5, 15, 16, 31, 32, 47, 48, 63}) {
        k<<<1, 64>>>(d, o, s, hot);
        std::vector<unsigned> r(192);
        hipMemcpy(r.data(), o, 768, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int t = 0; t < 64; ++t) {
            for (int q = 0; q < 2; ++q) {
                const float x = h[2 * t + q] * s;
                const float hi = h2f((r[3 * t] >> (16 * q)) & 0xffff), lo = h2f((r[3 * t + 1] >> (16 * q)) & 0xffff);
                const double e = fabs((double)hi + lo - x) / fabs(x);
                if (e > worst) worst = e;
                if (hot == 0 && t < 3) printf("  x %.8g hi %.8g lo %.8g rel err %.3g\n", x, hi, lo, e);
            }
            if (r[3 * t + 2] != (unsigned)(1000 + hot)) { if (bad < 5) printf("  wave max: lane %d got %u, expected %d\n", t, r[3 * t + 2], 1000 + hot); ++bad; }
        }
        printf("hot lane %2d: split worst relative error %.3g (fp16 hi + lo: ~2.4e-7), wave max %s\n", hot, worst, bad ? "WRONG" : "ok");
    }
    return bad ? 1 : 0;
}
