// Microbenchmark: "march over z" streaming pattern of the spectral kernels:
// each lane owns VEC consecutive x of a (nz, ny, nx) cube and walks all planes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T, int U, int STORE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, long nz, long plane, float* sink) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int V = sizeof(T) / 4;
    if (g * V >= plane) return;
    const T* p = (const T*)(in + g * V);
    T* q = (T*)(out + g * V);
    T acc{};
    for (long z = 0; z + U <= nz; z += U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const T*)((const float*)p + (z + u) * plane));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (STORE == 1) *(T*)((float*)q + (z + u) * plane) = v[u] * 1.5f;
            else if (STORE == 2) __builtin_nontemporal_store(v[u] * 1.5f, (T*)((float*)q + (z + u) * plane));
            else acc += v[u];
        }
    }
    if (!STORE) { float s = 0; const float* a = (const float*)&acc; for (int i = 0; i < V; ++i) s += a[i]; if (s == 12345.f) sink[0] = s; }
}
template <typename T, int U, int STORE>
void run(const char* name, const float* in, float* out, long nz, long plane, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    constexpr int V = sizeof(T) / 4;
    dim3 grid((unsigned)((plane / V + 255) / 256));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<T, U, STORE><<<grid, 256>>>(in, out, nz, plane, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double bytes = (double)nz * plane * 4 * (STORE ? 2 : 1);
    printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9);
}
int main() {
    const long nz = 1024, plane = 1024 * 1024;
    float *in, *out, *sink; hipMalloc(&in, nz * plane * 4); hipMalloc(&out, nz * plane * 4); hipMalloc(&sink, 64);
    hipMemset(in, 0, nz * plane * 4);
    run<float, 8, false>("read  4B/lane U8", in, out, nz, plane, sink);
    run<float, 33, false>("read  4B/lane U33", in, out, nz, plane, sink);
    run<f2, 8, false>("read  8B/lane U8", in, out, nz, plane, sink);
    run<f2, 33, false>("read  8B/lane U33", in, out, nz, plane, sink);
    run<f4, 8, false>("read 16B/lane U8", in, out, nz, plane, sink);
    run<float, 8, true>("copy  4B/lane U8", in, out, nz, plane, sink);
    run<float, 33, true>("copy  4B/lane U33", in, out, nz, plane, sink);
    run<f2, 8, true>("copy  8B/lane U8", in, out, nz, plane, sink);
    run<f2, 33, true>("copy  8B/lane U33", in, out, nz, plane, sink);
    run<f4, 8, true>("copy 16B/lane U8", in, out, nz, plane, sink);
    run<float, 8, 2>("copy-nt  4B/lane U8", in, out, nz, plane, sink);
    run<f2, 8, 2>("copy-nt  8B/lane U8", in, out, nz, plane, sink);
    run<f2, 33, 2>("copy-nt  8B/lane U33", in, out, nz, plane, sink);
    run<f4, 8, 2>("copy-nt 16B/lane U8", in, out, nz, plane, sink);
    run<f4, 2, 2>("copy-nt 16B/lane U2", in, out, nz, plane, sink);
    return 0;
}
