// build + run: hipcc --offload-arch=gfx950 -O3 -o write_fill write_fill.hip && rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./write_fill
// Round 6: does a store that covers whole 128-byte lines make the L2 FETCH the line first?  A 1 GiB buffer written once by
// every kernel; FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes) say what moved.  Patterns: B bytes per lane,
// lanes consecutive (a wave writes 64 B contiguous bytes), and the sigma-clip kernel's pattern: 32 lanes x 4 bytes = one
// 128-byte run per half wave, the two halves 4 MiB apart.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void w4(float* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.f; }
__global__ void w8(f2* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = f2{1.f, 2.f}; }
__global__ void w16(f4* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = f4{1.f, 2.f, 3.f, 4.f}; }
// plane-strided: lane (r = t % 32, j = t / 32) writes p[(j + 8 i) * plane + x0 + r], i = 0 .. : 128-byte runs, 8 planes per block-instruction
__global__ void wclip(float* p, size_t plane, int nz) {
    const int r = threadIdx.x % 32, j = threadIdx.x / 32;
    const size_t x0 = (size_t)blockIdx.x * 32;
    for (int z = j; z < nz; z += 8) p[(size_t)z * plane + x0 + r] = 1.f;
}
// the same with 16 lanes x 4 bytes = 64-byte runs (the dense-mask shape)
__global__ void wclip16(float* p, size_t plane, int nz) {
    const int r = threadIdx.x % 16, j = threadIdx.x / 16;
    const size_t x0 = (size_t)blockIdx.x * 16;
    for (int z = j; z < nz; z += 16) p[(size_t)z * plane + x0 + r] = 1.f;
}
// 8 bytes per lane, rows marched by a block of 256 lanes (the float64 ring kernels' stores): 2 KiB per block and row
__global__ void wrow8(f2* p, size_t row_elems, int nrows) {
    const size_t x = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (int y = 0; y < nrows; ++y) p[(size_t)y * row_elems + x] = f2{1.f, 2.f};
}
int main() {
    const size_t bytes = (size_t)1 << 30;
    void* d; hipMalloc(&d, bytes);
    hipMemset(d, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(w4, dim3(4096), dim3(256), 0, 0, (float*)d, bytes / 4);
        hipLaunchKernelGGL(w8, dim3(4096), dim3(256), 0, 0, (f2*)d, bytes / 8);
        hipLaunchKernelGGL(w16, dim3(4096), dim3(256), 0, 0, (f4*)d, bytes / 16);
        hipLaunchKernelGGL(wclip, dim3(1024 * 1024 / 32), dim3(256), 0, 0, (float*)d, (size_t)1024 * 1024, 256);
        hipLaunchKernelGGL(wclip16, dim3(1024 * 1024 / 16), dim3(256), 0, 0, (float*)d, (size_t)1024 * 1024, 256);
        hipLaunchKernelGGL(wrow8, dim3(128 * 1024 / 256 * 8), dim3(256), 0, 0, (f2*)d, (size_t)128 * 1024 * 8, 128);
    }
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
