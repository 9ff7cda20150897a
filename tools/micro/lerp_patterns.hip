// build + run: hipcc --offload-arch=gfx950 -O3 -o tools/micro/lerp_patterns tools/micro/lerp_patterns.hip && gpurun -- tools/micro/lerp_patterns
// Round 5: does the access ORDER that lets a textbook copy reach 6.3 - 6.6 TB/s (tools/micro/copy_ceiling.hip: short-lived
// blocks, one 16-byte load + store per lane) carry over to the cube -> cube operators?  spectral_interpolate at C5 reads
// 2048 planes and writes 4096 (8 + 16 GiB at 1024^2): the product kernel marches every lane along z (each input plane read
// once, 5.1 - 5.6 ms = 4.3 - 4.7 TB/s).  Forms measured here, out[j] = (1 - t) in[j / 2] + t in[j / 2 + 1]:
//   march<U>     : the product's pattern - a lane owns 4 columns and walks z, U output planes per iteration
//   tiles<UO>    : short-lived blocks - a block owns 256 x 16 B of one row segment and UO consecutive output planes
//                  (UO / 2 + 1 input planes: the boundary plane is read by two blocks), plane-major or tile-major order
// Rates are algorithmic bytes (each input plane once + each output plane once) / time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(256) void march(const float* __restrict__ in, float* __restrict__ out, long nz_in, long plane) {
    const long g = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (g >= plane) return;
    f4 prev = *(const f4*)(in + g);
    for (long k = 0; k + U <= nz_in - 1; k += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const f4*)(in + (k + u + 1) * plane + g));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f4 a = u ? v[u - 1] : prev, b = v[u];
            const f4 o0 = a, o1 = 0.5f * a + 0.5f * b;
            if (NT) { __builtin_nontemporal_store(o0, (f4*)(out + (2 * (k + u)) * plane + g)); __builtin_nontemporal_store(o1, (f4*)(out + (2 * (k + u) + 1) * plane + g)); }
            else { *(f4*)(out + (2 * (k + u)) * plane + g) = o0; *(f4*)(out + (2 * (k + u) + 1) * plane + g) = o1; }
        }
        prev = v[U - 1];
    }
}

// block b -> (tile, z group): PM (plane-major): consecutive blocks walk the tiles of one z group; else tile-major
template <int UO, int NT, int PM>
__global__ __launch_bounds__(256) void tiles(const float* __restrict__ in, float* __restrict__ out, long nz_in, long plane, long ntiles, long ngroups) {
    const long b = blockIdx.x;
    const long tile = PM ? b % ntiles : b / ngroups, grp = PM ? b / ntiles : b % ngroups;
    const long g = (tile * 256 + threadIdx.x) * 4;
    const long k0 = grp * (UO / 2);
    if (k0 + UO / 2 > nz_in - 1) return;
    f4 v[UO / 2 + 1];
#pragma unroll
    for (int u = 0; u <= UO / 2; ++u) v[u] = NT ? __builtin_nontemporal_load((const f4*)(in + (k0 + u) * plane + g)) : *(const f4*)(in + (k0 + u) * plane + g);
#pragma unroll
    for (int u = 0; u < UO / 2; ++u) {
        const f4 o0 = v[u], o1 = 0.5f * v[u] + 0.5f * v[u + 1];
        if (NT) { __builtin_nontemporal_store(o0, (f4*)(out + (2 * (k0 + u)) * plane + g)); __builtin_nontemporal_store(o1, (f4*)(out + (2 * (k0 + u) + 1) * plane + g)); }
        else { *(f4*)(out + (2 * (k0 + u)) * plane + g) = o0; *(f4*)(out + (2 * (k0 + u) + 1) * plane + g) = o1; }
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
static void timeit(const char* name, double bytes, F f) {
    std::vector<float> t;
    float ms = 0;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0);
        f();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("%-72s median %.3f ms (min %.3f)  %.2f TB/s algorithmic\n", name, t[t.size() / 2], t[0], bytes / t[t.size() / 2] / 1e9);
    fflush(stdout);
}

int main() {
    const long nz_in = 2049, ny = 1024, nx = 1024, plane = ny * nx, nz_out = 2 * (nz_in - 1);
    float *in, *out;
    if (hipMalloc(&in, nz_in * plane * 4) != hipSuccess || hipMalloc(&out, nz_out * plane * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    fill_random<<<4096, 256>>>(in, nz_in * plane, 77u);
    (void)hipMemset(out, 0, nz_out * plane * 4);
    (void)hipDeviceSynchronize();
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double alg = (double)(nz_in + nz_out) * plane * 4;
    const long ntiles = plane / 1024;
    char name[160];
#define MARCH(U, NT) snprintf(name, 160, "march U=%d %s", U, NT ? "nt" : "plain"); timeit(name, alg, [&] { march<U, NT><<<(unsigned)ntiles, 256>>>(in, out, nz_in, plane); })
    MARCH(2, 1); MARCH(4, 1); MARCH(8, 1); MARCH(8, 0);
#define TILES(UO, NT, PM) { const long ng = (nz_in - 1) / (UO / 2); snprintf(name, 160, "tiles UO=%d %s %s (input planes x%.3f)", UO, NT ? "nt" : "plain", PM ? "plane-major" : "tile-major", (double)(UO / 2 + 1) / (UO / 2)); \
        timeit(name, alg, [&] { tiles<UO, NT, PM><<<(unsigned)(ntiles * ng), 256>>>(in, out, nz_in, plane, ntiles, ng); }); }
    TILES(2, 1, 1); TILES(4, 1, 1); TILES(8, 1, 1); TILES(16, 1, 1); TILES(32, 1, 1);
    TILES(4, 0, 1); TILES(8, 0, 1); TILES(16, 0, 1);
    TILES(8, 1, 0); TILES(16, 1, 0); TILES(32, 1, 0); TILES(16, 0, 0);
    return 0;
}
