// Device management, staging copies, streams and events behind the C ABI.
#include "spc_common.h"
#include <map>
#include <mutex>
#include <unordered_map>

static thread_local char g_err[512] = "";

// ---- device buffer pool behind spc_malloc / spc_free ------------------------------------------
// Cube-sized hipMalloc calls are the slowest thing a pipeline of operators meets on this stack:
// usually ~0.2 ms, but every few calls of >= 4 GiB one takes 0.7 - 3.8 s (measured, MI355X /
// ROCm 7.2: tools/bench_alloc.py), two orders of magnitude more than the kernel that fills the
// buffer.  Freed blocks are therefore kept per device and handed out again for requests of
// (nearly) the same size - operator pipelines allocate the same few cube / map sizes over and
// over.  Semantics stay those of hipMalloc / hipFree: spc_free drains the device before the
// block becomes reusable (hipFree does the same), so no stream ordering is assumed.  The cache
// is bounded (SPC_POOL_MAX_BYTES, default half of the device memory; SPC_POOL=0 turns it off),
// gives everything back when a real hipMalloc runs out of memory, and on spc_pool_trim().
namespace {
constexpr int kMaxDevices = 16;
struct DevicePool {
    std::mutex mu;
    std::multimap<size_t, void*> idle;              // size -> block, ready for reuse
    std::unordered_map<void*, size_t> size_of;      // every block this pool handed out or holds
    size_t idle_bytes = 0, live_bytes = 0, cap = 0;
    bool cap_known = false;
};
DevicePool g_pool[kMaxDevices];

bool pool_enabled() {
    static const bool on = [] { const char* e = getenv("SPC_POOL"); return !(e && atoi(e) == 0); }();
    return on;
}
// SPC_POOL_POISON=1 (debugging): every block spc_malloc returns is filled with 0xFF bytes (NaN as float /
// double, -1 as integer), so that an operator relying on zeroed fresh memory fails the test-suite at once
bool pool_poison() {
    static const bool on = [] { const char* e = getenv("SPC_POOL_POISON"); return e && atoi(e) != 0; }();
    return on;
}
size_t pool_round(size_t bytes) {
    const size_t q = bytes >= (1u << 20) ? (size_t)2 << 20 : 512;     // 2 MiB pages for anything large
    return (bytes + q - 1) / q * q;
}
// callers hold the lock and have the device set
void pool_release_idle(DevicePool& P, size_t keep_bytes) {
    while (P.idle_bytes > keep_bytes && !P.idle.empty()) {
        auto it = std::prev(P.idle.end());                            // largest first
        (void)hipFree(it->second);
        P.size_of.erase(it->second);
        P.idle_bytes -= it->first;
        P.idle.erase(it);
    }
}
}  // namespace

// ---- small host tables as kernel arguments (see spc_common.h) ---------------------------------
namespace {
constexpr int kTableWords = 896;                            // 3584 bytes of the 4 KiB kernarg segment
struct TableChunk { uint32_t w[kTableWords]; };
__global__ __launch_bounds__(256) void table_write_kernel(uint32_t* dst, const TableChunk c, int nwords) {
    for (int i = threadIdx.x; i < nwords; i += 256) dst[i] = c.w[i];
}
__global__ __launch_bounds__(64) void table_write_bytes_kernel(unsigned char* dst, const TableChunk c, int nbytes) {
    const unsigned char* src = reinterpret_cast<const unsigned char*>(c.w);
    for (int i = threadIdx.x; i < nbytes; i += 64) dst[i] = src[i];
}
}  // namespace

hipError_t spc_table_upload(void* d_dst, const void* h_src, size_t bytes, hipStream_t st) {
    const char* src = (const char*)h_src;
    char* dst = (char*)d_dst;
    while (bytes) {
        const size_t n = bytes < sizeof(TableChunk) ? bytes : sizeof(TableChunk);
        TableChunk c;
        memcpy(c.w, src, n);
        if ((n & 3) == 0 && (((uintptr_t)dst) & 3) == 0)
            hipLaunchKernelGGL(table_write_kernel, dim3(1), dim3(256), 0, st, (uint32_t*)dst, c, (int)(n / 4));
        else
            hipLaunchKernelGGL(table_write_bytes_kernel, dim3(1), dim3(64), 0, st, (unsigned char*)dst, c, (int)n);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        src += n; dst += n; bytes -= n;
    }
    return hipSuccess;
}

void spc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int spc_abi_version(void) { return SPC_ABI_VERSION; }
const char* spc_last_error(void) { return g_err; }

size_t spc_moments_workspace_bytes(int64_t nz, int64_t ny, int64_t nx);

size_t spc_workspace_bytes(int kind, int64_t nz, int64_t ny, int64_t nx, int64_t p0, int64_t p1) {
    if (nz <= 0 || ny <= 0 || nx <= 0) return 0;
    switch (kind) {
        case SPC_WS_MOMENTS: return spc_moments_workspace_bytes(nz, ny, nx);
        case SPC_WS_SPECTRAL_CONV: return spc_ws_spectral_conv(nz, ny, nx, p0, false);
        case SPC_WS_SPECTRAL_CONV_MOMENTS: return spc_ws_spectral_conv(nz, ny, nx, p0, true);
        case SPC_WS_SPATIAL_CONV_SEP: return spc_ws_spatial_conv_sep(nz, ny, nx, p0, p1);
        case SPC_WS_SPATIAL_CONV2D: return spc_ws_spatial_conv2d(nz, ny, nx, p0, p1);
        case SPC_WS_RESAMPLE_BILINEAR: return spc_ws_resample_bilinear(p0, p1);
        case SPC_WS_STATS_GLOBAL: case SPC_WS_STATS_PLANES: case SPC_WS_MAP_CONV2D: case SPC_WS_CLIP_OUTSIDE:
            return spc_ws_stats(kind, nz, ny, nx, p0, p1);
        case SPC_WS_PERCENTILE_GLOBAL: return spc_ws_percentile_global();
        case SPC_WS_SPATIAL_CONV_MFMA: return spc_ws_spatial_conv_mfma(nz, ny, nx, p0);    /* p0 = 3: with moments 1 / 2 */
        case SPC_WS_SIGMA_CLIP: return spc_ws_sigma_clip();
        case SPC_WS_RESAMPLE_BILINEAR_LERP: return spc_ws_resample_bilinear_lerp(nz, p0, p1);
        case SPC_WS_STATS_GLOBAL_F64: case SPC_WS_SPECTRAL_CONV_F64: case SPC_WS_SPATIAL_CONV_F64:
            return spc_ws_wide(kind, nz, ny, nx, p0, p1);
    }
    return 0;
}

int spc_device_count(int* count) {
    SPC_REQUIRE(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
    *count = n;
    return SPC_OK;
}

int spc_get_device_info(int device, spc_device_info* info) {
    SPC_REQUIRE(info, "info is NULL");
    hipDeviceProp_t p;
    SPC_HIP(hipGetDeviceProperties(&p, device));
    memset(info, 0, sizeof(*info));
    snprintf(info->name, sizeof(info->name), "%s", p.name);
    snprintf(info->arch, sizeof(info->arch), "%s", p.gcnArchName);
    info->compute_units = p.multiProcessorCount;
    info->wavefront_size = p.warpSize;
    info->clock_khz = p.clockRate;
    SPC_DEVICE(device);
    size_t fr = 0, tot = 0;
    SPC_HIP(hipMemGetInfo(&fr, &tot));
    info->total_mem = (int64_t)tot;
    info->free_mem = (int64_t)fr;
    if (device >= 0 && device < kMaxDevices) {      // idle pool blocks are available to spc_malloc
        std::lock_guard<std::mutex> lk(g_pool[device].mu);
        info->free_mem += (int64_t)g_pool[device].idle_bytes;
    }
    return SPC_OK;
}

int spc_malloc(int device, size_t bytes, void** d_ptr) {
    SPC_REQUIRE(d_ptr, "d_ptr is NULL");
    SPC_DEVICE(device);
    *d_ptr = nullptr;
    if (bytes == 0) return SPC_OK;
    const bool pooled = pool_enabled() && device >= 0 && device < kMaxDevices;
    if (!pooled) {
        hipError_t e = hipMalloc(d_ptr, bytes);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            spc_set_error("hipMalloc(%zu bytes) out of memory on device %d", bytes, device);
            return SPC_ERR_NOMEM;
        }
        SPC_HIP(e);
        return SPC_OK;
    }
    DevicePool& P = g_pool[device];
    const size_t want = pool_round(bytes);
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.idle.lower_bound(want);
    if (it != P.idle.end() && it->first <= want + want / 8) {          // at most 12.5 % larger than asked
        *d_ptr = it->second;
        P.idle_bytes -= it->first;
        P.live_bytes += it->first;
        P.idle.erase(it);
        if (pool_poison()) { SPC_HIP(hipMemset(*d_ptr, 0xFF, bytes)); SPC_HIP(hipDeviceSynchronize()); }
        return SPC_OK;
    }
    hipError_t e = hipMalloc(d_ptr, want);
    if (e == hipErrorOutOfMemory && !P.idle.empty()) {                // give the cache back and try again
        (void)hipGetLastError();
        pool_release_idle(P, 0);
        e = hipMalloc(d_ptr, want);
    }
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        *d_ptr = nullptr;
        spc_set_error("hipMalloc(%zu bytes) out of memory on device %d", bytes, device);
        return SPC_ERR_NOMEM;
    }
    SPC_HIP(e);
    P.size_of[*d_ptr] = want;
    P.live_bytes += want;
    if (pool_poison()) { SPC_HIP(hipMemset(*d_ptr, 0xFF, bytes)); SPC_HIP(hipDeviceSynchronize()); }
    return SPC_OK;
}

int spc_free(int device, void* d_ptr) {
    if (!d_ptr) return SPC_OK;
    SPC_DEVICE(device);
    if (pool_enabled() && device >= 0 && device < kMaxDevices) {
        DevicePool& P = g_pool[device];
        std::unique_lock<std::mutex> lk(P.mu);
        auto it = P.size_of.find(d_ptr);
        if (it != P.size_of.end()) {
            const size_t sz = it->second;
            if (!P.cap_known) {
                size_t fr = 0, tot = 0;
                const char* e = getenv("SPC_POOL_MAX_BYTES");
                if (e) P.cap = (size_t)strtoull(e, nullptr, 10);
                else if (hipMemGetInfo(&fr, &tot) == hipSuccess) P.cap = tot / 2;
                P.cap_known = true;
            }
            lk.unlock();
            SPC_HIP(hipDeviceSynchronize());        // what hipFree guarantees: nothing queued still uses the block
            lk.lock();
            P.live_bytes -= sz;
            if (sz > P.cap) {
                P.size_of.erase(d_ptr);
                SPC_HIP(hipFree(d_ptr));
                return SPC_OK;
            }
            P.idle.emplace(sz, d_ptr);
            P.idle_bytes += sz;
            pool_release_idle(P, P.cap);            // over the bound: the largest idle blocks go back to the driver
            return SPC_OK;
        }
    }
    SPC_HIP(hipFree(d_ptr));                        // not one of ours (any hipMalloc'd pointer is accepted)
    return SPC_OK;
}

int spc_pool_trim(int device) {
    SPC_REQUIRE(device >= 0 && device < kMaxDevices, "device index out of range");
    SPC_DEVICE(device);
    DevicePool& P = g_pool[device];
    std::lock_guard<std::mutex> lk(P.mu);
    pool_release_idle(P, 0);
    return SPC_OK;
}

int spc_pool_stats(int device, int64_t* live_bytes, int64_t* idle_bytes) {
    SPC_REQUIRE(device >= 0 && device < kMaxDevices, "device index out of range");
    DevicePool& P = g_pool[device];
    std::lock_guard<std::mutex> lk(P.mu);
    if (live_bytes) *live_bytes = (int64_t)P.live_bytes;
    if (idle_bytes) *idle_bytes = (int64_t)P.idle_bytes;
    return SPC_OK;
}

int spc_host_alloc(size_t bytes, void** h_ptr) {
    SPC_REQUIRE(h_ptr, "h_ptr is NULL");
    *h_ptr = nullptr;
    if (bytes == 0) return SPC_OK;
    SPC_HIP(hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    return SPC_OK;
}

int spc_host_free(void* h_ptr) {
    if (!h_ptr) return SPC_OK;
    SPC_HIP(hipHostFree(h_ptr));
    return SPC_OK;
}

static int copy_(int device, void* dst, const void* src, size_t bytes, void* stream, hipMemcpyKind kind) {
    if (bytes == 0) return SPC_OK;
    SPC_REQUIRE(dst && src, "memcpy with NULL pointer");
    SPC_DEVICE(device);
    if (stream) {
        SPC_HIP(hipMemcpyAsync(dst, src, bytes, kind, (hipStream_t)stream));
    } else {
        SPC_HIP(hipMemcpy(dst, src, bytes, kind));
        // same insurance as in spc_memcpy3d_h2d: the host buffer may be freed right after we return
        if (kind == hipMemcpyHostToDevice) SPC_HIP(hipDeviceSynchronize());
    }
    return SPC_OK;
}

int spc_memcpy_h2d(int device, void* d_dst, const void* h_src, size_t bytes, void* stream) {
    return copy_(device, d_dst, h_src, bytes, stream, hipMemcpyHostToDevice);
}
int spc_memcpy_d2h(int device, void* h_dst, const void* d_src, size_t bytes, void* stream) {
    return copy_(device, h_dst, d_src, bytes, stream, hipMemcpyDeviceToHost);
}
int spc_memcpy_d2d(int device, void* d_dst, const void* d_src, size_t bytes, void* stream) {
    return copy_(device, d_dst, d_src, bytes, stream, hipMemcpyDeviceToDevice);
}

int spc_memcpy3d_h2d(int device, void* d_dst, size_t d_row_pitch, size_t d_plane_pitch,
                     const void* h_src, size_t h_row_pitch, size_t h_plane_pitch,
                     size_t row_bytes, size_t ny, size_t nz, void* stream) {
    SPC_REQUIRE(d_dst && h_src, "memcpy3d with NULL pointer");
    SPC_REQUIRE(d_row_pitch >= row_bytes && h_row_pitch >= row_bytes, "row pitch smaller than row");
    SPC_DEVICE(device);
    for (size_t z = 0; z < nz; ++z) {
        char* d = (char*)d_dst + z * d_plane_pitch;
        const char* s = (const char*)h_src + z * h_plane_pitch;
        if (stream) {
            SPC_HIP(hipMemcpy2DAsync(d, d_row_pitch, s, h_row_pitch, row_bytes, ny,
                                     hipMemcpyHostToDevice, (hipStream_t)stream));
        } else {
            SPC_HIP(hipMemcpy2D(d, d_row_pitch, s, h_row_pitch, row_bytes, ny, hipMemcpyHostToDevice));
        }
    }
    // synchronous form: the caller may free h_src as soon as this returns.  An explicit drain, because
    // with many host threads issuing pageable 2-D copies a "Memory access fault by GPU" on a host address
    // was seen (bench.py's input staging, ~1 run in 5) - a copy still reading pages the caller had freed.
    if (!stream) SPC_HIP(hipDeviceSynchronize());
    return SPC_OK;
}

int spc_memcpy3d_d2d(int device, void* d_dst, size_t dst_row_pitch, size_t dst_plane_pitch,
                     const void* d_src, size_t src_row_pitch, size_t src_plane_pitch,
                     size_t row_bytes, size_t ny, size_t nz, void* stream) {
    if (row_bytes == 0 || ny == 0 || nz == 0) return SPC_OK;
    SPC_REQUIRE(d_dst && d_src, "memcpy3d with NULL pointer");
    SPC_REQUIRE(dst_row_pitch >= row_bytes && src_row_pitch >= row_bytes, "row pitch smaller than row");
    SPC_DEVICE(device);
    hipStream_t st = (hipStream_t)stream;
    if (dst_row_pitch == row_bytes && src_row_pitch == row_bytes) {
        // whole (ny x row) blocks are contiguous on both sides: ONE 2-D copy with the planes as rows
        SPC_HIP(hipMemcpy2DAsync(d_dst, dst_plane_pitch, d_src, src_plane_pitch, row_bytes * ny, nz,
                                 hipMemcpyDeviceToDevice, st));
    } else {
        for (size_t z = 0; z < nz; ++z)
            SPC_HIP(hipMemcpy2DAsync((char*)d_dst + z * dst_plane_pitch, dst_row_pitch,
                                     (const char*)d_src + z * src_plane_pitch, src_row_pitch, row_bytes, ny,
                                     hipMemcpyDeviceToDevice, st));
    }
    if (!stream) SPC_HIP(hipStreamSynchronize(st));
    return SPC_OK;
}

int spc_memset(int device, void* d_ptr, int value, size_t bytes, void* stream) {
    if (bytes == 0) return SPC_OK;
    SPC_DEVICE(device);
    SPC_HIP(hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)stream));
    return SPC_OK;
}

int spc_stream_create(int device, void** stream) {
    SPC_REQUIRE(stream, "stream is NULL");
    SPC_DEVICE(device);
    hipStream_t s;
    SPC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return SPC_OK;
}
int spc_stream_destroy(int device, void* stream) {
    if (!stream) return SPC_OK;
    SPC_DEVICE(device);
    SPC_HIP(hipStreamDestroy((hipStream_t)stream));
    return SPC_OK;
}
int spc_stream_sync(int device, void* stream) {
    SPC_DEVICE(device);
    SPC_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SPC_OK;
}
int spc_device_sync(int device) {
    SPC_DEVICE(device);
    SPC_HIP(hipDeviceSynchronize());
    return SPC_OK;
}
int spc_event_create(int device, void** event) {
    SPC_REQUIRE(event, "event is NULL");
    SPC_DEVICE(device);
    hipEvent_t e;
    SPC_HIP(hipEventCreate(&e));
    *event = (void*)e;
    return SPC_OK;
}
int spc_event_destroy(int device, void* event) {
    if (!event) return SPC_OK;
    SPC_DEVICE(device);
    SPC_HIP(hipEventDestroy((hipEvent_t)event));
    return SPC_OK;
}
int spc_event_record(int device, void* event, void* stream) {
    SPC_DEVICE(device);
    SPC_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return SPC_OK;
}
int spc_event_sync(int device, void* event) {
    SPC_DEVICE(device);
    SPC_HIP(hipEventSynchronize((hipEvent_t)event));
    return SPC_OK;
}
int spc_stream_wait_event(int device, void* stream, void* event) {
    SPC_DEVICE(device);
    SPC_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return SPC_OK;
}
int spc_event_elapsed_ms(int device, void* start, void* stop, float* ms) {
    SPC_REQUIRE(ms, "ms is NULL");
    SPC_DEVICE(device);
    SPC_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SPC_OK;
}

}  // extern "C"
