// Device management, staging copies, streams and events behind the C ABI.
#include "spc_common.h"

static thread_local char g_err[512] = "";

void spc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int spc_abi_version(void) { return SPC_ABI_VERSION; }
const char* spc_last_error(void) { return g_err; }

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
    return SPC_OK;
}

int spc_malloc(int device, size_t bytes, void** d_ptr) {
    SPC_REQUIRE(d_ptr, "d_ptr is NULL");
    SPC_DEVICE(device);
    *d_ptr = nullptr;
    if (bytes == 0) return SPC_OK;
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        spc_set_error("hipMalloc(%zu bytes) out of memory on device %d", bytes, device);
        return SPC_ERR_NOMEM;
    }
    SPC_HIP(e);
    return SPC_OK;
}

int spc_free(int device, void* d_ptr) {
    if (!d_ptr) return SPC_OK;
    SPC_DEVICE(device);
    SPC_HIP(hipFree(d_ptr));
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
