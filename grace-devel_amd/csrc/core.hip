// Status reporting, device-memory helpers and the workspace of libgrace_hip.so.
#include "common.hpp"

#include <cstring>

namespace grace_hip {

static thread_local char g_last_error[512] = "no error";

grace_status set_error(grace_status code, const char* file, int line, const char* what)
{
    const char* base = std::strrchr(file, '/');
    std::snprintf(g_last_error, sizeof(g_last_error), "%s:%d: %s", base ? base + 1 : file,
                  line, what);
    return code;
}

char* Workspace::base_ = nullptr;
size_t Workspace::capacity_ = 0;
size_t Workspace::used_ = 0;

// The workspace (like the status word, the prepared scene and the tuning knobs) is
// process-global: one device, one stream at a time -- the shape of one process per GPU.  A call
// made with another device current would get pointers into the first device's memory: refuse it.
static int g_owner_device = -1;

grace_status Workspace::reserve(size_t bytes)
{
    int dev = -1;
    GRACE_TRY_HIP(hipGetDevice(&dev));
    if (g_owner_device < 0) g_owner_device = dev;
    if (dev != g_owner_device)
        return set_error(GRACE_INVALID_ARGUMENT, __FILE__, __LINE__,
                         "libgrace_hip.so serves one device per process (its workspace lives on the "
                         "device of the first call): run one process per GPU");
    if (bytes <= capacity_) return GRACE_OK;
    // Grow with headroom so that steady-state calls never allocate.
    size_t want = bytes + bytes / 4 + (size_t(1) << 20);
    if (base_) {
        GRACE_TRY_HIP(hipDeviceSynchronize());
        GRACE_TRY_HIP(hipFree(base_));
        base_ = nullptr;
        capacity_ = 0;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess)
        return set_error(GRACE_OUT_OF_MEMORY, __FILE__, __LINE__, hipGetErrorString(e));
    base_ = static_cast<char*>(p);
    capacity_ = want;
    return GRACE_OK;
}

grace_status Workspace::begin(size_t bytes, hipStream_t stream)
{
    GRACE_TRY(reserve(bytes));
    // Frames alias: order this one behind the previous frame's stream when the stream changes.
    static hipStream_t last_stream = nullptr;
    static bool have_last = false;
    static hipEvent_t fence = nullptr;
    if (have_last && stream != last_stream) {
        if (!fence) GRACE_TRY_HIP(hipEventCreateWithFlags(&fence, hipEventDisableTiming));
        GRACE_TRY_HIP(hipEventRecord(fence, last_stream));
        GRACE_TRY_HIP(hipStreamWaitEvent(stream, fence, 0));
    }
    last_stream = stream;
    have_last = true;
    used_ = 0;
    return GRACE_OK;
}

grace_status Workspace::release()
{
    if (base_) {
        GRACE_TRY_HIP(hipDeviceSynchronize());
        GRACE_TRY_HIP(hipFree(base_));
    }
    base_ = nullptr;
    capacity_ = used_ = 0;
    g_owner_device = -1;      // the next call may adopt another device
    return GRACE_OK;
}

} // namespace grace_hip

using namespace grace_hip;

extern "C" {

int grace_version(void) { return 100; }

const char* grace_last_error(void) { return g_last_error; }

grace_status grace_device_malloc(void** d_ptr, size_t bytes)
{
    GRACE_REQUIRE(d_ptr != nullptr, "null output pointer");
    *d_ptr = nullptr;
    if (bytes == 0) return GRACE_OK;
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e != hipSuccess)
        return set_error(GRACE_OUT_OF_MEMORY, __FILE__, __LINE__, hipGetErrorString(e));
    return GRACE_OK;
}

grace_status grace_device_free(void* d_ptr)
{
    if (d_ptr) GRACE_TRY_HIP(hipFree(d_ptr));
    return GRACE_OK;
}

grace_status grace_memcpy_htod(void* d_dst, const void* h_src, size_t bytes, grace_stream s)
{
    if (bytes == 0) return GRACE_OK;
    GRACE_TRY_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, as_stream(s)));
    GRACE_TRY_HIP(hipStreamSynchronize(as_stream(s)));
    return GRACE_OK;
}

grace_status grace_memcpy_dtoh(void* h_dst, const void* d_src, size_t bytes, grace_stream s)
{
    if (bytes == 0) return GRACE_OK;
    GRACE_TRY_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, as_stream(s)));
    GRACE_TRY_HIP(hipStreamSynchronize(as_stream(s)));
    return GRACE_OK;
}

grace_status grace_memcpy_dtod(void* d_dst, const void* d_src, size_t bytes, grace_stream s)
{
    if (bytes == 0) return GRACE_OK;
    GRACE_TRY_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, as_stream(s)));
    return GRACE_OK;
}

grace_status grace_memset(void* d_dst, int byte, size_t bytes, grace_stream s)
{
    if (bytes == 0) return GRACE_OK;
    GRACE_TRY_HIP(hipMemsetAsync(d_dst, byte, bytes, as_stream(s)));
    return GRACE_OK;
}

grace_status grace_stream_synchronize(grace_stream s)
{
    GRACE_TRY_HIP(hipStreamSynchronize(as_stream(s)));
    return GRACE_OK;
}

grace_status grace_workspace_reserve(size_t bytes) { return Workspace::reserve(bytes); }

grace_status grace_workspace_release(void) { return Workspace::release(); }

} // extern "C"
