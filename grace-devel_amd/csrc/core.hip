// Status reporting, device-memory helpers and the workspace of libgrace_hip.so.
#include "common.hpp"

#include <cstring>
#include <mutex>

namespace grace_hip {

static thread_local char g_last_error[512] = "no error";

grace_status set_error(grace_status code, const char* file, int line, const char* what)
{
    const char* base = std::strrchr(file, '/');
    std::snprintf(g_last_error, sizeof(g_last_error), "%s:%d: %s", base ? base + 1 : file,
                  line, what);
    return code;
}

// ---- contexts ----------------------------------------------------------------------------------
constexpr int MAX_DEVICES = 64;
static Context* g_default_context[MAX_DEVICES] = {};
static std::mutex g_context_mutex;                    // guards creation / destruction only
static thread_local Context* t_current = nullptr;     // the thread's explicit context, if any
static thread_local Context* t_frame = nullptr;       // context of the thread's open frame

grace_status (*g_trace_state_destroy)(Context&) = nullptr;

grace_status current_context(Context** out)
{
    int dev = -1;
    GRACE_TRY_HIP(hipGetDevice(&dev));
    if (t_current) {
        if (t_current->device != dev)
            return set_error(GRACE_INVALID_ARGUMENT, __FILE__, __LINE__,
                             "the calling thread's grace context belongs to another device than the "
                             "current one: hipSetDevice() to its device, or make a context of this "
                             "device current (grace_context_set_current)");
        *out = t_current;
        return GRACE_OK;
    }
    if (dev < 0 || dev >= MAX_DEVICES)
        return set_error(GRACE_INVALID_ARGUMENT, __FILE__, __LINE__, "device ordinal out of range");
    Context* c = g_default_context[dev];
    if (!c) {
        std::lock_guard<std::mutex> lock(g_context_mutex);
        c = g_default_context[dev];
        if (!c) {
            c = new Context();
            c->device = dev;
            g_default_context[dev] = c;
        }
    }
    *out = c;
    return GRACE_OK;
}

// Frees everything a context owns on its device (which must be current).
static grace_status context_clear(Context& c)
{
    if (g_trace_state_destroy) GRACE_TRY(g_trace_state_destroy(c));
    if (c.ws_base) {
        GRACE_TRY_HIP(hipDeviceSynchronize());
        GRACE_TRY_HIP(hipFree(c.ws_base));
    }
    c.ws_base = nullptr;
    c.ws_capacity = c.ws_used = 0;
    if (c.ws_fence) GRACE_TRY_HIP(hipEventDestroy(c.ws_fence));
    c.ws_fence = nullptr;
    c.ws_fence_valid = false;
    c.ws_fence_stream = nullptr;
    for (auto& e : c.phase_events) {
        if (e) GRACE_TRY_HIP(hipEventDestroy(e));
        e = nullptr;
    }
    c.phase_valid = false;
    if (c.side_stream) {
        GRACE_TRY_HIP(hipStreamSynchronize(c.side_stream));
        GRACE_TRY_HIP(hipStreamDestroy(c.side_stream));
        GRACE_TRY_HIP(hipEventDestroy(c.side_fork_ev));
        GRACE_TRY_HIP(hipEventDestroy(c.side_join_ev));
    }
    c.side_stream = nullptr;
    c.side_fork_ev = c.side_join_ev = nullptr;
    if (c.sort_overflow_host) (void)hipHostFree(const_cast<uint32_t*>(c.sort_overflow_host));
    c.sort_overflow_host = nullptr;
    c.sort_overflow_dev = nullptr;
    c.sort_hint_skips = 0;
    return GRACE_OK;
}

uint32_t* sort_overflow_word(Context** ctx_out)
{
    Context* c = Workspace::frame_context();
    if (ctx_out) *ctx_out = c;
    if (!c) return nullptr;
    if (!c->sort_overflow_host) {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipHostFree(h);
            return nullptr;
        }
        *static_cast<uint32_t*>(h) = 0u;
        c->sort_overflow_host = static_cast<volatile uint32_t*>(h);
        c->sort_overflow_dev = static_cast<uint32_t*>(d);
    }
    return c->sort_overflow_dev;
}

grace_status side_fork(hipStream_t stream, hipStream_t* side)
{
    Context* c = Workspace::frame_context();
    GRACE_REQUIRE(c && c->ws_frame_open, "side_fork: no open frame");
    if (!c->side_stream) {
        hipStream_t s = nullptr;
        GRACE_TRY_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess
            || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
            if (a) (void)hipEventDestroy(a);
            (void)hipStreamDestroy(s);
            return set_error(GRACE_HIP_ERROR, __FILE__, __LINE__, "side_fork: event creation failed");
        }
        c->side_stream = s; c->side_fork_ev = a; c->side_join_ev = b;
    }
    GRACE_TRY_HIP(hipEventRecord(c->side_fork_ev, stream));
    GRACE_TRY_HIP(hipStreamWaitEvent(c->side_stream, c->side_fork_ev, 0));
    *side = c->side_stream;
    return GRACE_OK;
}

grace_status side_join(hipStream_t stream)
{
    Context* c = Workspace::frame_context();
    GRACE_REQUIRE(c && c->side_stream, "side_join: no side stream");
    GRACE_TRY_HIP(hipEventRecord(c->side_join_ev, c->side_stream));
    GRACE_TRY_HIP(hipStreamWaitEvent(stream, c->side_join_ev, 0));
    return GRACE_OK;
}

Context* Workspace::frame_context() { return t_frame; }

grace_status Workspace::reserve(size_t bytes)
{
    Context* c = nullptr;
    GRACE_TRY(current_context(&c));
    if (bytes <= c->ws_capacity) return GRACE_OK;
    // Grow with headroom so that steady-state calls never allocate.
    size_t want = bytes + bytes / 4 + (size_t(1) << 20);
    if (c->ws_base) {
        GRACE_TRY_HIP(hipDeviceSynchronize());
        GRACE_TRY_HIP(hipFree(c->ws_base));
        c->ws_base = nullptr;
        c->ws_capacity = 0;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess)
        return set_error(GRACE_OUT_OF_MEMORY, __FILE__, __LINE__, hipGetErrorString(e));
    c->ws_base = static_cast<char*>(p);
    c->ws_capacity = want;
    return GRACE_OK;
}

grace_status Workspace::begin(size_t bytes, hipStream_t stream)
{
    Context* c = nullptr;
    GRACE_TRY(current_context(&c));
    GRACE_TRY(reserve(bytes));
    // Frames alias: a frame on another stream than the previous one waits for the event that frame
    // recorded on ITS stream when it closed (Workspace::end).  The previous stream's handle is only
    // compared, never used: the caller may have destroyed it.
    if (c->ws_fence_valid && stream != c->ws_fence_stream)
        GRACE_TRY_HIP(hipStreamWaitEvent(stream, c->ws_fence, 0));
    c->ws_used = 0;
    c->ws_frame_open = true;
    t_frame = c;
    return GRACE_OK;
}

void Workspace::end(hipStream_t stream)
{
    Context* c = t_frame;
    if (!c || !c->ws_frame_open) return;
    c->ws_frame_open = false;
    if (!c->ws_fence && hipEventCreateWithFlags(&c->ws_fence, hipEventDisableTiming) != hipSuccess) {
        c->ws_fence = nullptr;
    }
    if (c->ws_fence && hipEventRecord(c->ws_fence, stream) == hipSuccess) {
        c->ws_fence_valid = true;
        c->ws_fence_stream = stream;
    } else {
        // no fence: make the next frame safe the blunt way
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
        c->ws_fence_valid = false;
    }
}

grace_status Workspace::release()
{
    Context* c = nullptr;
    GRACE_TRY(current_context(&c));
    if (c->ws_base) {
        GRACE_TRY_HIP(hipDeviceSynchronize());
        GRACE_TRY_HIP(hipFree(c->ws_base));
    }
    c->ws_base = nullptr;
    c->ws_capacity = c->ws_used = 0;
    c->ws_fence_valid = false;
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

grace_status grace_context_create(grace_context* ctx)
{
    GRACE_REQUIRE(ctx, "context_create: null output");
    *ctx = nullptr;
    int dev = -1;
    GRACE_TRY_HIP(hipGetDevice(&dev));
    Context* c = new Context();
    c->device = dev;
    *ctx = reinterpret_cast<grace_context>(c);
    return GRACE_OK;
}

grace_status grace_context_destroy(grace_context ctx)
{
    if (!ctx) return GRACE_OK;
    Context* c = reinterpret_cast<Context*>(ctx);
    for (Context* d : g_default_context)
        GRACE_REQUIRE(d != c, "context_destroy: a device's default context cannot be destroyed");
    int dev = -1;
    GRACE_TRY_HIP(hipGetDevice(&dev));
    if (dev != c->device) GRACE_TRY_HIP(hipSetDevice(c->device));
    grace_status st = context_clear(*c);
    if (dev != c->device) GRACE_TRY_HIP(hipSetDevice(dev));
    GRACE_TRY(st);
    if (t_current == c) t_current = nullptr;
    if (t_frame == c) t_frame = nullptr;
    delete c;
    return GRACE_OK;
}

grace_status grace_context_set_current(grace_context ctx)
{
    t_current = reinterpret_cast<Context*>(ctx);
    return GRACE_OK;
}

grace_status grace_context_get_current(grace_context* ctx)
{
    GRACE_REQUIRE(ctx, "context_get_current: null output");
    Context* c = nullptr;
    GRACE_TRY(current_context(&c));
    *ctx = reinterpret_cast<grace_context>(c);
    return GRACE_OK;
}

} // extern "C"
