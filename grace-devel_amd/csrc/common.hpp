// Shared host-side plumbing of libgrace_hip.so: status/error reporting, the grow-only
// device workspace, launch checks.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "grace_hip.h"

namespace grace_hip {

grace_status set_error(grace_status code, const char* file, int line, const char* what);

// The reference checks every API call and peeks at the launch error after every kernel
// (include/grace/error.h:35-64); a failure here becomes a status instead of exit().
#define GRACE_TRY_HIP(expr)                                                              \
    do {                                                                                 \
        hipError_t grace_e_ = (expr);                                                    \
        if (grace_e_ != hipSuccess)                                                      \
            return ::grace_hip::set_error(GRACE_HIP_ERROR, __FILE__, __LINE__,           \
                                          hipGetErrorString(grace_e_));                  \
    } while (0)

#define GRACE_TRY(expr)                                                                  \
    do {                                                                                 \
        grace_status grace_s_ = (expr);                                                  \
        if (grace_s_ != GRACE_OK) return grace_s_;                                       \
    } while (0)

#define GRACE_CHECK_LAUNCH() GRACE_TRY_HIP(hipGetLastError())

#define GRACE_REQUIRE(cond, msg)                                                         \
    do {                                                                                 \
        if (!(cond))                                                                     \
            return ::grace_hip::set_error(GRACE_INVALID_ARGUMENT, __FILE__, __LINE__, msg); \
    } while (0)

inline hipStream_t as_stream(grace_stream s) { return reinterpret_cast<hipStream_t>(s); }

// ---- library state -----------------------------------------------------------------------------
// Everything the library keeps between calls lives in a Context: the grow-only workspace, the
// traversal's status word / cached scene and ray order / tuning knobs / timing events
// (TraceState, csrc/trace_state.hpp) and the build's phase events.  A context belongs to ONE
// device.  Every device has a default context, created on first use; a thread that wants state of
// its own -- several host threads driving one GPU, or one thread per GPU of a node in a single
// process (the reference's ncclCommInitAll shape) -- creates contexts with
// grace_context_create and makes one current for itself (grace_context_set_current, thread-local).
// A context serves one call at a time: two threads may run concurrently iff their current
// contexts differ.
struct TraceState;
struct Context {
    int device = -1;
    // workspace: one device buffer, bump-allocated per call ("frame")
    char* ws_base = nullptr;
    size_t ws_capacity = 0, ws_used = 0;
    // recorded on a frame's own stream when the frame closes; a frame opened on ANOTHER stream
    // waits for it (frames alias).  No stream handle is ever kept: a caller may destroy its stream
    // between calls.
    hipEvent_t ws_fence = nullptr;
    bool ws_fence_valid = false;
    hipStream_t ws_fence_stream = nullptr;   // compared only, never dereferenced
    bool ws_frame_open = false;
    // ALBVH phase timing (grace_albvh_enable_timing)
    bool phase_timing = false, phase_valid = false;
    hipEvent_t phase_events[3] = { nullptr, nullptr, nullptr };
    // a second stream for launches that are off the critical path (the sort's gated fallback):
    // forked from and joined to the call's stream by events, see side_fork / side_join
    hipStream_t side_stream = nullptr;
    hipEvent_t side_fork_ev = nullptr, side_join_ev = nullptr;
    // Bucket sort: did the last large sort on this context overflow a bucket?  A word of pinned,
    // device-mapped host memory that the sort's flag kernel writes (no copy, no synchronisation);
    // the NEXT large sort reads it as a hint -- keys that overflowed once (clustered snapshots)
    // mostly do again, and the attempt costs ~0.06 ms -- and goes straight to the index sort,
    // retrying the bucket sort every 8th time.  Only ever a choice between two correct paths.
    volatile uint32_t* sort_overflow_host = nullptr;
    uint32_t* sort_overflow_dev = nullptr;      // the device's view of the same word
    uint32_t sort_hint_skips = 0;
    // traversal state, owned by the trace translation units
    TraceState* trace = nullptr;
};

// The frame context's side stream, ordered after everything enqueued on `stream` so far ...
grace_status side_fork(hipStream_t stream, hipStream_t* side);
// ... and `stream` ordered after everything enqueued on the side stream so far.
grace_status side_join(hipStream_t stream);
// The frame context's overflow hint word (allocated on first use; null if pinned memory is refused).
uint32_t* sort_overflow_word(Context** ctx_out);

// The calling thread's context: the one it made current, else the current device's default
// context.  Fails if the thread's explicit context belongs to another device than the current one.
grace_status current_context(Context** out);
// Registered by the trace translation unit: frees a context's TraceState (device buffers, events).
extern grace_status (*g_trace_state_destroy)(Context&);

// Bump allocator over the current context's workspace.  A call opens a frame (FrameGuard below),
// carves its temporaries, and the frame's memory is reused by the next call on the context; stream
// order keeps earlier kernels safe, and a frame opened on another stream than the previous one first
// waits (device side) for the event the previous frame recorded when it closed.
class Workspace {
public:
    template <typename T>
    static T* take(size_t count)
    {
        Context& c = *frame_context();
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        char* p = c.ws_base + c.ws_used;
        c.ws_used += bytes;
        return reinterpret_cast<T*>(p);
    }
    static size_t aligned(size_t bytes) { return (bytes + 255) & ~size_t(255); }
    static grace_status reserve(size_t bytes);
    static grace_status release();
    // (use FrameGuard; these are its two halves)
    static grace_status begin(size_t bytes, hipStream_t stream);
    static void end(hipStream_t stream);
    static Context* frame_context();    // the context of the calling thread's open frame
};

// Scope of one call's workspace frame: begin() makes `bytes` available on the calling thread's
// context and resets the bump pointer; leaving the scope records the context's fence on `stream`.
class FrameGuard {
public:
    FrameGuard() = default;
    FrameGuard(const FrameGuard&) = delete;
    FrameGuard& operator=(const FrameGuard&) = delete;
    grace_status begin(size_t bytes, hipStream_t stream)
    {
        grace_status s = Workspace::begin(bytes, stream);
        if (s == GRACE_OK) { open_ = true; stream_ = stream; }
        return s;
    }
    ~FrameGuard() { if (open_) Workspace::end(stream_); }
private:
    bool open_ = false;
    hipStream_t stream_ = nullptr;
};

constexpr int WAVE = 64;

inline int ceil_div(size_t a, size_t b) { return static_cast<int>((a + b - 1) / b); }

// Grid for a streaming kernel: enough workgroups to fill 256 CUs several times over,
// grid-stride for the rest.
inline int stream_grid(size_t n, int block, int items_per_thread = 1)
{
    size_t want = (n + size_t(block) * items_per_thread - 1) / (size_t(block) * items_per_thread);
    const size_t cap = 256 * 16;
    if (want < 1) want = 1;
    return static_cast<int>(want < cap ? want : cap);
}

// ---- internal device utilities shared between translation units ------------------------

// Exclusive prefix sum of n uint32 (in place allowed).  d_total (optional, device) gets
// the grand total.  Needs scan_ws_count(n) uint32 of scratch.
size_t scan_ws_count(size_t n);
// run_if (device, optional): every kernel returns at once if *run_if == 0.
grace_status exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, size_t n,
                                uint32_t* d_scratch, uint32_t* d_total, hipStream_t stream,
                                const uint32_t* run_if = nullptr);

// Stable radix sort used inside other entry points.  The caller has already opened a
// workspace frame that includes sort_ws_bytes(); temporaries are carved from it.
size_t sort_ws_bytes(size_t n, int key_bytes, int value_bytes);
grace_status sort_pairs_u32_nested(uint32_t* d_keys, void* d_values, size_t n, int value_bytes,
                                   int begin_bit, int end_bit, uint32_t* d_perm,
                                   hipStream_t stream, const uint32_t* run_if = nullptr);
grace_status sort_pairs_u64_nested(uint64_t* d_keys, void* d_values, size_t n, int value_bytes,
                                   int begin_bit, int end_bit, uint32_t* d_perm,
                                   hipStream_t stream);

// Drops the prepared trace scene (grace_trace_prepare_*) if it was built over d_written.
grace_status scene_invalidate_if_written(const void* d_written);
// ... and a prepared ray batch (grace_trace_prepare_rays) when a ray generator writes to its array.
grace_status rays_invalidate_if_written(const void* d_written);

} // namespace grace_hip
