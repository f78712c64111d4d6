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

// Bump allocator over the library's grow-only workspace.  A call opens a frame, carves
// its temporaries, and the frame is implicitly dropped at the next call (stream order
// keeps earlier kernels safe because every entry point runs on one stream at a time).
class Workspace {
public:
    // Makes sure `bytes` are available and resets the bump pointer.  `stream` is the stream the
    // call's kernels run on: a frame opened on another stream than the previous one first waits
    // (device side) for everything the previous frame's stream has been given, since the two
    // frames share the same memory.
    static grace_status begin(size_t bytes, hipStream_t stream);
    template <typename T>
    static T* take(size_t count)
    {
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        char* p = base_ + used_;
        used_ += bytes;
        return reinterpret_cast<T*>(p);
    }
    static size_t aligned(size_t bytes) { return (bytes + 255) & ~size_t(255); }
    static grace_status reserve(size_t bytes);
    static grace_status release();

private:
    static char* base_;
    static size_t capacity_;
    static size_t used_;
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
grace_status exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, size_t n,
                                uint32_t* d_scratch, uint32_t* d_total, hipStream_t stream);

// Stable radix sort used inside other entry points.  The caller has already opened a
// workspace frame that includes sort_ws_bytes(); temporaries are carved from it.
size_t sort_ws_bytes(size_t n, int key_bytes, int value_bytes);
grace_status sort_pairs_u32_nested(uint32_t* d_keys, void* d_values, size_t n, int value_bytes,
                                   int begin_bit, int end_bit, uint32_t* d_perm,
                                   hipStream_t stream);
grace_status sort_pairs_u64_nested(uint64_t* d_keys, void* d_values, size_t n, int value_bytes,
                                   int begin_bit, int end_bit, uint32_t* d_perm,
                                   hipStream_t stream);

// Drops the prepared trace scene (grace_trace_prepare_*) if it was built over d_written.
grace_status scene_invalidate_if_written(const void* d_written);
// ... and a prepared ray batch (grace_trace_prepare_rays) when a ray generator writes to its array.
grace_status rays_invalidate_if_written(const void* d_written);

} // namespace grace_hip
