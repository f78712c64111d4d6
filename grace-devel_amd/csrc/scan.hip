// Prefix sums for gfx950: the exclusive scan of per-ray hit counts
// (reference include/grace/cuda/trace_sph.cuh:135-137, thrust::exclusive_scan), the
// CSR-segmented exclusive scan (include/grace/cuda/scan.cuh:15-37 -> sgpu
// SegScanCsrPreprocess/SegScanApply, external/sgpu/kernels/segscancsr.cuh:497-774) and
// multiply_by_weights (include/grace/cuda/kernels/weights.cuh:13-27).
//
// Structure: reduce-then-scan over contiguous per-workgroup slabs; inside a slab a wave64
// shuffle scan plus a 4-entry LDS hand-off between the workgroup's waves.  All loads are
// 16 B per lane, coalesced.  HBM-bound: 4 B read (reduce) + 4 B read + 4 B write (scan)
// per element = 12 B/element (fp64: 24).
#include "common.hpp"

using namespace grace_hip;

namespace {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_VEC = 4;                          // items per thread per chunk
constexpr int SCAN_CHUNK = SCAN_BLOCK * SCAN_VEC;    // 1024
constexpr int SCAN_CHUNKS_PER_SLAB = 8;
constexpr int SCAN_SLAB = SCAN_CHUNK * SCAN_CHUNKS_PER_SLAB; // 8192 items per workgroup
static_assert(SCAN_VEC == 4, "the u32 scan kernels move uint4 vectors: SCAN_VEC is not a tuning knob");

__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// Exclusive prefix of `v` over the 256 threads of the workgroup; *total = sum of all.
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t* s_wave,
                                                        uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_inclusive_sum(v, lane);
    __syncthreads(); // s_wave may still be read from the previous chunk
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        const uint32_t t = s_wave[w];
        if (w < wave) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_reduce_kernel(const uint32_t* __restrict__ in,
                                                                 size_t n,
                                                                 uint32_t* __restrict__ sums,
                                                                 const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
    const size_t slab0 = size_t(blockIdx.x) * SCAN_SLAB;
    uint32_t acc = 0;
#pragma unroll
    for (int c = 0; c < SCAN_CHUNKS_PER_SLAB; ++c) {
        const size_t i = slab0 + size_t(c) * SCAN_CHUNK + threadIdx.x * SCAN_VEC;
        if (i + SCAN_VEC <= n) {
            const uint4 v = *reinterpret_cast<const uint4*>(in + i);
            acc += v.x + v.y + v.z + v.w;
        } else {
            for (int k = 0; k < SCAN_VEC; ++k)
                if (i + k < n) acc += in[i + k];
        }
    }
    __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
    uint32_t total;
    block_exclusive_sum(acc, s_wave, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_slab_kernel(const uint32_t* in, uint32_t* out,
                                                               size_t n,
                                                               const uint32_t* __restrict__ carry_in,
                                                               uint32_t* __restrict__ total_out,
                                                               const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
    __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
    const size_t slab0 = size_t(blockIdx.x) * SCAN_SLAB;
    uint32_t carry = carry_in ? carry_in[blockIdx.x] : 0u;
    for (int c = 0; c < SCAN_CHUNKS_PER_SLAB; ++c) {
        const size_t chunk0 = slab0 + size_t(c) * SCAN_CHUNK;
        if (chunk0 >= n) break;
        const size_t i = chunk0 + threadIdx.x * SCAN_VEC;
        uint32_t v[SCAN_VEC] = { 0, 0, 0, 0 };
        const bool full = i + SCAN_VEC <= n;
        if (full) {
            const uint4 q = *reinterpret_cast<const uint4*>(in + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            for (int k = 0; k < SCAN_VEC; ++k)
                if (i + k < n) v[k] = in[i + k];
        }
        uint32_t total;
        uint32_t run = carry + block_exclusive_sum(v[0] + v[1] + v[2] + v[3], s_wave, &total);
        uint32_t o[SCAN_VEC];
#pragma unroll
        for (int k = 0; k < SCAN_VEC; ++k) { o[k] = run; run += v[k]; }
        if (full) {
            *reinterpret_cast<uint4*>(out + i) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            for (int k = 0; k < SCAN_VEC; ++k)
                if (i + k < n) out[i + k] = o[k];
        }
        carry += total;
    }
    if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = carry;
}

} // namespace

namespace grace_hip {

size_t scan_ws_count(size_t n)
{
    size_t count = 0;
    while (n > size_t(SCAN_SLAB)) {
        n = (n + SCAN_SLAB - 1) / SCAN_SLAB;
        count += (n + 63) & ~size_t(63);
    }
    return count + 64;
}

grace_status exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, size_t n,
                                uint32_t* d_scratch, uint32_t* d_total, hipStream_t stream,
                                const uint32_t* run_if)
{
    if (n == 0) {
        if (d_total) GRACE_TRY_HIP(hipMemsetAsync(d_total, 0, 4, stream));
        return GRACE_OK;
    }
    const size_t n_slabs = (n + SCAN_SLAB - 1) / SCAN_SLAB;
    // (A single-workgroup kernel for tables of a few slabs -- the ray-order sort's 65536 counters --
    // was tried in round 3 to save two launches: one workgroup streams the table at ~1.5 GB/s,
    // 97 us against the 12 us of the three launches below.)
    if (n_slabs == 1) {
        scan_slab_kernel<<<1, SCAN_BLOCK, 0, stream>>>(d_in, d_out, n, nullptr, d_total, run_if);
        GRACE_CHECK_LAUNCH();
        return GRACE_OK;
    }
    uint32_t* sums = d_scratch;
    uint32_t* next_scratch = d_scratch + ((n_slabs + 63) & ~size_t(63));
    scan_reduce_kernel<<<int(n_slabs), SCAN_BLOCK, 0, stream>>>(d_in, n, sums, run_if);
    GRACE_CHECK_LAUNCH();
    // The carry of slab b is the exclusive prefix of the slab sums; the last slab's
    // running carry is the grand total, so the recursion does not need to report one.
    GRACE_TRY(exclusive_scan_u32(sums, sums, n_slabs, next_scratch, nullptr, stream, run_if));
    scan_slab_kernel<<<int(n_slabs), SCAN_BLOCK, 0, stream>>>(d_in, d_out, n, sums, d_total, run_if);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

} // namespace grace_hip

// ---------------------------------------------------------------------------------------
// Segmented scan
// ---------------------------------------------------------------------------------------
namespace {

// (flag, value) pairs under the segmented-sum operator:
//   (fa, a) (+) (fb, b) = (fa | fb, fb ? b : a + b)
template <typename T>
struct SegPair {
    T v;
    int f;
};

template <typename T>
__device__ __forceinline__ SegPair<T> seg_combine(SegPair<T> a, SegPair<T> b)
{
    SegPair<T> r;
    r.f = a.f | b.f;
    r.v = b.f ? b.v : a.v + b.v;
    return r;
}

template <typename T>
__device__ __forceinline__ T shfl_up_t(T v, int off)
{
    return __shfl_up(v, off);
}

// Inclusive segmented scan across the workgroup of one pair per thread; returns the
// EXCLUSIVE prefix pair for this thread (identity for thread 0) and the workgroup total.
template <typename T>
__device__ __forceinline__ SegPair<T> block_exclusive_seg(SegPair<T> p, SegPair<T>* s_wave,
                                                          SegPair<T>* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    SegPair<T> incl = p;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        SegPair<T> t;
        t.v = shfl_up_t(incl.v, off);
        t.f = __shfl_up(incl.f, off);
        if (lane >= off) incl = seg_combine(t, incl);
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    // Exclusive within the wave: shift by one lane.
    SegPair<T> excl;
    excl.v = shfl_up_t(incl.v, 1);
    excl.f = __shfl_up(incl.f, 1);
    if (lane == 0) { excl.v = T(0); excl.f = 0; }
    SegPair<T> base; base.v = T(0); base.f = 0;
    SegPair<T> tot; tot.v = T(0); tot.f = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        const SegPair<T> t = s_wave[w];
        if (w < wave) base = seg_combine(base, t);
        tot = seg_combine(tot, t);
    }
    *total = tot;
    return seg_combine(base, excl);
}

__device__ __forceinline__ size_t lower_bound_i32(const int* __restrict__ a, size_t n,
                                                  long long key)
{
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) >> 1;
        if ((long long)a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// table[b] = index of the first segment starting at or after slab b's first element, for
// b = 0 .. n_slabs: one thread per slab, so the binary searches over the offsets run in
// parallel once instead of serially in every workgroup (they used to be two dependent chains
// of ~17 global loads per 1024-element chunk, which bounded the scan).
__global__ __launch_bounds__(256) void seg_slab_bounds_kernel(const int* __restrict__ offsets,
                                                              size_t n_seg, size_t n_slabs,
                                                              uint32_t* __restrict__ table)
{
    const size_t b = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (b <= n_slabs) table[b] = uint32_t(lower_bound_i32(offsets, n_seg, (long long)(b * SCAN_SLAB)));
}

// Head flags of one 8192-element slab into LDS, from CSR offsets (level 0; the slab's range of
// segments comes from `table`) or from an explicit per-element flag array (spine levels).
template <bool FROM_OFFSETS>
__device__ __forceinline__ void load_heads(unsigned char* s_head, size_t slab0, size_t n,
                                           const int* __restrict__ offsets,
                                           const uint32_t* __restrict__ table,
                                           const unsigned char* __restrict__ flags)
{
    if (FROM_OFFSETS) {
        uint32_t* w = reinterpret_cast<uint32_t*>(s_head);
        for (int k = threadIdx.x; k < SCAN_SLAB / 4; k += SCAN_BLOCK) w[k] = 0u;
        __syncthreads();
        const size_t s0 = table[blockIdx.x], s1 = table[blockIdx.x + 1];
        for (size_t s = s0 + threadIdx.x; s < s1; s += SCAN_BLOCK)
            s_head[(long long)offsets[s] - (long long)slab0] = 1;
    } else {
        for (int k = threadIdx.x; k < SCAN_SLAB; k += SCAN_BLOCK)
            s_head[k] = slab0 + k < n ? flags[slab0 + k] : 0;
    }
    __syncthreads();
}

// Upsweep: per slab, (has a head, sum of the elements after the slab's last head).
template <typename T, bool FROM_OFFSETS>
__global__ __launch_bounds__(SCAN_BLOCK) void seg_reduce_kernel(
    const T* __restrict__ data, size_t n, const int* __restrict__ offsets,
    const uint32_t* __restrict__ table, const unsigned char* __restrict__ flags,
    T* __restrict__ agg_v, unsigned char* __restrict__ agg_f)
{
    __shared__ __align__(4) unsigned char s_slab_head[SCAN_SLAB];
    __shared__ SegPair<T> s_wave[SCAN_BLOCK / 64];
    const size_t slab0 = size_t(blockIdx.x) * SCAN_SLAB;
    load_heads<FROM_OFFSETS>(s_slab_head, slab0, n, offsets, table, flags);
    SegPair<T> carry; carry.v = T(0); carry.f = 0;
    for (int c = 0; c < SCAN_CHUNKS_PER_SLAB; ++c) {
        const size_t chunk0 = slab0 + size_t(c) * SCAN_CHUNK;
        if (chunk0 >= n) break;
        const unsigned char* s_head = s_slab_head + c * SCAN_CHUNK;
        const size_t i = chunk0 + threadIdx.x * SCAN_VEC;
        SegPair<T> p; p.v = T(0); p.f = 0;
#pragma unroll
        for (int k = 0; k < SCAN_VEC; ++k) {
            if (i + k < n) {
                SegPair<T> e; e.v = data[i + k]; e.f = s_head[threadIdx.x * SCAN_VEC + k];
                p = seg_combine(p, e);
            }
        }
        SegPair<T> total;
        block_exclusive_seg(p, s_wave, &total);
        carry = seg_combine(carry, total);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        agg_v[blockIdx.x] = carry.v;
        agg_f[blockIdx.x] = (unsigned char)carry.f;
    }
}

// Downsweep.  CARRY_MODE false: out[i] = sum of the elements of i's segment before i
// (0 at a head).  CARRY_MODE true (spine): out[i] = segmented sum of everything before i
// since the last head before i (heads are not reset first) = the carry into slab i.
template <typename T, bool FROM_OFFSETS, bool CARRY_MODE>
__global__ __launch_bounds__(SCAN_BLOCK) void seg_scan_kernel(
    const T* data, T* out, size_t n, const int* __restrict__ offsets,
    const uint32_t* __restrict__ table, const unsigned char* __restrict__ flags,
    const T* __restrict__ carry_in)
{
    __shared__ __align__(4) unsigned char s_slab_head[SCAN_SLAB];
    __shared__ SegPair<T> s_wave[SCAN_BLOCK / 64];
    const size_t slab0 = size_t(blockIdx.x) * SCAN_SLAB;
    load_heads<FROM_OFFSETS>(s_slab_head, slab0, n, offsets, table, flags);
    SegPair<T> carry; carry.v = carry_in ? carry_in[blockIdx.x] : T(0); carry.f = 0;
    for (int c = 0; c < SCAN_CHUNKS_PER_SLAB; ++c) {
        const size_t chunk0 = slab0 + size_t(c) * SCAN_CHUNK;
        if (chunk0 >= n) break;
        const unsigned char* s_head = s_slab_head + c * SCAN_CHUNK;
        const size_t i = chunk0 + threadIdx.x * SCAN_VEC;
        T v[SCAN_VEC]; int h[SCAN_VEC];
        SegPair<T> p; p.v = T(0); p.f = 0;
#pragma unroll
        for (int k = 0; k < SCAN_VEC; ++k) {
            v[k] = T(0); h[k] = 0;
            if (i + k < n) {
                v[k] = data[i + k]; h[k] = s_head[threadIdx.x * SCAN_VEC + k];
                SegPair<T> e; e.v = v[k]; e.f = h[k];
                p = seg_combine(p, e);
            }
        }
        SegPair<T> total;
        const SegPair<T> before = seg_combine(carry, block_exclusive_seg(p, s_wave, &total));
        T run = before.v;
#pragma unroll
        for (int k = 0; k < SCAN_VEC; ++k) {
            if (i + k < n) {
                if (CARRY_MODE) {
                    out[i + k] = run;
                    run = h[k] ? v[k] : run + v[k];
                } else {
                    if (h[k]) run = T(0);
                    out[i + k] = run;
                    run = run + v[k];
                }
            }
        }
        carry = seg_combine(carry, total);
        __syncthreads();
    }
}

template <typename T>
size_t seg_ws_bytes(size_t n)
{
    size_t bytes = 0;
    while (n > size_t(SCAN_SLAB)) {
        n = (n + SCAN_SLAB - 1) / SCAN_SLAB;
        bytes += Workspace::aligned(n * sizeof(T)) + Workspace::aligned(n);
    }
    return bytes + 512;
}

// Level >= 1: data/flags arrays of slab aggregates; produces carries in place.
template <typename T>
grace_status seg_spine(T* d_v, const unsigned char* d_f, size_t n, hipStream_t stream)
{
    const size_t n_slabs = (n + SCAN_SLAB - 1) / SCAN_SLAB;
    if (n_slabs == 1) {
        seg_scan_kernel<T, false, true><<<1, SCAN_BLOCK, 0, stream>>>(d_v, d_v, n, nullptr,
                                                                      nullptr, d_f, nullptr);
        GRACE_CHECK_LAUNCH();
        return GRACE_OK;
    }
    T* agg_v = Workspace::take<T>(n_slabs);
    unsigned char* agg_f = Workspace::take<unsigned char>(n_slabs);
    seg_reduce_kernel<T, false><<<int(n_slabs), SCAN_BLOCK, 0, stream>>>(d_v, n, nullptr, nullptr,
                                                                         d_f, agg_v, agg_f);
    GRACE_CHECK_LAUNCH();
    GRACE_TRY(seg_spine<T>(agg_v, agg_f, n_slabs, stream));
    seg_scan_kernel<T, false, true><<<int(n_slabs), SCAN_BLOCK, 0, stream>>>(d_v, d_v, n, nullptr,
                                                                             nullptr, d_f, agg_v);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

template <typename T>
grace_status segscan(const int* d_offsets, size_t n_seg, const T* d_data, size_t n, T* d_out,
                     hipStream_t stream)
{
    GRACE_REQUIRE(n == 0 || (d_data && d_out), "segmented scan: null data");
    GRACE_REQUIRE(n_seg == 0 || d_offsets, "segmented scan: null offsets");
    if (n == 0) return GRACE_OK;
    const size_t n_slabs = (n + SCAN_SLAB - 1) / SCAN_SLAB;
    FrameGuard frame;
    GRACE_TRY(frame.begin(seg_ws_bytes<T>(n) + Workspace::aligned((n_slabs + 1) * 4), stream));
    uint32_t* table = Workspace::take<uint32_t>(n_slabs + 1);
    seg_slab_bounds_kernel<<<int((n_slabs + 256) / 256), 256, 0, stream>>>(d_offsets, n_seg, n_slabs,
                                                                            table);
    GRACE_CHECK_LAUNCH();
    if (n_slabs == 1) {
        seg_scan_kernel<T, true, false><<<1, SCAN_BLOCK, 0, stream>>>(d_data, d_out, n, d_offsets,
                                                                      table, nullptr, nullptr);
        GRACE_CHECK_LAUNCH();
        return GRACE_OK;
    }
    T* agg_v = Workspace::take<T>(n_slabs);
    unsigned char* agg_f = Workspace::take<unsigned char>(n_slabs);
    seg_reduce_kernel<T, true><<<int(n_slabs), SCAN_BLOCK, 0, stream>>>(
        d_data, n, d_offsets, table, nullptr, agg_v, agg_f);
    GRACE_CHECK_LAUNCH();
    GRACE_TRY(seg_spine<T>(agg_v, agg_f, n_slabs, stream));
    seg_scan_kernel<T, true, false><<<int(n_slabs), SCAN_BLOCK, 0, stream>>>(
        d_data, d_out, n, d_offsets, table, nullptr, agg_v);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

// thrust::transform(offsets, counting_iterator(0), plus) of trace_with_sentinels_sph
// (include/grace/cuda/trace_sph.cuh:205-208) and the sentinel fill of its resize calls
// (trace_sph.cuh:212-214).
__global__ __launch_bounds__(256) void add_iota_kernel(int* __restrict__ v, size_t n)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x)
        v[i] += int(i);
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ v, size_t n,
                                                       uint32_t value)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x)
        v[i] = value;
}

template <typename Real>
__global__ __launch_bounds__(256) void multiply_by_weights_kernel(
    const Real* __restrict__ x, size_t n, const Real* __restrict__ w,
    const uint32_t* __restrict__ map, Real* __restrict__ out)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x)
        out[i] = w[map[i]] * x[i];
}

// 64-bit sum of non-negative int32 counts (one atomic per workgroup).
__global__ __launch_bounds__(256) void sum_i32_u64_kernel(const int* __restrict__ in, size_t n,
                                                          unsigned long long* __restrict__ total)
{
    unsigned long long acc = 0;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x)
        acc += (unsigned long long)(uint32_t)in[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    __shared__ unsigned long long s_acc[4];
    if ((threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, s_acc[0] + s_acc[1] + s_acc[2] + s_acc[3]);
}

} // namespace

extern "C" {

grace_status grace_scan_exclusive_i32(const int* d_in, size_t n, int* d_out, long long* h_total,
                                      grace_stream stream)
{
    GRACE_REQUIRE(n == 0 || (d_in && d_out), "scan: null pointer");
    hipStream_t st = as_stream(stream);
    FrameGuard frame;
    GRACE_TRY(frame.begin((scan_ws_count(n) + 128) * sizeof(uint32_t), st));
    unsigned long long* d_total = Workspace::take<unsigned long long>(1);
    uint32_t* scratch = Workspace::take<uint32_t>(scan_ws_count(n));
    if (h_total) {
        // The grand total in 64 bits, taken BEFORE the scan (d_out may alias d_in): the scan
        // itself wraps modulo 2^32 like the reference's int scan (trace_sph.cuh:135-137), and
        // the callers must be able to tell (offsets past INT32_MAX cannot address the per-hit
        // arrays).
        GRACE_TRY_HIP(hipMemsetAsync(d_total, 0, 8, st));
        if (n) {
            sum_i32_u64_kernel<<<stream_grid(n, 256, 4), 256, 0, st>>>(d_in, n, d_total);
            GRACE_CHECK_LAUNCH();
        }
    }
    GRACE_TRY(exclusive_scan_u32(reinterpret_cast<const uint32_t*>(d_in),
                                 reinterpret_cast<uint32_t*>(d_out), n, scratch, nullptr, st));
    if (h_total) {
        unsigned long long t = 0;
        GRACE_TRY_HIP(hipMemcpyAsync(&t, d_total, 8, hipMemcpyDeviceToHost, st));
        GRACE_TRY_HIP(hipStreamSynchronize(st));
        *h_total = (long long)t;
    }
    return GRACE_OK;
}

grace_status grace_segscan_exclusive_f32(const int* d_segment_offsets, size_t n_segments,
                                         const float* d_data, size_t n, float* d_results,
                                         grace_stream stream)
{
    return segscan<float>(d_segment_offsets, n_segments, d_data, n, d_results, as_stream(stream));
}

grace_status grace_segscan_exclusive_f64(const int* d_segment_offsets, size_t n_segments,
                                         const double* d_data, size_t n, double* d_results,
                                         grace_stream stream)
{
    return segscan<double>(d_segment_offsets, n_segments, d_data, n, d_results,
                           as_stream(stream));
}

grace_status grace_add_iota_i32(int* d_values, size_t n, grace_stream stream)
{
    GRACE_REQUIRE(n == 0 || d_values, "add_iota: null pointer");
    if (n == 0) return GRACE_OK;
    add_iota_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(d_values, n);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status grace_fill_u32(void* d_values, size_t n, uint32_t bits, grace_stream stream)
{
    GRACE_REQUIRE(n == 0 || d_values, "fill: null pointer");
    if (n == 0) return GRACE_OK;
    fill_u32_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(
        static_cast<uint32_t*>(d_values), n, bits);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status grace_multiply_by_weights_f32(const float* d_unweighted, size_t n,
                                           const float* d_weights, const uint32_t* d_weight_map,
                                           float* d_weighted, grace_stream stream)
{
    GRACE_REQUIRE(n == 0 || (d_unweighted && d_weights && d_weight_map && d_weighted),
                  "multiply_by_weights: null pointer");
    if (n == 0) return GRACE_OK;
    multiply_by_weights_kernel<float><<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(
        d_unweighted, n, d_weights, d_weight_map, d_weighted);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status grace_multiply_by_weights_f64(const double* d_unweighted, size_t n,
                                           const double* d_weights, const uint32_t* d_weight_map,
                                           double* d_weighted, grace_stream stream)
{
    GRACE_REQUIRE(n == 0 || (d_unweighted && d_weights && d_weight_map && d_weighted),
                  "multiply_by_weights: null pointer");
    if (n == 0) return GRACE_OK;
    multiply_by_weights_kernel<double><<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(
        d_unweighted, n, d_weights, d_weight_map, d_weighted);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

} // extern "C"
