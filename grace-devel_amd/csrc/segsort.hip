// Per-ray sort of hits by distance for gfx950: grace::sort_by_distance
// (reference include/grace/cuda/sort.cuh:100-131 -> sgpu SegSortPairsFromIndices,
// external/sgpu/kernels/segmentedsort.cuh:733-790, then thrust::gather of indices and data).
//
// Contract restated: within every segment [offsets[s], offsets[s+1]) the distances end up
// in non-decreasing order, equal distances keep their input order (merge sort is stable),
// and hit_indices / hit_data are permuted by the same map.
//
// Design: hits of one ray are already contiguous.
//   * typical segments (mean length <= 32768): one wavefront per segment runs a stable LSD
//     radix sort of the segment alone (segsort_wave_kernel below) -- no segment bits to sort,
//     no inter-wave communication, ~100 B of cache-resident traffic per hit;
//   * very long segments on average: a stable LSD radix sort over the composite 64-bit key
//     (segment id << 32 | order-preserving bits of the distance) sorts all segments at once
//     with the wave64 radix sort of sort.hip: 4 digit passes over the distance bits +
//     ceil(log2(n_segments) / 8) passes over the segment bits, every pass streaming
//     (key, index) pairs; the three payload arrays are gathered once at the end.
#include "common.hpp"

using namespace grace_hip;

namespace {

__global__ __launch_bounds__(256) void seg_heads_kernel(const int* __restrict__ offsets, size_t n_seg,
                                                        size_t n, uint32_t* __restrict__ heads)
{
    // heads[i] = number of segments that start at element i (segment 0 excluded): empty
    // segments stack up on the same element, so atomics.
    for (size_t s = blockIdx.x * size_t(blockDim.x) + threadIdx.x + 1; s < n_seg;
         s += size_t(gridDim.x) * blockDim.x) {
        const size_t o = size_t(offsets[s]);
        if (o < n) atomicAdd(&heads[o], 1u);
    }
}

__global__ __launch_bounds__(256) void seg_keys_kernel(const float* __restrict__ dist,
                                                       const uint32_t* __restrict__ heads,
                                                       const uint32_t* __restrict__ heads_excl,
                                                       size_t n, uint64_t* __restrict__ keys)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        const uint32_t seg = heads_excl[i] + heads[i]; // inclusive scan = segment id
        float d = dist[i];
        if (d == 0.0f) d = 0.0f;                       // -0 and +0 compare equal under less<>
        uint32_t u = __float_as_uint(d);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        keys[i] = (uint64_t(seg) << 32) | u;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_kernel(const T* __restrict__ in,
                                                     const uint32_t* __restrict__ perm, size_t n,
                                                     T* __restrict__ out)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x)
        out[i] = in[perm[i]];
}

// One wavefront per segment: a stable LSD radix sort (four 8-bit digits of the order-preserving
// distance bits) of the segment's (key, local index) pairs, ping-ponging between two global
// buffers that stay cache-resident (a ray's hits are a few kB), then one gather of the three
// payload arrays into temporaries and a coalesced copy back.  No inter-wave communication at
// all; a wave's 256 digit counters live in LDS.  Traffic ~100 B per hit instead of the ~250 B
// of the composite-key global sort (which also has to sort the segment bits), and ~10^4
// independent waves hide each other's latency.
constexpr int SEG_WAVES = 4;

__device__ __forceinline__ uint32_t dist_key(float d)
{
    if (d == 0.0f) d = 0.0f;                           // -0 and +0 compare equal under less<>
    const uint32_t u = __float_as_uint(d);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ uint64_t dist_key(double d)
{
    if (d == 0.0) d = 0.0;
    const uint64_t u = uint64_t(__double_as_longlong(d));
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

// Real = float (Key uint32_t, four digit passes) or double (Key uint64_t, eight).
template <typename Real, typename Key>
__global__ __launch_bounds__(64 * SEG_WAVES) void segsort_wave_kernel(
    Real* __restrict__ dist, const int* __restrict__ offsets, size_t n_seg, size_t n,
    int* __restrict__ hidx, Real* __restrict__ data, Key* __restrict__ k0,
    Key* __restrict__ k1, uint32_t* __restrict__ i0, uint32_t* __restrict__ i1,
    Real* __restrict__ t_d, int* __restrict__ t_i, Real* __restrict__ t_w)
{
    constexpr int PASSES = int(sizeof(Key));
    __shared__ uint32_t s_base[SEG_WAVES][PASSES][256];   // one digit histogram per pass
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t seg = size_t(blockIdx.x) * SEG_WAVES + wave;
    if (seg >= n_seg) return;
    const size_t beg = size_t(offsets[seg]);
    const size_t end = seg + 1 < n_seg ? size_t(offsets[seg + 1]) : n;
    if (end <= beg + 1) return;                        // empty or single hit
    const int len = int(end - beg);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // all digit histograms in one read of the distances
#pragma unroll
    for (int b = 0; b < 4 * PASSES; ++b) (&s_base[wave][0][0])[4 * PASSES * lane + b] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < len; i += 64) {
        const Key key = dist_key(dist[beg + i]);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) atomicAdd(&s_base[wave][p][uint32_t(key >> (8 * p)) & 255u], 1u);
    }
    __builtin_amdgcn_wave_barrier();

    Key* kin = k0; Key* kout = k1;
    uint32_t* iin = i0; uint32_t* iout = i1;
    for (int pass = 0; pass < PASSES; ++pass) {
        const int shift = 8 * pass;
        uint32_t* base = s_base[wave][pass];
        // exclusive scan of the 256 counters: four per lane, then across the wave
        uint32_t c[4], s4 = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) { c[b] = base[4 * lane + b]; s4 += c[b]; }
        uint32_t incl = s4;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        uint32_t run = incl - s4;
#pragma unroll
        for (int b = 0; b < 4; ++b) { base[4 * lane + b] = run; run += c[b]; }
        __builtin_amdgcn_wave_barrier();
        // stable scatter, 64 elements at a time in input order
        for (int t0 = 0; t0 < len; t0 += 64) {
            const int i = t0 + lane;
            const bool valid = i < len;
            Key key = 0;
            uint32_t idx = uint32_t(i);
            if (valid) {
                key = pass == 0 ? dist_key(dist[beg + i]) : kin[beg + i];
                if (pass != 0) idx = iin[beg + i];
            }
            const uint32_t d = uint32_t(key >> shift) & 255u;
            unsigned long long same = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const unsigned long long vote = __ballot(bit);
                same &= bit ? vote : ~vote;
            }
            const uint32_t below = __popcll(same & lt_mask);
            const uint32_t prior = base[d];            // every lane reads before any leader writes
            if (valid && below == 0) base[d] = prior + __popcll(same);
            if (valid) {
                kout[beg + prior + below] = key;
                iout[beg + prior + below] = idx;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // the next pass (other lanes of this wave) reads what this pass wrote
        __threadfence_block();   // same CU, same L1: workgroup scope is enough (no L2 write-back)
        Key* tk = kin; kin = kout; kout = tk;
        uint32_t* t = iin; iin = iout; iout = t;
    }
    // after an even number of passes the sorted order sits in iin (= i0): gather, then copy back
    for (int i = lane; i < len; i += 64) {
        const size_t src = beg + iin[beg + i];
        t_d[beg + i] = dist[src];
        if (hidx) t_i[beg + i] = hidx[src];
        if (data) t_w[beg + i] = data[src];
    }
    __threadfence_block();   // same CU, same L1: workgroup scope is enough (no L2 write-back)
    for (int i = lane; i < len; i += 64) {
        dist[beg + i] = t_d[beg + i];
        if (hidx) hidx[beg + i] = t_i[beg + i];
        if (data) data[beg + i] = t_w[beg + i];
    }
}

} // namespace

extern "C" {

grace_status grace_sort_by_distance_f32(float* d_distances, const int* d_ray_offsets,
                                        size_t n_rays, size_t n_hits, int* d_hit_indices,
                                        float* d_hit_data, grace_stream stream)
{
    GRACE_REQUIRE(n_hits == 0 || (d_distances && d_ray_offsets), "sort_by_distance: null pointer");
    GRACE_REQUIRE(n_hits < (size_t(1) << 32), "sort_by_distance: at most 2^32 - 1 hits");
    if (n_hits < 2 || n_rays == 0) return GRACE_OK;
    hipStream_t st = as_stream(stream);
    FrameGuard frame;
    // Typical ray segments (up to tens of thousands of hits): one wavefront per segment.
    // Very long segments on average: the composite-key global sort below.
    if (n_hits / n_rays <= 32768) {
        GRACE_TRY(frame.begin(7 * Workspace::aligned(n_hits * 4) + 1024, st));
        uint32_t* k0 = Workspace::take<uint32_t>(n_hits);
        uint32_t* k1 = Workspace::take<uint32_t>(n_hits);
        uint32_t* i0 = Workspace::take<uint32_t>(n_hits);
        uint32_t* i1 = Workspace::take<uint32_t>(n_hits);
        float* t_d = Workspace::take<float>(n_hits);
        int* t_i = Workspace::take<int>(n_hits);
        float* t_w = Workspace::take<float>(n_hits);
        segsort_wave_kernel<float, uint32_t><<<ceil_div(n_rays, size_t(SEG_WAVES)), 64 * SEG_WAVES, 0, st>>>(
            d_distances, d_ray_offsets, n_rays, n_hits, d_hit_indices, d_hit_data, k0, k1, i0, i1,
            t_d, t_i, t_w);
        GRACE_CHECK_LAUNCH();
        return GRACE_OK;
    }
    const size_t ws = 3 * Workspace::aligned(n_hits * 4) + Workspace::aligned(scan_ws_count(n_hits) * 4)
        + Workspace::aligned(n_hits * 8) + Workspace::aligned(n_hits * 4)
        + sort_ws_bytes(n_hits, 8, 0) + 1024;
    GRACE_TRY(frame.begin(ws, st));
    uint32_t* heads = Workspace::take<uint32_t>(n_hits);
    uint32_t* heads_excl = Workspace::take<uint32_t>(n_hits);
    uint32_t* perm = Workspace::take<uint32_t>(n_hits);
    uint32_t* scan_ws = Workspace::take<uint32_t>(scan_ws_count(n_hits));
    uint64_t* keys = Workspace::take<uint64_t>(n_hits);
    float* tmp = Workspace::take<float>(n_hits);

    GRACE_TRY_HIP(hipMemsetAsync(heads, 0, n_hits * 4, st));
    seg_heads_kernel<<<stream_grid(n_rays, 256), 256, 0, st>>>(d_ray_offsets, n_rays, n_hits, heads);
    GRACE_CHECK_LAUNCH();
    GRACE_TRY(exclusive_scan_u32(heads, heads_excl, n_hits, scan_ws, nullptr, st));
    seg_keys_kernel<<<stream_grid(n_hits, 256), 256, 0, st>>>(d_distances, heads, heads_excl, n_hits,
                                                             keys);
    GRACE_CHECK_LAUNCH();
    int seg_bits = 1;
    while ((size_t(1) << seg_bits) < n_rays && seg_bits < 31) ++seg_bits;
    GRACE_TRY(sort_pairs_u64_nested(keys, nullptr, n_hits, 0, 0, 32 + seg_bits, perm, st));

    const int grid = stream_grid(n_hits, 256);
    GRACE_TRY_HIP(hipMemcpyAsync(tmp, d_distances, n_hits * 4, hipMemcpyDeviceToDevice, st));
    gather_kernel<float><<<grid, 256, 0, st>>>(tmp, perm, n_hits, d_distances);
    GRACE_CHECK_LAUNCH();
    if (d_hit_indices) {
        int* itmp = reinterpret_cast<int*>(tmp);
        GRACE_TRY_HIP(hipMemcpyAsync(itmp, d_hit_indices, n_hits * 4, hipMemcpyDeviceToDevice, st));
        gather_kernel<int><<<grid, 256, 0, st>>>(itmp, perm, n_hits, d_hit_indices);
        GRACE_CHECK_LAUNCH();
    }
    if (d_hit_data) {
        GRACE_TRY_HIP(hipMemcpyAsync(tmp, d_hit_data, n_hits * 4, hipMemcpyDeviceToDevice, st));
        gather_kernel<float><<<grid, 256, 0, st>>>(tmp, perm, n_hits, d_hit_data);
        GRACE_CHECK_LAUNCH();
    }
    return GRACE_OK;
}

// sort_by_distance<double, int, double> (sort.cuh:97-131 with the double outputs of
// trace_sph<double4, int, double>): one wavefront per segment, eight digit passes.
grace_status grace_sort_by_distance_f64(double* d_distances, const int* d_ray_offsets,
                                        size_t n_rays, size_t n_hits, int* d_hit_indices,
                                        double* d_hit_data, grace_stream stream)
{
    GRACE_REQUIRE(n_hits == 0 || (d_distances && d_ray_offsets), "sort_by_distance: null pointer");
    GRACE_REQUIRE(n_hits < (size_t(1) << 32), "sort_by_distance: at most 2^32 - 1 hits");
    if (n_hits < 2 || n_rays == 0) return GRACE_OK;
    hipStream_t st = as_stream(stream);
    FrameGuard frame;
    GRACE_TRY(frame.begin(4 * Workspace::aligned(n_hits * 8) + 3 * Workspace::aligned(n_hits * 4) + 1024, st));
    uint64_t* k0 = Workspace::take<uint64_t>(n_hits);
    uint64_t* k1 = Workspace::take<uint64_t>(n_hits);
    uint32_t* i0 = Workspace::take<uint32_t>(n_hits);
    uint32_t* i1 = Workspace::take<uint32_t>(n_hits);
    double* t_d = Workspace::take<double>(n_hits);
    int* t_i = Workspace::take<int>(n_hits);
    double* t_w = Workspace::take<double>(n_hits);
    segsort_wave_kernel<double, uint64_t><<<ceil_div(n_rays, size_t(SEG_WAVES)), 64 * SEG_WAVES, 0, st>>>(
        d_distances, d_ray_offsets, n_rays, n_hits, d_hit_indices, d_hit_data, k0, k1, i0, i1,
        t_d, t_i, t_w);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

} // extern "C"
