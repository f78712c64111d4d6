// Per-ray sort of hits by distance for gfx950: grace::sort_by_distance
// (reference include/grace/cuda/sort.cuh:100-131 -> sgpu SegSortPairsFromIndices,
// external/sgpu/kernels/segmentedsort.cuh:733-790, then thrust::gather of indices and data).
//
// Contract restated: within every segment [offsets[s], offsets[s+1]) the distances end up
// in non-decreasing order, equal distances keep their input order (merge sort is stable),
// and hit_indices / hit_data are permuted by the same map.
//
// Design: hits of one ray are already contiguous, so a stable LSD radix sort over the
// composite 64-bit key (segment id << 32 | order-preserving bits of the distance) sorts all
// segments at once with the wave64 radix sort of sort.hip: 4 digit passes over the distance
// bits + ceil(log2(n_segments) / 8) passes over the segment bits, every pass streaming
// (key, index) pairs; the three payload arrays are gathered once at the end.
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
    const size_t ws = 3 * Workspace::aligned(n_hits * 4) + Workspace::aligned(scan_ws_count(n_hits) * 4)
        + Workspace::aligned(n_hits * 8) + Workspace::aligned(n_hits * 4)
        + sort_ws_bytes(n_hits, 8, 0) + 1024;
    GRACE_TRY(Workspace::begin(ws));
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

} // extern "C"
