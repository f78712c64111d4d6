// Small kernels around the traversal launches of trace.hip: the upper levels of the class sum
// tree for split packets, and the plan of the split per-hit trace.
#pragma once

#include "trace_state.hpp"

namespace {

using namespace grace_hip;

// Upper levels of the pairwise summation tree for split packets: K subtree sums per ray.
template <typename Real>
__global__ __launch_bounds__(256) void combine_classes_kernel(const Real* __restrict__ partial,
                                                              int n_rays, int split,
                                                              const int* __restrict__ split_dev,
                                                              Real* __restrict__ out,
                                                              const int* __restrict__ run_if = nullptr)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    if (run_if && *run_if == 0) return;   // the one-wave-per-packet kernel ran: `out` is final
    if (split_dev) split = *split_dev;
    Real t[SUM_CLASSES];
    for (int k = 0; k < split; ++k) t[k] = partial[size_t(r) * split + k];
    for (int w = 1; w < split; w *= 2)
        for (int k = 0; k < split; k += 2 * w) t[k] = t[k] + t[k + w];
    out[r] = t[0];
}

// Plan of the split per-hit trace.
// (1) hits_offsets_kernel / hits_plan_kernel: each ray's chunk counts become output offsets
//     (exclusive scan along the chunks, starting at the ray's own offset); the packet's running
//     chunk totals are kept (pk_prefix) with its grand total (pk_total).
// (2) hits_assign_kernel, one workgroup: the W launched waves are dealt to the packets in
//     proportion to their hit totals -- K_p = 1 + floor((W - P) H_p / H) -- so that waves, not
//     packets, carry equal work (HEALPix / isotropic bundles: rays along a box diagonal collect
//     1.7x the hits of rays along an axis; with a fixed K the slowest packet set the kernel time
//     at 2.5x the mean wave's).
// (3) hits_bounds_kernel, one wave per packet: its chunks are cut into K_p contiguous ranges of
//     about equal hit totals; wave first_p + k gets {packet, first chunk, end chunk}.
// (1a) one wavefront per RAY: the ray's row of chunk counts (contiguous: coalesced) becomes its
//      row of output offsets.
__global__ __launch_bounds__(256) void hits_offsets_kernel(const int* __restrict__ chunk_counts,
                                                           const int* __restrict__ ray_offsets,
                                                           int n_rays, int n_chunks,
                                                           int* __restrict__ chunk_off)
{
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const int* row = chunk_counts + size_t(ray) * n_chunks;
    int* out = chunk_off + size_t(ray) * n_chunks;
    int carry = ray_offsets[ray];
    for (int c0 = 0; c0 < n_chunks; c0 += 64) {
        const int c = c0 + lane;
        const int v = c < n_chunks ? row[c] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        if (c < n_chunks) out[c] = carry + incl - v;
        carry += __shfl(incl, 63);
    }
}

// (1b) one workgroup per PACKET, one thread per chunk: the packet's hits per chunk (sum over its
//      64 rays, coalesced along the chunks), their running totals and the grand total.
__global__ __launch_bounds__(MAX_HIT_CHUNKS) void hits_plan_kernel(const int* __restrict__ chunk_counts,
                                                                   const uint32_t* __restrict__ perm,
                                                                   int n_rays, int n_chunks,
                                                                   uint32_t* __restrict__ pk_prefix,
                                                                   uint32_t* __restrict__ pk_total)
{
    __shared__ uint32_t s_wave[MAX_HIT_CHUNKS / 64];
    const int packet = blockIdx.x, c = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t t = 0;
    if (c < n_chunks)
        for (int r = 0; r < 64; ++r) {
            const int slot = packet * 64 + r;
            if (slot >= n_rays) break;
            const int ray = perm ? int(perm[slot]) : slot;
            t += uint32_t(chunk_counts[size_t(ray) * n_chunks + c]);
        }
    uint32_t incl = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(incl, o); if (lane >= o) incl += up; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (c < n_chunks) pk_prefix[size_t(packet) * n_chunks + c] = before + incl;   // inclusive
    if (c == n_chunks - 1) pk_total[packet] = before + incl;
}

__global__ __launch_bounds__(1024) void hits_assign_kernel(const uint32_t* __restrict__ pk_total,
                                                           int n_packets, int n_waves, int n_chunks,
                                                           int* __restrict__ pk_first,
                                                           int* __restrict__ pk_parts,
                                                           int* __restrict__ n_used,
                                                           const unsigned long long stage_min_per_packet)
{
    __shared__ unsigned long long s_red[16];
    __shared__ int s_scan[16];
    __shared__ int s_carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long h = 0;
    for (int p = threadIdx.x; p < n_packets; p += blockDim.x) h += pk_total[p];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
    if (lane == 0) s_red[wave] = h;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    unsigned long long H = 0;
    for (int w = 0; w < 16; ++w) H += s_red[w];
    const unsigned long long pool = (unsigned long long)(n_waves > n_packets ? n_waves - n_packets : 0);
    for (int base = 0; base < n_packets; base += blockDim.x) {
        const int p = base + threadIdx.x;
        int k = 0;
        if (p < n_packets) {
            k = 1 + (H ? int(pool * pk_total[p] / H) : 0);
            if (k > n_chunks) k = n_chunks;
        }
        int incl = k;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o); if (lane >= o) incl += up; }
        if (lane == 63) s_scan[wave] = incl;
        __syncthreads();
        int before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_scan[w];
        if (p < n_packets) { pk_first[p] = before + incl - k; pk_parts[p] = k; }
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *n_used = s_carry;
        *reinterpret_cast<unsigned long long*>(n_used + 2) = H;   // the batch's hit total
        // n_used[4]: the per-hit walk's variant -- heavy packets stage their hits in LDS
        n_used[4] = (stage_min_per_packet != 0ull && H / (unsigned long long)n_packets >= stage_min_per_packet) ? 1 : 0;
    }
}

__global__ __launch_bounds__(64) void hits_bounds_kernel(const uint32_t* __restrict__ pk_prefix,
                                                         const uint32_t* __restrict__ pk_total,
                                                         const int* __restrict__ pk_first,
                                                         const int* __restrict__ pk_parts, int n_chunks,
                                                         int4* __restrict__ wave_map)
{
    const int packet = blockIdx.x, lane = threadIdx.x;
    const uint32_t* pre = pk_prefix + size_t(packet) * n_chunks;
    const unsigned long long run = pk_total[packet];
    const int parts = pk_parts[packet], first = pk_first[packet];
    // Boundary j = first chunk whose inclusive prefix reaches j / parts of the total.
    auto boundary = [&](const int j) {
        if (j <= 0) return 0;
        if (j >= parts) return n_chunks;
        const unsigned long long want = (run * (unsigned long long)j + parts - 1) / parts;
        int lo = 0, hi = n_chunks - 1;           // smallest c with pre[c] >= want
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] >= want) hi = mid; else lo = mid + 1;
        }
        return lo + 1 > n_chunks ? n_chunks : lo + 1;   // chunks [.., lo] belong to the parts before
    };
    for (int k = lane; k < parts; k += 64)
        wave_map[first + k] = make_int4(packet, boundary(k), boundary(k + 1), 0);
}

} // namespace
