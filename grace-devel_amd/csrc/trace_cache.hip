// Signatures of the inputs behind the traversal's cached records, for gfx950 (see "cached records
// and their validation" in trace_state.hpp).
//
// One streaming pass over the arrays a trace call was given (HBM-bound: every byte read once,
// 16 B per lane, coalesced) reduces them to 128 bits per group: with the array cut into 16-byte
// chunks (lo, hi) numbered i = 0, 1, ... across the group's arrays,
//     x_i = mix(lo_i ^ i K1),   y_i = mix((hi_i + rotl(x_i, 31)) ^ i K2),   sig = (sum x_i, sum y_i)
// (64-bit wrap-around sums: order-independent, so no atomics and no dependence on scheduling).
// mix is a bijection of its 64-bit argument, so changing any single chunk changes the signature
// with certainty, moving a chunk to another index likewise (the index is mixed in before the
// multiply), and unrelated multi-chunk changes collide with probability 2^-128.  The signature is
// a staleness check of the caller's own arrays, not a defence against an adversary.
#include "trace_state.hpp"

#include <cstdint>

using namespace grace_hip;

namespace {

constexpr int SIG_BLOCK = 256;
constexpr int SIG_GRID = 1024;
constexpr int SIG_GROUPS = 2;

struct SigArgs {
    const uint32_t* base[4];     // arrays as 32-bit words (all inputs are made of 4-byte elements)
    size_t words[4];
    size_t chunk0[4];            // index of the array's first chunk within its group
    int group[4];
    int aligned[4];              // base is 16-byte aligned: whole chunks are loaded as uint4
    int n_arrays;
};

__device__ __forceinline__ unsigned long long mix64(unsigned long long v)
{
    v ^= v >> 32;
    v *= 0xD6E8FEB86659FD93ull;
    v ^= v >> 29;
    return v;
}

__global__ __launch_bounds__(SIG_BLOCK) void signature_kernel(const SigArgs a,
                                                              unsigned long long* __restrict__ partial)
{
    unsigned long long sx[SIG_GROUPS] = { 0ull, 0ull }, sy[SIG_GROUPS] = { 0ull, 0ull };
    const size_t t0 = blockIdx.x * size_t(blockDim.x) + threadIdx.x, stride = size_t(gridDim.x) * blockDim.x;
    for (int k = 0; k < a.n_arrays; ++k) {
        const size_t n_chunks = (a.words[k] + 3) / 4;
        const uint32_t* w = a.base[k];
        unsigned long long ax = 0ull, ay = 0ull;
        for (size_t c = t0; c < n_chunks; c += stride) {
            uint4 q;
            if (a.aligned[k] && 4 * c + 4 <= a.words[k]) {
                q = reinterpret_cast<const uint4*>(w)[c];
            } else {
                q.x = 4 * c + 0 < a.words[k] ? w[4 * c + 0] : 0u;
                q.y = 4 * c + 1 < a.words[k] ? w[4 * c + 1] : 0u;
                q.z = 4 * c + 2 < a.words[k] ? w[4 * c + 2] : 0u;
                q.w = 4 * c + 3 < a.words[k] ? w[4 * c + 3] : 0u;
            }
            const unsigned long long i = a.chunk0[k] + c;
            const unsigned long long lo = (unsigned long long)q.x | ((unsigned long long)q.y << 32);
            const unsigned long long hi = (unsigned long long)q.z | ((unsigned long long)q.w << 32);
            const unsigned long long x = mix64(lo ^ (i * 0x9E3779B97F4A7C15ull));
            const unsigned long long y = mix64((hi + ((x << 31) | (x >> 33))) ^ (i * 0xC2B2AE3D27D4EB4Full));
            ax += x;
            ay += y;
        }
        sx[a.group[k]] += ax;
        sy[a.group[k]] += ay;
    }
    __shared__ unsigned long long s_red[SIG_BLOCK / 64][2 * SIG_GROUPS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < SIG_GROUPS; ++g) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sx[g] += __shfl_xor(sx[g], off);
            sy[g] += __shfl_xor(sy[g], off);
        }
        if (lane == 0) { s_red[wave][2 * g] = sx[g]; s_red[wave][2 * g + 1] = sy[g]; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * SIG_GROUPS) {
        unsigned long long v = 0ull;
        for (int w = 0; w < SIG_BLOCK / 64; ++w) v += s_red[w][threadIdx.x];
        partial[size_t(blockIdx.x) * 2 * SIG_GROUPS + threadIdx.x] = v;
    }
}

struct CheckArgs {
    CacheCtl* ctl[SIG_GROUPS];
    int force[SIG_GROUPS];
    unsigned long long salt[SIG_GROUPS];   // array sizes of the group: a resized array never matches
    uint32_t* rays_ext;
};

__global__ __launch_bounds__(SIG_BLOCK) void signature_check_kernel(const unsigned long long* __restrict__ partial,
                                                                    const CheckArgs a)
{
    __shared__ unsigned long long s_red[SIG_BLOCK / 64][2 * SIG_GROUPS];
    unsigned long long v[2 * SIG_GROUPS] = { 0ull, 0ull, 0ull, 0ull };
    for (int b = threadIdx.x; b < SIG_GRID; b += blockDim.x)
#pragma unroll
        for (int k = 0; k < 2 * SIG_GROUPS; ++k) v[k] += partial[size_t(b) * 2 * SIG_GROUPS + k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 2 * SIG_GROUPS; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
        if (lane == 0) s_red[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < SIG_GROUPS) {
        const int g = threadIdx.x;
        CacheCtl* ctl = a.ctl[g];
        if (ctl) {
            unsigned long long x = a.salt[g], y = ~a.salt[g];
            for (int w = 0; w < SIG_BLOCK / 64; ++w) { x += s_red[w][2 * g]; y += s_red[w][2 * g + 1]; }
            const bool stale = a.force[g] || ctl->sig[0] != x || ctl->sig[1] != y;
            ctl->sig[0] = x;
            ctl->sig[1] = y;
            ctl->stale = stale ? 1u : 0u;
            if (g == 1 && stale && a.rays_ext) {
                // what ray_ext_init_kernel writes: minima at the top of the order, the rest zero
                for (int k = 0; k < 16; ++k) a.rays_ext[k] = k < 6 ? 0xFFFFFFFFu : 0u;
            }
        }
    }
}

} // namespace

namespace grace_hip {

size_t sig_partial_words() { return size_t(SIG_GRID) * 2 * SIG_GROUPS; }

grace_status launch_signatures(const SigRequest& rq, unsigned long long* partial, hipStream_t stream)
{
    SigArgs a = {};
    CheckArgs c = {};
    size_t next_chunk[SIG_GROUPS] = { 0, 0 };
    auto add = [&](const void* p, size_t bytes, int group) {
        if (!p || bytes == 0) return;
        const int k = a.n_arrays++;
        a.base[k] = static_cast<const uint32_t*>(p);
        a.words[k] = bytes / 4;
        a.group[k] = group;
        a.aligned[k] = (reinterpret_cast<uintptr_t>(p) & 15u) == 0 ? 1 : 0;
        a.chunk0[k] = next_chunk[group];
        next_chunk[group] += (a.words[k] + 3) / 4;
        c.salt[group] = c.salt[group] * 0x100000001B3ull + bytes;
    };
    if (rq.scene_ctl) {
        add(rq.prims, rq.prims_bytes, 0);
        add(rq.nodes, rq.nodes_bytes, 0);
        add(rq.leaves, rq.leaves_bytes, 0);
        c.ctl[0] = rq.scene_ctl;
        c.force[0] = rq.scene_force ? 1 : 0;
    }
    if (rq.rays_ctl) {
        add(rq.rays, rq.rays_bytes, 1);
        c.ctl[1] = rq.rays_ctl;
        c.force[1] = rq.rays_force ? 1 : 0;
        c.rays_ext = rq.rays_ext;
    }
    if (!c.ctl[0] && !c.ctl[1]) return GRACE_OK;
    signature_kernel<<<SIG_GRID, SIG_BLOCK, 0, stream>>>(a, partial);
    GRACE_CHECK_LAUNCH();
    signature_check_kernel<<<1, SIG_BLOCK, 0, stream>>>(partial, c);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

} // namespace grace_hip
