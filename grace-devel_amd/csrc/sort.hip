// Stable LSD radix sort of (key, payload) for gfx950 -- the thrust::sort_by_key call sites
// of the reference (include/grace/cuda/build_sph.cuh:46,57,70,81;
// include/grace/cuda/kernels/gen_rays.cuh:483,520,577,615).  Thrust's source is not part of
// the reference; the contract restated here is: ascending keys, equal keys keep input order.
//
// Design (wave64): 8-bit digits.  Per pass
//   1. histogram : 4096-key tiles, per-workgroup 256-bin histogram in LDS -> counts[digit][tile]
//   2. scan      : one exclusive scan over the digit-major count table = global base of
//                  every (digit, tile)
//   3. scatter   : each wave ranks 64 consecutive keys per round with a ballot-based
//                  match (8 ballots -> mask of lanes with the same digit, popcount below the
//                  lane = stable rank), per-wave running digit counters in LDS, then a
//                  cross-wave exclusive prefix per digit and a workgroup scan of the digit
//                  totals; the tile is staged in LDS in digit order and written out slot by
//                  slot, so that stores are contiguous within every digit's run.  (key, source
//                  index) pairs are moved; the payload (16/28/32/36 B) is gathered ONCE at the
//                  end.
// HBM traffic per 32-bit key: histogram 4 B + scatter (8 B in + 8 B out) per pass, plus
// the payload gather (index 4 B + payload in + payload out); for four passes and a 16-B
// payload: 80 + 36 = 116 B/element.
#include "common.hpp"

using namespace grace_hip;

namespace {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_WAVES = SORT_BLOCK / 64;
constexpr int SORT_ROUNDS = 16;                         // rounds of 64 keys per wave
constexpr int SORT_TILE = SORT_BLOCK * SORT_ROUNDS;     // 4096 keys per workgroup
constexpr int RADIX_BITS = 8;
constexpr int RADIX = 1 << RADIX_BITS;

template <typename Key>
__device__ __forceinline__ uint32_t digit_of(Key k, int shift, uint32_t mask)
{
    return static_cast<uint32_t>(k >> shift) & mask;
}

template <typename Key>
__global__ __launch_bounds__(SORT_BLOCK) void sort_hist_kernel(const Key* __restrict__ keys,
                                                               size_t n, int shift, uint32_t mask,
                                                               uint32_t n_tiles,
                                                               uint32_t* __restrict__ counts,
                                                               const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
    __shared__ uint32_t s_hist[RADIX];
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    const size_t tile0 = size_t(blockIdx.x) * SORT_TILE;
#pragma unroll 4
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const size_t i = tile0 + size_t(r) * SORT_BLOCK + threadIdx.x;
        if (i < n) atomicAdd(&s_hist[digit_of(keys[i], shift, mask)], 1u);
    }
    __syncthreads();
    counts[size_t(threadIdx.x) * n_tiles + blockIdx.x] = s_hist[threadIdx.x];
}

// FIRST: the payload index is the element's own position (no index array read).
template <typename Key, bool FIRST>
__global__ __launch_bounds__(SORT_BLOCK) void sort_scatter_kernel(
    const Key* __restrict__ keys_in, const uint32_t* __restrict__ idx_in,
    Key* __restrict__ keys_out, uint32_t* __restrict__ idx_out, size_t n, int shift,
    uint32_t mask, uint32_t n_tiles, const uint32_t* __restrict__ bases,
    const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
    __shared__ uint32_t s_cnt[SORT_WAVES][RADIX];
    for (int k = threadIdx.x; k < SORT_WAVES * RADIX; k += SORT_BLOCK) (&s_cnt[0][0])[k] = 0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const size_t wave0 = size_t(blockIdx.x) * SORT_TILE + size_t(wave) * (64 * SORT_ROUNDS);

    Key key[SORT_ROUNDS];
    uint32_t rank[SORT_ROUNDS]; // digit in the low 8 bits, wave-local rank above

#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const size_t i = wave0 + size_t(r) * 64 + lane;
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : Key(0);
        const uint32_t d = digit_of(key[r], shift, mask);
        // Lanes holding the same digit (invalid lanes match nobody).
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long vote = __ballot(bit);
            same &= bit ? vote : ~vote;
        }
        const uint32_t below = __popcll(same & lt_mask);
        const uint32_t prior = s_cnt[wave][d];
        // One lane per digit group publishes the new running count; the wave executes in
        // lockstep, so every lane has read `prior` before any lane writes.
        if (valid && below == 0) s_cnt[wave][d] = prior + __popcll(same);
        rank[r] = ((prior + below) << RADIX_BITS) | d;
    }
    __syncthreads();

    // Per digit: the waves' counts become exclusive prefixes within the digit; the digits'
    // tile totals are scanned across the workgroup to give each digit's start inside the tile.
    __shared__ uint32_t s_start[RADIX];   // tile-local start of the digit's run
    __shared__ uint32_t s_gbase[RADIX];   // global destination of tile-local slot 0 of the run
    __shared__ uint32_t s_wtot[SORT_WAVES];
    {
        const int d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            const uint32_t c = s_cnt[w][d];
            s_cnt[w][d] = run;
            run += c;
        }
        uint32_t incl = run;              // inclusive scan of the 256 totals: wave, then waves
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) s_wtot[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) before += w < wave ? s_wtot[w] : 0u;
        const uint32_t start = before + incl - run;
        s_start[d] = start;
        s_gbase[d] = bases[size_t(d) * n_tiles + blockIdx.x] - start;
    }
    __syncthreads();

    // Stage the tile in digit order (stable: wave, round, lane = input order), then write it
    // out slot by slot: consecutive threads hit consecutive addresses inside every digit's run
    // (16 elements on average at 4096 keys per tile) instead of one scattered element each.
    __shared__ Key s_key[SORT_TILE];
    __shared__ uint32_t s_idx[SORT_TILE];
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; ++r) {
        const size_t i = wave0 + size_t(r) * 64 + lane;
        if (i < n) {
            const uint32_t d = rank[r] & (RADIX - 1);
            const uint32_t slot = s_start[d] + s_cnt[wave][d] + (rank[r] >> RADIX_BITS);
            s_key[slot] = key[r];
            s_idx[slot] = FIRST ? static_cast<uint32_t>(i) : idx_in[i];
        }
    }
    __syncthreads();
    const size_t tile0 = size_t(blockIdx.x) * SORT_TILE;
    const uint32_t tile_n = uint32_t(n - tile0 < size_t(SORT_TILE) ? n - tile0 : size_t(SORT_TILE));
#pragma unroll 4
    for (uint32_t slot = threadIdx.x; slot < tile_n; slot += SORT_BLOCK) {
        const Key k = s_key[slot];
        const uint32_t dst = s_gbase[digit_of(k, shift, mask)] + slot;
        keys_out[dst] = k;
        idx_out[dst] = s_idx[slot];
    }
}

// out[i] = in[perm[i]], payload as `words` 32-bit words per element.
template <int WORDS>
__global__ __launch_bounds__(256) void gather_words_kernel(const uint32_t* __restrict__ in,
                                                           const uint32_t* __restrict__ perm,
                                                           uint32_t* __restrict__ out, size_t n)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x) {
        const uint32_t* src = in + size_t(perm[i]) * WORDS;
        uint32_t* dst = out + i * WORDS;
        if (WORDS % 4 == 0) {
#pragma unroll
            for (int k = 0; k < WORDS / 4; ++k)
                reinterpret_cast<uint4*>(dst)[k] = reinterpret_cast<const uint4*>(src)[k];
        } else {
#pragma unroll
            for (int k = 0; k < WORDS; ++k) dst[k] = src[k];
        }
    }
}

grace_status gather_payload(const void* d_in, const uint32_t* d_perm, void* d_out, size_t n,
                            int value_bytes, hipStream_t stream)
{
    const uint32_t* in = static_cast<const uint32_t*>(d_in);
    uint32_t* out = static_cast<uint32_t*>(d_out);
    const int grid = stream_grid(n, 256);
    switch (value_bytes / 4) {
    case 1: gather_words_kernel<1><<<grid, 256, 0, stream>>>(in, d_perm, out, n); break;
    case 2: gather_words_kernel<2><<<grid, 256, 0, stream>>>(in, d_perm, out, n); break;
    case 3: gather_words_kernel<3><<<grid, 256, 0, stream>>>(in, d_perm, out, n); break;
    case 4: gather_words_kernel<4><<<grid, 256, 0, stream>>>(in, d_perm, out, n); break;
    case 7: gather_words_kernel<7><<<grid, 256, 0, stream>>>(in, d_perm, out, n); break;
    case 8: gather_words_kernel<8><<<grid, 256, 0, stream>>>(in, d_perm, out, n); break;
    case 9: gather_words_kernel<9><<<grid, 256, 0, stream>>>(in, d_perm, out, n); break;
    default:
        return set_error(GRACE_INVALID_ARGUMENT, __FILE__, __LINE__,
                         "sort: value_bytes must be 4, 8, 12, 16, 28, 32 or 36");
    }
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

size_t sort_ws_bytes_impl(size_t n, int key_bytes, int value_bytes)
{
    const size_t n_tiles = (n + SORT_TILE - 1) / SORT_TILE;
    const size_t n_counts = size_t(RADIX) * n_tiles;
    return Workspace::aligned(n * size_t(key_bytes)) + 2 * Workspace::aligned(n * 4)
        + Workspace::aligned(n_counts * 4) + Workspace::aligned(scan_ws_count(n_counts) * 4)
        + Workspace::aligned(n * size_t(value_bytes)) + 1024;
}

template <typename Key>
grace_status sort_pairs(Key* d_keys, void* d_values, size_t n, int value_bytes, int begin_bit,
                        int end_bit, uint32_t* d_perm_out, hipStream_t stream, bool nested = false,
                        const uint32_t* run_if = nullptr)
{
    // (run_if gates the key / index passes only: the nested callers that use it sort keys alone)
    GRACE_REQUIRE(!run_if || (nested && !d_values), "sort: a gated sort moves no payload");
    GRACE_REQUIRE(n == 0 || d_keys, "sort: null keys");
    GRACE_REQUIRE(n < (size_t(1) << 32), "sort: at most 2^32 - 1 elements");
    GRACE_REQUIRE(begin_bit >= 0 && end_bit <= int(sizeof(Key) * 8) && begin_bit < end_bit,
                  "sort: bad bit range");
    GRACE_REQUIRE(!d_values || (value_bytes > 0 && value_bytes % 4 == 0),
                  "sort: value_bytes must be a positive multiple of 4");
    if (d_values) {   // cached trace records over this array are stale (a hint: they are validated anyway)
        GRACE_TRY(scene_invalidate_if_written(d_values));
        GRACE_TRY(rays_invalidate_if_written(d_values));
    }
    if (n <= 1) {
        if (n == 1 && d_perm_out) GRACE_TRY_HIP(hipMemsetAsync(d_perm_out, 0, 4, stream));
        return GRACE_OK;
    }
    const uint32_t n_tiles = uint32_t((n + SORT_TILE - 1) / SORT_TILE);
    const size_t n_counts = size_t(RADIX) * n_tiles;
    FrameGuard frame;
    if (!nested)
        GRACE_TRY(frame.begin(sort_ws_bytes_impl(n, sizeof(Key), d_values ? value_bytes : 0), stream));
    Key* keys_alt = Workspace::take<Key>(n);
    uint32_t* idx_a = Workspace::take<uint32_t>(n);
    uint32_t* idx_b = Workspace::take<uint32_t>(n);
    uint32_t* counts = Workspace::take<uint32_t>(n_counts);
    uint32_t* scan_ws = Workspace::take<uint32_t>(scan_ws_count(n_counts));

    Key* k_in = d_keys;
    Key* k_out = keys_alt;
    uint32_t* i_in = idx_a;
    uint32_t* i_out = idx_b;
    // A caller that wants the permutation gets it written by the last pass itself (its buffer
    // takes part in the ping-pong so that the final output lands in it): no copy afterwards.
    const int n_passes = (end_bit - begin_bit + RADIX_BITS - 1) / RADIX_BITS;
    if (d_perm_out) {
        if (n_passes % 2) i_out = d_perm_out;     // passes 1, 3, ... write i_out's first value
        else i_in = d_perm_out;                   // passes 2, 4, ... write what starts as i_in
    }
    bool first = true;
    for (int shift = begin_bit; shift < end_bit; shift += RADIX_BITS) {
        const int bits = (end_bit - shift) < RADIX_BITS ? (end_bit - shift) : RADIX_BITS;
        const uint32_t mask = (1u << bits) - 1u;
        sort_hist_kernel<Key><<<n_tiles, SORT_BLOCK, 0, stream>>>(k_in, n, shift, mask, n_tiles,
                                                                 counts, run_if);
        GRACE_CHECK_LAUNCH();
        GRACE_TRY(exclusive_scan_u32(counts, counts, n_counts, scan_ws, nullptr, stream, run_if));
        if (first)
            sort_scatter_kernel<Key, true><<<n_tiles, SORT_BLOCK, 0, stream>>>(
                k_in, nullptr, k_out, i_out, n, shift, mask, n_tiles, counts, run_if);
        else
            sort_scatter_kernel<Key, false><<<n_tiles, SORT_BLOCK, 0, stream>>>(
                k_in, i_in, k_out, i_out, n, shift, mask, n_tiles, counts, run_if);
        GRACE_CHECK_LAUNCH();
        first = false;
        Key* tk = k_in; k_in = k_out; k_out = tk;
        uint32_t* ti = i_in; i_in = i_out; i_out = ti;
    }
    // k_in / i_in now hold the sorted keys and their source indices.
    if (k_in != d_keys && !run_if)     // (a gated sort's keys are scratch: only the permutation is kept)
        GRACE_TRY_HIP(hipMemcpyAsync(d_keys, k_in, n * sizeof(Key), hipMemcpyDeviceToDevice,
                                     stream));
    if (d_values) {
        void* tmp = Workspace::take<char>(n * size_t(value_bytes));
        GRACE_TRY_HIP(hipMemcpyAsync(tmp, d_values, n * size_t(value_bytes),
                                     hipMemcpyDeviceToDevice, stream));
        GRACE_TRY(gather_payload(tmp, i_in, d_values, n, value_bytes, stream));
    }
    if (d_perm_out && i_in != d_perm_out)        // (cannot happen: see the ping-pong set-up)
        GRACE_TRY_HIP(hipMemcpyAsync(d_perm_out, i_in, n * 4, hipMemcpyDeviceToDevice, stream));
    return GRACE_OK;
}

} // namespace

namespace grace_hip {

size_t sort_ws_bytes(size_t n, int key_bytes, int value_bytes)
{
    return sort_ws_bytes_impl(n, key_bytes, value_bytes);
}

grace_status sort_pairs_u32_nested(uint32_t* d_keys, void* d_values, size_t n, int value_bytes,
                                   int begin_bit, int end_bit, uint32_t* d_perm,
                                   hipStream_t stream, const uint32_t* run_if)
{
    return sort_pairs<uint32_t>(d_keys, d_values, n, value_bytes, begin_bit, end_bit, d_perm,
                                stream, true, run_if);
}

grace_status sort_pairs_u64_nested(uint64_t* d_keys, void* d_values, size_t n, int value_bytes,
                                   int begin_bit, int end_bit, uint32_t* d_perm,
                                   hipStream_t stream)
{
    return sort_pairs<uint64_t>(d_keys, d_values, n, value_bytes, begin_bit, end_bit, d_perm,
                                stream, true);
}

} // namespace grace_hip

extern "C" {

grace_status grace_sort_pairs_u32(uint32_t* d_keys, void* d_values, size_t n, int value_bytes,
                                  int begin_bit, int end_bit, uint32_t* d_perm,
                                  grace_stream stream)
{
    return sort_pairs<uint32_t>(d_keys, d_values, n, value_bytes, begin_bit, end_bit, d_perm,
                                as_stream(stream));
}

grace_status grace_sort_pairs_u64(uint64_t* d_keys, void* d_values, size_t n, int value_bytes,
                                  int begin_bit, int end_bit, uint32_t* d_perm,
                                  grace_stream stream)
{
    return sort_pairs<uint64_t>(d_keys, d_values, n, value_bytes, begin_bit, end_bit, d_perm,
                                as_stream(stream));
}

} // extern "C"
