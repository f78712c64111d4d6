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

// blockIdx -> tile such that the workgroups of one XCD (blockIdx % 8) own a contiguous range of
// tiles; launch xcd_grid(n_tiles) workgroups and drop tiles >= n_tiles.
__host__ __device__ __forceinline__ uint32_t xcd_grid(uint32_t n_tiles) { return ((n_tiles + 7u) / 8u) * 8u; }
__device__ __forceinline__ uint32_t xcd_tile(uint32_t block, uint32_t n_tiles)
{
    return (block & 7u) * ((n_tiles + 7u) / 8u) + (block >> 3);
}

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
    // Workgroups are dealt to the 8 XCDs round-robin: consecutive TILES go to one XCD, so that
    // the adjacent runs two neighbouring tiles write for a digit meet in one L2 (-5 % per pass).
    const uint32_t tile = xcd_tile(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const size_t wave0 = size_t(tile) * SORT_TILE + size_t(wave) * (64 * SORT_ROUNDS);

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
        s_gbase[d] = bases[size_t(d) * n_tiles + tile] - start;
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
    const size_t tile0 = size_t(tile) * SORT_TILE;
    const uint32_t tile_n = uint32_t(n - tile0 < size_t(SORT_TILE) ? n - tile0 : size_t(SORT_TILE));
#pragma unroll 4
    for (uint32_t slot = threadIdx.x; slot < tile_n; slot += SORT_BLOCK) {
        const Key k = s_key[slot];
        const uint32_t dst = s_gbase[digit_of(k, shift, mask)] + slot;
        keys_out[dst] = k;
        idx_out[dst] = s_idx[slot];
    }
}


// ---------------------------------------------------------------------------------------------
// Bucket sort (large inputs): ONE most-significant-digit pass that carries the payload, then
// every bucket finished inside one workgroup's LDS.
//   1. bucket_hist    : per 8192-key tile, histogram of the top M bits -> counts[tile][bucket]
//   2. column scan    : global base of every (bucket, tile) -- three launches over the tile-major
//                       table; the middle one also writes the bucket starts and the device
//                       flags "every bucket fits a workgroup" / "one does not"
//   3. bucket_scatter : the tile is ordered by bucket in LDS (stable ballot-rank passes over the
//                       top M bits) and written to scratch run by run -- key, payload and, when
//                       asked for, source index
//   4. bucket_finish  : one workgroup per bucket: keys into LDS, stable LSD passes over the
//                       remaining low bits there, then keys and payload written to the caller's
//                       arrays in final order (the payload is read from the bucket's own window of
//                       scratch, ~80 KB: cache-resident)
// The payload moves twice, coalesced or cache-local both times, instead of once at random
// (the gather of the index sort above: 10^7 random 16-byte reads = 216 us of the 650); the keys
// make 2 trips through HBM instead of 4.  M is the smallest digit for which the MEAN bucket fills
// 60 % of a workgroup's capacity.  A bucket above the capacity (clustered keys) cannot be finished
// in LDS: the flag then turns the two bucket kernels off and the index sort on (gated launches,
// no host round trip), so that case costs the index sort plus the histogram of the top bits.
// HBM traffic per 30-bit key with a 16-byte payload: 4 + (20 + 20) + (20 + 20) = 84 B.

bool g_overflow_hint_on = true;     // grace_sort_set_overflow_hint

constexpr int LS_THREADS = 512;
constexpr int LS_WAVES = LS_THREADS / 64;
constexpr int LS_MAX_MSD_BITS = 12;

// Records per tile / per bucket capacity: 8192 for 32-bit keys with payloads of up to 16 bytes;
// 4096 for 64-bit keys (LDS) and for the wide payloads (28, 32, 36 bytes: the payload of a
// thread's records rides in registers, 8 records x 9 words at most).
constexpr int local_tile(int key_bytes, int words) { return (key_bytes == 8 || words > 4) ? 4096 : 8192; }
template <typename Key, int W> struct LocalSort { static constexpr int TILE = local_tile(int(sizeof(Key)), W); };

// One stable counting pass in LDS over cnt <= TILE (key, src) records: reordered in place by the
// digit key[shift, shift + bits), bits <= 8.  Wave w owns a contiguous, equal share of the
// records (order = wave, round, lane = position); ranks as in sort_scatter_kernel.
template <typename Key, int TILE, int BITS>
__device__ __forceinline__ void lds_radix_pass_b(Key* __restrict__ s_key, uint16_t* __restrict__ s_src,
                                                 uint32_t cnt, int shift, int bits,
                                                 uint16_t* __restrict__ s_cnt,      // [LS_WAVES][256]
                                                 uint32_t* __restrict__ s_start,    // [256]
                                                 uint32_t* __restrict__ s_wtot)     // [4]
{
    constexpr int ROUNDS = TILE / LS_THREADS;
    static_assert(TILE <= 8192 && ROUNDS * 64 <= 2048, "meta word: 13-bit source, 11-bit rank");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t lt_lo = lane < 32 ? (1u << lane) - 1u : 0xFFFFFFFFu;
    const uint32_t lt_hi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
    const uint32_t rounds = (cnt + LS_THREADS - 1) / LS_THREADS;
    const uint32_t per = rounds * 64;
    const uint32_t mask = (1u << bits) - 1u;
    for (int k = threadIdx.x; k < LS_WAVES * 256 / 2; k += LS_THREADS)
        reinterpret_cast<uint32_t*>(s_cnt)[k] = 0;
    __syncthreads();

    Key key[ROUNDS];
    uint32_t meta[ROUNDS];    // source << 19 | rank inside the wave << 8 | digit
    uint16_t* my_cnt = s_cnt + wave * 256;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (uint32_t(r) < rounds) {
            const uint32_t p = wave * per + r * 64 + lane;
            const bool valid = p < cnt;
            const uint32_t pc = valid ? p : 0u;          // (idle lanes read record 0: no exec juggling)
            key[r] = s_key[pc];
            const uint32_t src = s_src[pc];
            const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & mask;
            // lanes of the same digit: per bit one ballot and, per 32-lane half, one three-input
            // boolean (same & ~(vote ^ m), m = all ones where the lane's own bit is set); the
            // ballots of BITS (compile time) bits are independent and issue back to back
            const unsigned long long live = __ballot(valid);
            uint32_t same_lo = uint32_t(live), same_hi = uint32_t(live >> 32);
#pragma unroll
            for (int b = 0; b < BITS; ++b) {
                const uint32_t m = uint32_t(int32_t(d << (31 - b)) >> 31);       // v_bfe_i32
                const unsigned long long vote = __builtin_amdgcn_uicmp(m, 0u, 33);   // v_cmp_ne_u32 -> SGPR pair
                same_lo &= ~(uint32_t(vote) ^ m);
                same_hi &= ~(uint32_t(vote >> 32) ^ m);
            }
            const uint32_t below = __popc(same_lo & lt_lo) + __popc(same_hi & lt_hi);
            const uint32_t prior = my_cnt[d];
            if (valid && below == 0) my_cnt[d] = uint16_t(prior + __popc(same_lo) + __popc(same_hi));
            meta[r] = (src << 19) | ((prior + below) << 8) | d;
        }
    }
    __syncthreads();
    uint32_t run = 0, incl = 0;
    if (threadIdx.x < 256) {
        const int d = threadIdx.x;
#pragma unroll
        for (int w = 0; w < LS_WAVES; ++w) {
            const uint32_t c = s_cnt[w * 256 + d];
            s_cnt[w * 256 + d] = uint16_t(run);
            run += c;
        }
        incl = run;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) s_wtot[wave] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) before += w < wave ? s_wtot[w] : 0u;
        s_start[threadIdx.x] = before + incl - run;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (uint32_t(r) < rounds) {
            const uint32_t p = wave * per + r * 64 + lane;
            if (p < cnt) {
                const uint32_t d = meta[r] & 255u;
                const uint32_t slot = s_start[d] + my_cnt[d] + ((meta[r] >> 8) & 2047u);
                s_key[slot] = key[r];
                s_src[slot] = uint16_t(meta[r] >> 19);
            }
        }
    }
    __syncthreads();
}

template <typename Key, int TILE>
__device__ __forceinline__ void lds_radix_pass(Key* __restrict__ s_key, uint16_t* __restrict__ s_src,
                                               uint32_t cnt, int shift, int bits, uint16_t* __restrict__ s_cnt,
                                               uint32_t* __restrict__ s_start, uint32_t* __restrict__ s_wtot)
{
    if (bits > 4) lds_radix_pass_b<Key, TILE, 8>(s_key, s_src, cnt, shift, bits, s_cnt, s_start, s_wtot);
    else lds_radix_pass_b<Key, TILE, 4>(s_key, s_src, cnt, shift, bits, s_cnt, s_start, s_wtot);
}

template <typename Key>
__global__ __launch_bounds__(LS_THREADS) void bucket_hist_kernel(const Key* __restrict__ keys, size_t n,
                                                                 int shift, int msd_bits, int TILE,
                                                                 uint32_t* __restrict__ counts)   // [tile][bin]
{
    __shared__ uint32_t s_hist[1 << LS_MAX_MSD_BITS];
    const uint32_t bins = 1u << msd_bits;
    for (uint32_t k = threadIdx.x; k < bins; k += LS_THREADS) s_hist[k] = 0;
    __syncthreads();
    const size_t tile0 = size_t(blockIdx.x) * TILE;
#pragma unroll 4
    for (int r = 0; r < TILE / LS_THREADS; ++r) {
        const size_t i = tile0 + size_t(r) * LS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&s_hist[static_cast<uint32_t>(keys[i] >> shift) & (bins - 1u)], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < bins; k += LS_THREADS)
        counts[size_t(blockIdx.x) * bins + k] = s_hist[k];
}

// The table is tile-major (a tile's row is written and read coalesced); the bases wanted are the
// exclusive prefix in (bin, tile) order.  Three launches: per bin, the sums over BUCKET_SEGS
// segments of tiles; one workgroup turns them into each (segment, bin)'s base and derives the
// bucket bounds and the two flags; per bin, the running base down each segment's tiles.
constexpr int BUCKET_SEGS = 8;
constexpr int BUCKET_COL_BLOCK = 64;

__global__ __launch_bounds__(BUCKET_COL_BLOCK) void bucket_colsum_kernel(const uint32_t* __restrict__ counts, uint32_t bins,
                                                            uint32_t n_tiles, uint32_t* __restrict__ seg_sum)
{
    const uint32_t bin = blockIdx.x * BUCKET_COL_BLOCK + threadIdx.x;
    const uint32_t per = (n_tiles + BUCKET_SEGS - 1) / BUCKET_SEGS;
    const uint32_t t0 = blockIdx.y * per;
    const uint32_t t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    if (bin >= bins) return;
    uint32_t sum = 0;
#pragma unroll 8
    for (uint32_t t = t0; t < t1; ++t) sum += counts[size_t(t) * bins + bin];
    seg_sum[blockIdx.y * bins + bin] = sum;
}

// seg_sum[seg][bin] -> base of (bin, first tile of seg); bounds[b] = first element of bucket b
// (bounds[bins] = n); ctl[0] = 1 when every bucket fits a workgroup (the bucket kernels run),
// ctl[1] = 1 when one does not (the index sort runs).  One workgroup of 1024.
__global__ __launch_bounds__(1024) void bucket_bases_kernel(uint32_t* __restrict__ seg_sum, uint32_t n,
                                                            uint32_t bins, uint32_t cap,
                                                            uint32_t* __restrict__ bounds,
                                                            uint32_t* __restrict__ ctl,
                                                            uint32_t* __restrict__ overflow_hint)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_over;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_over = 0;
    // thread t owns bins [t * k, (t + 1) * k), k = bins / 1024 (or one bin for the first `bins` threads)
    const uint32_t k = bins > 1024 ? bins / 1024 : 1;
    const uint32_t b0 = threadIdx.x * k;
    uint32_t tot[4] = { 0, 0, 0, 0 };      // k <= 4 (LS_MAX_MSD_BITS = 12)
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t b = b0 + j;
        if (j < k && b < bins) {
            uint32_t v[BUCKET_SEGS];         // (independent loads, all in flight at once)
#pragma unroll
            for (int sg = 0; sg < BUCKET_SEGS; ++sg) v[sg] = seg_sum[sg * bins + b];
#pragma unroll
            for (int sg = 0; sg < BUCKET_SEGS; ++sg) tot[j] += v[sg];
        }
        mine += tot[j];
    }
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    uint32_t run = before + incl - mine;
    uint32_t over = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t b = b0 + j;
        if (j < k && b < bins) {
            bounds[b] = run;
            over |= tot[j] > cap ? 1u : 0u;
            uint32_t v[BUCKET_SEGS];
#pragma unroll
            for (int sg = 0; sg < BUCKET_SEGS; ++sg) v[sg] = seg_sum[sg * bins + b];
            uint32_t r2 = run;
#pragma unroll
            for (int sg = 0; sg < BUCKET_SEGS; ++sg) {
                seg_sum[sg * bins + b] = r2;
                r2 += v[sg];
            }
            run += tot[j];
        }
    }
    if (threadIdx.x == 0) bounds[bins] = n;
    if (over) atomicOr(&s_over, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl[0] = s_over ? 0u : 1u; ctl[1] = s_over ? 1u : 0u;
        if (overflow_hint) *overflow_hint = s_over ? 1u : 0u;      // (pinned host word: see Context)
    }
}

__global__ __launch_bounds__(BUCKET_COL_BLOCK) void bucket_colscan_kernel(uint32_t* __restrict__ counts, uint32_t bins,
                                                             uint32_t n_tiles, const uint32_t* __restrict__ seg_base)
{
    const uint32_t bin = blockIdx.x * BUCKET_COL_BLOCK + threadIdx.x;
    const uint32_t per = (n_tiles + BUCKET_SEGS - 1) / BUCKET_SEGS;
    const uint32_t t0 = blockIdx.y * per;
    const uint32_t t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    if (bin >= bins) return;
    uint32_t run = seg_base[blockIdx.y * bins + bin];
#pragma unroll 4
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t v = counts[size_t(t) * bins + bin];
        counts[size_t(t) * bins + bin] = run;
        run += v;
    }
}

// The payload's trip through a bucket kernel.  Reading it at the slots' sources (random 16-byte
// reads inside the tile's 128 KB window) was measured at 200 us per kernel: with 64 tiles in
// flight per XCD the windows (8 MB) outlive their stay in the 4 MB L2 and every sector comes in up
// to four times.  Instead each thread reads the payload of ITS elements (source order: coalesced,
// issued before the LDS passes and so hidden behind them) into registers, learns their final
// slots from the inverse of the sorted source list, and the workgroup transposes through a
// 16 KB LDS window, 1024 slots at a time, so that the stores are in slot order too.
constexpr int LS_CHUNK = 1024;

template <int W> struct Payload { uint32_t w[W > 0 ? W : 1]; };

template <int W>
__device__ __forceinline__ Payload<W> load_payload(const uint32_t* __restrict__ src)
{
    Payload<W> v;
    if constexpr (W == 4 || W == 8) {
#pragma unroll
        for (int h = 0; h < W / 4; ++h) {
            const uint4 q = reinterpret_cast<const uint4*>(src)[h];
            v.w[4 * h + 0] = q.x; v.w[4 * h + 1] = q.y; v.w[4 * h + 2] = q.z; v.w[4 * h + 3] = q.w;
        }
    } else if constexpr (W == 2) {
        const uint2 q = *reinterpret_cast<const uint2*>(src);
        v.w[0] = q.x; v.w[1] = q.y;
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) v.w[k] = src[k];
    }
    return v;
}

template <int W>
__device__ __forceinline__ void store_payload(uint32_t* __restrict__ dst, const Payload<W>& v)
{
    if constexpr (W == 4 || W == 8) {
#pragma unroll
        for (int h = 0; h < W / 4; ++h)
            reinterpret_cast<uint4*>(dst)[h] = make_uint4(v.w[4 * h], v.w[4 * h + 1], v.w[4 * h + 2], v.w[4 * h + 3]);
    } else if constexpr (W == 2) {
        *reinterpret_cast<uint2*>(dst) = make_uint2(v.w[0], v.w[1]);
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) dst[k] = v.w[k];
    }
}

// s_src[slot] = source of the slot  ->  s_src[source] = slot (through s_tmp, >= cnt uint16)
__device__ __forceinline__ void invert_sources(uint16_t* __restrict__ s_src, uint16_t* __restrict__ s_tmp,
                                               uint32_t cnt)
{
    for (uint32_t p = threadIdx.x; p < cnt; p += LS_THREADS) s_tmp[s_src[p]] = uint16_t(p);
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < cnt; p += LS_THREADS) s_src[p] = s_tmp[p];
    __syncthreads();
}

template <typename Key, int W>
__global__ __launch_bounds__(LS_THREADS, 4) void bucket_scatter_kernel(
    const Key* __restrict__ keys, const uint32_t* __restrict__ vals, size_t n, int shift,
    int msd_bits, uint32_t n_tiles, const uint32_t* __restrict__ bases, Key* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, uint32_t* __restrict__ idx_out, const uint32_t* __restrict__ run_if)
{
    if (*run_if == 0u) return;
    constexpr int TILE = LocalSort<Key, W>::TILE;
    constexpr int ROUNDS = TILE / LS_THREADS;
    constexpr int CHUNK = W > 4 ? ((LS_CHUNK * 4) / W) / 64 * 64 : LS_CHUNK;   // slots of the 16 KB window
    __shared__ Key s_key[TILE];
    __shared__ uint16_t s_src[TILE];
    __shared__ uint16_t s_cnt[LS_WAVES * 256];
    __shared__ uint32_t s_start[256];
    __shared__ uint32_t s_wtot[4];
    __shared__ __attribute__((aligned(16))) uint32_t s_chunk[W > 0 ? LS_CHUNK * 4 : 4];   // (also: TILE uint16 of scratch)
    extern __shared__ uint32_t s_gbase[];    // [1 << msd_bits]: global slot of the tile's slot 0 of the run
    static_assert(LS_CHUNK * 4 * 4 >= TILE * 2 && CHUNK >= 64, "the inverse map borrows the chunk window");

    const uint32_t tile = xcd_tile(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const size_t tile0 = size_t(tile) * TILE;
    const uint32_t cnt = uint32_t(n - tile0 < size_t(TILE) ? n - tile0 : size_t(TILE));
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const uint32_t p = threadIdx.x + k * LS_THREADS;
        if (p < cnt) {
            s_key[p] = keys[tile0 + p];
            s_src[p] = uint16_t(p);
        }
    }
    __syncthreads();
    if (msd_bits > 8) lds_radix_pass<Key, TILE>(s_key, s_src, cnt, shift, msd_bits - 8, s_cnt, s_start, s_wtot);
    {
        const int hi_bits = msd_bits > 8 ? 8 : msd_bits;
        lds_radix_pass<Key, TILE>(s_key, s_src, cnt, shift + msd_bits - hi_bits, hi_bits, s_cnt, s_start, s_wtot);
    }
    // (the payload is read only now: held across the passes, its 64 registers made them spill)
    Payload<W> pay[ROUNDS];
    if constexpr (W > 0) {
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const uint32_t p = threadIdx.x + k * LS_THREADS;
            if (p < cnt) pay[k] = load_payload<W>(vals + (tile0 + p) * W);
        }
    }
    const uint32_t dmask = (1u << msd_bits) - 1u;
    for (uint32_t p = threadIdx.x; p < cnt; p += LS_THREADS) {
        const uint32_t d = static_cast<uint32_t>(s_key[p] >> shift) & dmask;
        if (p == 0 || (static_cast<uint32_t>(s_key[p - 1] >> shift) & dmask) != d)
            s_gbase[d] = bases[size_t(tile) * (dmask + 1u) + d] - p;
    }
    __syncthreads();
    // keys (and source indices) out in slot order; each slot's destination kept for the payload
    uint32_t dst_of[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const uint32_t p = threadIdx.x + k * LS_THREADS;
        if (p < cnt) {
            const Key key = s_key[p];
            dst_of[k] = s_gbase[static_cast<uint32_t>(key >> shift) & dmask] + p;
            keys_out[dst_of[k]] = key;
            if (idx_out) idx_out[dst_of[k]] = uint32_t(tile0 + s_src[p]);
        }
    }
    if constexpr (W > 0) {
        __syncthreads();
        uint32_t* s_dst = reinterpret_cast<uint32_t*>(s_key);
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const uint32_t p = threadIdx.x + k * LS_THREADS;
            if (p < cnt) s_dst[p] = dst_of[k];
        }
        invert_sources(s_src, reinterpret_cast<uint16_t*>(s_chunk), cnt);   // (syncs)
        uint32_t slot_of[ROUNDS];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const uint32_t p = threadIdx.x + k * LS_THREADS;
            slot_of[k] = p < cnt ? s_src[p] : 0xFFFFFFFFu;
        }
        for (uint32_t c0 = 0; c0 < cnt; c0 += CHUNK) {
#pragma unroll
            for (int k = 0; k < ROUNDS; ++k) {
                const uint32_t rel = slot_of[k] - c0;
                if (rel < uint32_t(CHUNK)) store_payload<W>(s_chunk + rel * W, pay[k]);
            }
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < uint32_t(CHUNK) && c0 + j < cnt; j += LS_THREADS)
                store_payload<W>(vals_out + size_t(s_dst[c0 + j]) * W, load_payload<W>(s_chunk + j * W));
            __syncthreads();
        }
    }
}

template <typename Key, int W>
__global__ __launch_bounds__(LS_THREADS, 4) void bucket_finish_kernel(
    const Key* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, const uint32_t* __restrict__ idx_in,
    int begin_bit, int low_bits, const uint32_t* __restrict__ bounds,
    Key* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t* __restrict__ idx_out,
    const uint32_t* __restrict__ run_if)
{
    if (*run_if == 0u) return;
    constexpr int TILE = LocalSort<Key, W>::TILE;
    constexpr int ROUNDS = TILE / LS_THREADS;
    constexpr int CHUNK = W > 4 ? ((LS_CHUNK * 4) / W) / 64 * 64 : LS_CHUNK;   // slots of the 16 KB window
    __shared__ Key s_key[TILE];
    __shared__ uint16_t s_src[TILE];
    __shared__ uint16_t s_cnt[LS_WAVES * 256];
    __shared__ uint32_t s_start[256];
    __shared__ uint32_t s_wtot[4];
    __shared__ __attribute__((aligned(16))) uint32_t s_chunk[W > 0 ? LS_CHUNK * 4 : 4];
    const uint32_t first = bounds[blockIdx.x];
    const uint32_t cnt = bounds[blockIdx.x + 1] - first;
    if (cnt == 0 || cnt > uint32_t(TILE)) return;     // (cnt > TILE cannot happen when *run_if is set)
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const uint32_t p = threadIdx.x + k * LS_THREADS;
        if (p < cnt) {
            s_key[p] = keys_in[first + p];
            s_src[p] = uint16_t(p);
        }
    }
    __syncthreads();
    // Keys with more than 24 bits below the bucket digit (63-bit Morton keys: 51): the passes cover
    // the TOP 24 of them; what is left below decides only between records that agree on everything
    // above -- a handful of pairs among 10^7 spread keys -- and is settled by a stable insertion
    // sort of those runs, one thread per run.  A bucket with a run of more than TIE_RUN_MAX records
    // (keys that differ in their low bits only) is sorted over all its bits instead: the passes are
    // stable, so starting from the present order is as good as starting from the input's.
    constexpr int TIE_RUN_MAX = 8;
    const int tie_bits = low_bits > 24 ? low_bits - 24 : 0;
    for (int done = tie_bits; done < low_bits; done += 8) {
        const int bits = low_bits - done < 8 ? low_bits - done : 8;
        lds_radix_pass<Key, TILE>(s_key, s_src, cnt, begin_bit + done, bits, s_cnt, s_start, s_wtot);
    }
    if (tie_bits > 0) {
        __shared__ uint32_t s_long_run;
        if (threadIdx.x == 0) s_long_run = 0;
        __syncthreads();
        const int above = begin_bit + tie_bits;                         // first bit the passes covered
        const Key tie_mask = (Key(1) << tie_bits) - Key(1);
        // (bits above the bucket digit are not part of the key: the digit's own bits and everything
        // the passes covered are compared; records of one bucket agree on the digit anyway)
        const int top = begin_bit + low_bits;
        auto prefix = [&](const Key k) { return (k >> above) & ((Key(1) << (top - above)) - Key(1)); };
        for (uint32_t p = threadIdx.x; p < cnt; p += LS_THREADS) {
            const Key mine = prefix(s_key[p]);
            const bool head = (p == 0 || prefix(s_key[p - 1]) != mine) && p + 1 < cnt && prefix(s_key[p + 1]) == mine;
            if (head) {
                uint32_t e = p + 1;
                while (e < cnt && e - p <= uint32_t(TIE_RUN_MAX) && prefix(s_key[e]) == mine) ++e;
                if (e - p > uint32_t(TIE_RUN_MAX)) {
                    s_long_run = 1u;            // (benign race: every writer writes 1)
                } else {
                    // stable insertion sort of [p, e) by the low bits (this thread owns the run)
                    for (uint32_t i = p + 1; i < e; ++i) {
                        const Key ki = s_key[i];
                        const uint16_t si = s_src[i];
                        const Key li = (ki >> begin_bit) & tie_mask;
                        uint32_t j = i;
                        while (j > p && ((s_key[j - 1] >> begin_bit) & tie_mask) > li) {
                            s_key[j] = s_key[j - 1];
                            s_src[j] = s_src[j - 1];
                            --j;
                        }
                        s_key[j] = ki;
                        s_src[j] = si;
                    }
                }
            }
        }
        __syncthreads();
        if (s_long_run) {
            for (int done = 0; done < low_bits; done += 8) {
                const int bits = low_bits - done < 8 ? low_bits - done : 8;
                lds_radix_pass<Key, TILE>(s_key, s_src, cnt, begin_bit + done, bits, s_cnt, s_start, s_wtot);
            }
        }
    }
    Payload<W> pay[ROUNDS];
    if constexpr (W > 0) {
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const uint32_t p = threadIdx.x + k * LS_THREADS;
            if (p < cnt) pay[k] = load_payload<W>(vals_in + size_t(first + p) * W);
        }
    }
    for (uint32_t p = threadIdx.x; p < cnt; p += LS_THREADS) {
        keys_out[size_t(first) + p] = s_key[p];
        if (idx_out) idx_out[size_t(first) + p] = idx_in[size_t(first) + s_src[p]];
    }
    if constexpr (W > 0) {
        __syncthreads();
        invert_sources(s_src, reinterpret_cast<uint16_t*>(s_chunk), cnt);   // (syncs)
        uint32_t slot_of[ROUNDS];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const uint32_t p = threadIdx.x + k * LS_THREADS;
            slot_of[k] = p < cnt ? s_src[p] : 0xFFFFFFFFu;
        }
        for (uint32_t c0 = 0; c0 < cnt; c0 += CHUNK) {
#pragma unroll
            for (int k = 0; k < ROUNDS; ++k) {
                const uint32_t rel = slot_of[k] - c0;
                if (rel < uint32_t(CHUNK)) store_payload<W>(s_chunk + rel * W, pay[k]);
            }
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < uint32_t(CHUNK) && c0 + j < cnt; j += LS_THREADS)
                store_payload<W>(vals_out + (size_t(first) + c0 + j) * W, load_payload<W>(s_chunk + j * W));
            __syncthreads();
        }
    }
}

// out = in (words), gated: the copies of the gated index sort
__global__ __launch_bounds__(256) void copy_words_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                         size_t n_words, const uint32_t* __restrict__ run_if)
{
    if (*run_if == 0u) return;
    const size_t tid = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0) {
        const size_t n4 = n_words / 4;
        for (size_t i = tid; i < n4; i += stride)
            reinterpret_cast<uint4*>(out)[i] = reinterpret_cast<const uint4*>(in)[i];
        for (size_t i = n4 * 4 + tid; i < n_words; i += stride) out[i] = in[i];
    } else {
        for (size_t i = tid; i < n_words; i += stride) out[i] = in[i];
    }
}

// out[i] = in[perm[i]], payload as `words` 32-bit words per element.
template <int WORDS>
__global__ __launch_bounds__(256) void gather_words_kernel(const uint32_t* __restrict__ in,
                                                           const uint32_t* __restrict__ perm,
                                                           uint32_t* __restrict__ out, size_t n,
                                                           const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
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
                            int value_bytes, hipStream_t stream, const uint32_t* run_if = nullptr)
{
    const uint32_t* in = static_cast<const uint32_t*>(d_in);
    uint32_t* out = static_cast<uint32_t*>(d_out);
    const int grid = stream_grid(n, 256);
    switch (value_bytes / 4) {
    case 1: gather_words_kernel<1><<<grid, 256, 0, stream>>>(in, d_perm, out, n, run_if); break;
    case 2: gather_words_kernel<2><<<grid, 256, 0, stream>>>(in, d_perm, out, n, run_if); break;
    case 3: gather_words_kernel<3><<<grid, 256, 0, stream>>>(in, d_perm, out, n, run_if); break;
    case 4: gather_words_kernel<4><<<grid, 256, 0, stream>>>(in, d_perm, out, n, run_if); break;
    case 7: gather_words_kernel<7><<<grid, 256, 0, stream>>>(in, d_perm, out, n, run_if); break;
    case 8: gather_words_kernel<8><<<grid, 256, 0, stream>>>(in, d_perm, out, n, run_if); break;
    case 9: gather_words_kernel<9><<<grid, 256, 0, stream>>>(in, d_perm, out, n, run_if); break;
    default:
        return set_error(GRACE_INVALID_ARGUMENT, __FILE__, __LINE__,
                         "sort: value_bytes must be 4, 8, 12, 16, 28, 32 or 36");
    }
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

template <typename Key>
bool bucket_plan(size_t n, int bits, int words, int& msd_bits)
{
    if (words > 9) return false;
    const size_t TILE = size_t(local_tile(int(sizeof(Key)), words));
    // (measured, 30-bit keys + 16 B: equal at 2^17 elements, 0.127 against 0.164 ms at 2^18, 0.131 /
    // 0.190 at 2^20, 0.40 / 0.61 at 10^7; 16-bit keys -- two index passes -- are a draw at any size)
    if (n < (size_t(1) << 18) || bits <= 16) return false;
    int m = 1;
    while (m < LS_MAX_MSD_BITS && (n >> m) * 10 > TILE * 6) ++m;
    if ((n >> m) * 4 > TILE * 3) return false;      // mean bucket above 75 % of a workgroup: too many would overflow
    if (m > bits) return false;                     // fewer key values than buckets needed
    if (bits - m > 24 && sizeof(Key) < 8) return false;   // (cannot happen: 32-bit keys leave <= 31 - m bits)
    msd_bits = m;
    return true;
}

size_t sort_ws_bytes_impl(size_t n, int key_bytes, int value_bytes)
{
    const size_t n_tiles = (n + SORT_TILE - 1) / SORT_TILE;
    const size_t n_counts = size_t(RADIX) * n_tiles;
    const size_t tile_b = 4096;      // (the smallest tile of the bucket sort: most tiles)
    const size_t n_counts_b = (size_t(1) << LS_MAX_MSD_BITS) * ((n + tile_b - 1) / tile_b);
    return 2 * Workspace::aligned(n * size_t(key_bytes)) + 3 * Workspace::aligned(n * 4)
        + Workspace::aligned(n * size_t(value_bytes)) + Workspace::aligned(n_counts * 4) + Workspace::aligned(scan_ws_count(n_counts) * 4)
        + Workspace::aligned(n_counts_b * 4) + Workspace::aligned((size_t(BUCKET_SEGS) << LS_MAX_MSD_BITS) * 4)
        + Workspace::aligned(((size_t(1) << LS_MAX_MSD_BITS) + 8) * 4)
        + Workspace::aligned(n * size_t(value_bytes)) + 1024;
}

// The LSD passes over key bits [begin_bit, end_bit): (k_in, i_in) -> ... ; on return k_in / i_in
// name the buffers that hold the sorted keys and their source indices (first pass: index = position).
template <typename Key>
grace_status lsd_passes(Key*& k_in, Key*& k_out, uint32_t*& i_in, uint32_t*& i_out, size_t n,
                        int begin_bit, int end_bit, uint32_t* counts, uint32_t* scan_ws,
                        hipStream_t stream, const uint32_t* run_if)
{
    const uint32_t n_tiles = uint32_t((n + SORT_TILE - 1) / SORT_TILE);
    const size_t n_counts = size_t(RADIX) * n_tiles;
    bool first = true;
    for (int shift = begin_bit; shift < end_bit; shift += RADIX_BITS) {
        const int bits = (end_bit - shift) < RADIX_BITS ? (end_bit - shift) : RADIX_BITS;
        const uint32_t mask = (1u << bits) - 1u;
        sort_hist_kernel<Key><<<n_tiles, SORT_BLOCK, 0, stream>>>(k_in, n, shift, mask, n_tiles,
                                                                 counts, run_if);
        GRACE_CHECK_LAUNCH();
        GRACE_TRY(exclusive_scan_u32(counts, counts, n_counts, scan_ws, nullptr, stream, run_if));
        if (first)
            sort_scatter_kernel<Key, true><<<xcd_grid(n_tiles), SORT_BLOCK, 0, stream>>>(
                k_in, nullptr, k_out, i_out, n, shift, mask, n_tiles, counts, run_if);
        else
            sort_scatter_kernel<Key, false><<<xcd_grid(n_tiles), SORT_BLOCK, 0, stream>>>(
                k_in, i_in, k_out, i_out, n, shift, mask, n_tiles, counts, run_if);
        GRACE_CHECK_LAUNCH();
        first = false;
        Key* tk = k_in; k_in = k_out; k_out = tk;
        uint32_t* ti = i_in; i_in = i_out; i_out = ti;
    }
    return GRACE_OK;
}

// Large inputs: see "Bucket sort" above.  Launches the bucket kernels, gated by the device flag
// "every bucket fits"; *slow_flag (device) is the complementary flag for the index sort.
template <typename Key>
grace_status sort_pairs_bucketed(Key* d_keys, void* d_values, size_t n, int value_bytes, int begin_bit,
                                 int end_bit, int msd_bits, uint32_t* d_perm_out, hipStream_t stream,
                                 const uint32_t** slow_flag, hipStream_t* side)
{
    const int words = d_values ? value_bytes / 4 : 0;
    const size_t TILE = size_t(local_tile(int(sizeof(Key)), words));
    const uint32_t bins = 1u << msd_bits;
    const uint32_t n_tiles = uint32_t((n + TILE - 1) / TILE);
    const size_t n_counts = size_t(bins) * n_tiles;
    const int shift = end_bit - msd_bits;

    Key* s_keys = Workspace::take<Key>(n);
    uint32_t* s_vals = words ? Workspace::take<uint32_t>(n * size_t(words)) : nullptr;
    uint32_t* s_idx = d_perm_out ? Workspace::take<uint32_t>(n) : nullptr;
    uint32_t* counts = Workspace::take<uint32_t>(n_counts);
    uint32_t* seg_sum = Workspace::take<uint32_t>(size_t(BUCKET_SEGS) * bins);
    uint32_t* bounds = Workspace::take<uint32_t>(bins + 8);
    uint32_t* ctl = bounds + bins + 2;
    *slow_flag = ctl + 1;
    *side = nullptr;

    bucket_hist_kernel<Key><<<n_tiles, LS_THREADS, 0, stream>>>(d_keys, n, shift, msd_bits, int(TILE), counts);
    GRACE_CHECK_LAUNCH();
    const dim3 col_grid((bins + BUCKET_COL_BLOCK - 1) / BUCKET_COL_BLOCK, BUCKET_SEGS);
    bucket_colsum_kernel<<<col_grid, BUCKET_COL_BLOCK, 0, stream>>>(counts, bins, n_tiles, seg_sum);
    GRACE_CHECK_LAUNCH();
    bucket_bases_kernel<<<1, 1024, 0, stream>>>(seg_sum, uint32_t(n), bins, uint32_t(TILE), bounds, ctl,
                                                sort_overflow_word(nullptr));
    GRACE_CHECK_LAUNCH();
    bucket_colscan_kernel<<<col_grid, BUCKET_COL_BLOCK, 0, stream>>>(counts, bins, n_tiles, seg_sum);
    GRACE_CHECK_LAUNCH();
    GRACE_TRY(side_fork(stream, side));     // the gated index sort needs the flag only
    const uint32_t* v_in = static_cast<const uint32_t*>(d_values);
    uint32_t* v_out = static_cast<uint32_t*>(d_values);
    const size_t dyn = size_t(bins) * 4;
#define GRACE_BUCKET_LAUNCH(W)                                                                           \
    bucket_scatter_kernel<Key, W><<<xcd_grid(n_tiles), LS_THREADS, dyn, stream>>>(                        \
        d_keys, v_in, n, shift, msd_bits, n_tiles, counts, s_keys, s_vals, s_idx, ctl);                   \
    bucket_finish_kernel<Key, W><<<bins, LS_THREADS, 0, stream>>>(                                        \
        s_keys, s_vals, s_idx, begin_bit, shift - begin_bit, bounds, d_keys, v_out, d_perm_out, ctl)
    switch (words) {
    case 0: GRACE_BUCKET_LAUNCH(0); break;
    case 1: GRACE_BUCKET_LAUNCH(1); break;
    case 2: GRACE_BUCKET_LAUNCH(2); break;
    case 3: GRACE_BUCKET_LAUNCH(3); break;
    case 4: GRACE_BUCKET_LAUNCH(4); break;
    case 7: GRACE_BUCKET_LAUNCH(7); break;
    case 8: GRACE_BUCKET_LAUNCH(8); break;
    default: GRACE_BUCKET_LAUNCH(9); break;
    }
#undef GRACE_BUCKET_LAUNCH
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

grace_status copy_gated(const void* in, void* out, size_t bytes, const uint32_t* gate, hipStream_t stream)
{
    if (!gate) {
        GRACE_TRY_HIP(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, stream));
        return GRACE_OK;
    }
    copy_words_kernel<<<stream_grid(bytes / 16 + 1, 256, 2), 256, 0, stream>>>(
        static_cast<const uint32_t*>(in), static_cast<uint32_t*>(out), bytes / 4, gate);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
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
    if (d_values) {
        const int w = value_bytes / 4;
        GRACE_REQUIRE((w >= 1 && w <= 4) || (w >= 7 && w <= 9),
                      "sort: value_bytes must be 4, 8, 12, 16, 28, 32 or 36");
    }
    if (d_values) {   // cached trace records over this array are stale (a hint: they are validated anyway)
        GRACE_TRY(scene_invalidate_if_written(d_values));
        GRACE_TRY(rays_invalidate_if_written(d_values));
    }
    if (n <= 1) {
        if (n == 1 && d_perm_out) GRACE_TRY_HIP(hipMemsetAsync(d_perm_out, 0, 4, stream));
        return GRACE_OK;
    }
    FrameGuard frame;
    if (!nested)
        GRACE_TRY(frame.begin(sort_ws_bytes_impl(n, sizeof(Key), d_values ? value_bytes : 0), stream));

    // Large inputs: the bucket sort; the index sort below then runs only if a bucket overflowed
    // (all of its launches gated by the device flag -- no host round trip -- and enqueued on the
    // context's side stream, so that their ~20 empty launches pass beside the bucket kernels
    // instead of after them).
    const uint32_t* gate = run_if;
    // (joins the side stream back into the call's stream on every way out, errors included: the
    // frame's fence is recorded on the call's stream only)
    struct SideJoin {
        hipStream_t call_stream = nullptr;
        bool forked = false;
        grace_status join() { const bool f = forked; forked = false; return f ? side_join(call_stream) : GRACE_OK; }
        ~SideJoin() { (void)join(); }
    } side_guard;
    side_guard.call_stream = stream;
    int msd_bits = 0;
    bool try_buckets = !run_if && bucket_plan<Key>(n, end_bit - begin_bit, d_values ? value_bytes / 4 : 0, msd_bits);
    if (try_buckets) {
        // the hint of the last large sort on this context (see Context::sort_overflow_host)
        Context* ctx = nullptr;
        (void)sort_overflow_word(&ctx);
        if (g_overflow_hint_on && ctx && ctx->sort_overflow_host && *ctx->sort_overflow_host != 0u) {
            if (++ctx->sort_hint_skips < 8) try_buckets = false;
            else ctx->sort_hint_skips = 0;               // every 8th time: look again
        }
    }
    if (try_buckets) {
        hipStream_t side = nullptr;
        const grace_status st = sort_pairs_bucketed<Key>(d_keys, d_values, n, value_bytes, begin_bit, end_bit,
                                                         msd_bits, d_perm_out, stream, &gate, &side);
        side_guard.forked = side != nullptr;
        if (st != GRACE_OK) return st;
        stream = side;
    }

    const uint32_t n_tiles = uint32_t((n + SORT_TILE - 1) / SORT_TILE);
    const size_t n_counts = size_t(RADIX) * n_tiles;
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
    GRACE_TRY(lsd_passes<Key>(k_in, k_out, i_in, i_out, n, begin_bit, end_bit, counts, scan_ws, stream, gate));
    // k_in / i_in now hold the sorted keys and their source indices.
    if (k_in != d_keys && !run_if)     // (a caller-gated sort's keys are scratch: only the permutation is kept)
        GRACE_TRY(copy_gated(k_in, d_keys, n * sizeof(Key), gate, stream));
    if (d_values) {
        void* tmp = Workspace::take<char>(n * size_t(value_bytes));
        GRACE_TRY(copy_gated(d_values, tmp, n * size_t(value_bytes), gate, stream));
        GRACE_TRY(gather_payload(tmp, i_in, d_values, n, value_bytes, stream, gate));
    }
    if (d_perm_out && i_in != d_perm_out)        // (cannot happen: see the ping-pong set-up)
        GRACE_TRY(copy_gated(i_in, d_perm_out, n * 4, gate, stream));
    return side_guard.join();
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

grace_status grace_sort_set_overflow_hint(int enabled)
{
    g_overflow_hint_on = enabled != 0;
    return GRACE_OK;
}

grace_status grace_sort_pairs_u64(uint64_t* d_keys, void* d_values, size_t n, int value_bytes,
                                  int begin_bit, int end_bit, uint32_t* d_perm,
                                  grace_stream stream)
{
    return sort_pairs<uint64_t>(d_keys, d_values, n, value_bytes, begin_bit, end_bit, d_perm,
                                as_stream(stream));
}

} // extern "C"
