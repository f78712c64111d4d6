// Agglomerative LBVH ("ALBVH") construction for gfx950.
//
// Produces the same tree as the reference's build_ALBVH
// (include/grace/cuda/kernels/albvh.cuh:986-1021): leaves = maximal subtrees with at most
// max_per_leaf primitives of the hierarchy in which a cluster [l, r] joins its left
// neighbour iff delta(l-1) < delta(r) (albvh.cuh:125-143, 181-186, 246-291), internal nodes
// numbered by the leaf index of their split (albvh.cuh:470,491), children / ranges / child
// AABBs laid out as include/grace/cuda/nodes.h:22-37, root = the node spanning every leaf
// (albvh.cuh:572-573).
//
// It is NOT the reference's algorithm.  The reference climbs inside 512-wide windows with
// LDS flags, then iterates "slices" (3 kernels + 2 Thrust calls per ~log32 level,
// albvh.cuh:879-939).  Here:
//   1. leaf_heads   : the hierarchy is the Cartesian tree of the deltas under the total
//                     order (delta, -index); a node's extent is bounded by its nearest
//                     "greater" neighbours.  Each primitive decides whether it starts a leaf
//                     from two bounded (<= max_per_leaf) run lengths over its neighbours'
//                     deltas -- range-maxima probes for max_per_leaf <= 64, plain scans
//                     above -- no atomics, no inter-thread hand-off, fully deterministic.
//   2. scan         : leaf index = exclusive scan of the head flags (scan.hip) -- of the
//                     per-wave ballot words' population counts on the fast path.
//   3. write_leaves : compact leaf records + per-leaf deltas (fuses the reference's
//                     remove_if + copy_leaf_deltas_kernel, albvh.cuh:51-74,826-846).
//   4. leaf_boxes   : leaf AABBs, eight lanes per leaf (coalesced primitive reads).
//   5. pyramids     : 32-ary maxima of the leaf deltas and 32-ary unions of the leaf boxes
//                     (log32 n tiny launches).
//   6. nodes_direct : one thread per node finds its leaf range by nearest-greater search on
//                     the maxima pyramid, its two child boxes by range unions on the box
//                     pyramid, and plugs itself into its parent; one thread per leaf does the
//                     latter.  No atomics, no fences, every word written exactly once.
// A handful of launches + one scan instead of ~5 launches per level.  Traffic is HBM-bound and
// small: deltas 8 B, flags/counts 12 B, sphere 16 B per primitive, 64 B per node.
#include "common.hpp"

#include <type_traits>

using namespace grace_hip;

namespace {

// hipcc (ROCm 7.2) reuses the divergent while-loop's size test after the loop through VCC,
// whose bits for lanes that left the loop early were zeroed by later iterations (seen in
// the ISA and as wrong leaves on MI355X).  Re-deriving the size from an opaque VGPR copy
// forces a fresh compare for every lane.
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// ds is the delta array with the reference's +1 shift: ds[k + 1] = delta(k).
//
// Leaves are the maximal subtrees with at most max_per_leaf primitives, and the node that
// separates primitives i-1 and i (node i-1) is their lowest common ancestor: they share a
// leaf iff that node's subtree holds <= max_per_leaf primitives.  Node j's subtree is the
// maximal run of nodes around j that merge before j does -- to the left while
// delta(k) < delta(j), to the right while !(delta(j) < delta(k)) (ties merge right first:
// the "left parent iff delta(l-1) < delta(r)" rule) -- and holds run_left + run_right + 2
// primitives.  So one thread per primitive decides its head flag with two bounded scans
// (<= max_per_leaf steps in total, stopping as soon as the count exceeds the bound) over a
// window of the deltas that the workgroup first copies into LDS; no climbing, no outer loop.
// (The previous formulation grew every primitive's cluster bottom-up: the same answer, but
// a wave serialised its lanes' differently-phased scans: 0.76 ms at 8.4 M primitives; this:
// see profiles/.)
constexpr int LEAF_TILE_MAX_MPL = 256;

template <typename D, bool TILED>
__global__ __launch_bounds__(256) void leaf_heads_kernel(const D* __restrict__ ds, int n, int mpl,
                                                         uint32_t* __restrict__ flags)
{
    __shared__ D tile[TILED ? 256 + 2 * LEAF_TILE_MAX_MPL + 4 : 1];
    const int b0 = blockIdx.x * blockDim.x;
    const int lo = b0 - mpl - 1;                   // first delta index held by the tile
    if (TILED) {
        const int hi = min(n, b0 + 255 + mpl + 1); // last index (ds has n + 1 entries)
        for (int k = max(lo, 0) + int(threadIdx.x); k <= hi; k += 256) tile[k - lo] = ds[k];
        __syncthreads();
    }
    auto at = [&](const int k) -> D { return TILED ? tile[k - lo] : ds[k]; };
    const int i = b0 + threadIdx.x;
    if (i >= n) return;
    bool head = true;                              // primitive 0 starts the first leaf
    if (i > 0) {
        const int j = i - 1;                       // the node between primitives i-1 and i
        const D dj = at(j + 1);
        int size = 2;
        for (int k = j - 1; size <= mpl && k >= 0 && at(k + 1) < dj; --k) ++size;
        for (int k = j + 1; size <= mpl && k <= n - 2 && !(dj < at(k + 1)); ++k) ++size;
        head = opaque(size) > mpl;   // fresh compare: see opaque()
    }
    flags[i] = head ? 1u : 0u;
}

// Leaf records and per-leaf deltas (the reference's remove_if + copy_leaf_deltas_kernel,
// albvh.cuh:51-74,826-846, fused).  A head finds its leaf's size by looking for the next head
// (<= max_per_leaf steps).
template <typename D>
__global__ __launch_bounds__(256) void write_leaves_kernel(const uint32_t* __restrict__ flags,
                                                           const uint32_t* __restrict__ pos,
                                                           const D* __restrict__ ds, int n,
                                                           int4* __restrict__ leaves,
                                                           D* __restrict__ leaf_ds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0) leaf_ds[0] = ds[0];
    if (flags[i]) {
        const int k = int(pos[i]);
        int c = 1;
        while (i + c < n && !flags[i + c]) ++c;
        leaves[k] = make_int4(i, c, 0, 0);
        leaf_ds[k + 1] = ds[i + c]; // delta(last primitive of the leaf)
    }
}

// ---- leaf stage, fast form (max_per_leaf <= LEAF_FAST_MAX_MPL) -------------------------------
// The two bounded scans above are threshold searches ("how far do the deltas stay below /
// not above d(j)?"), so range maxima answer them: the workgroup builds a sparse table of
// maxima over windows of 1, 2, 4 ... 2^(L-1) deltas in LDS (L = ceil(log2 max_per_leaf)) and
// every thread finds each run length with L probes, longest window first, instead of up to
// max_per_leaf dependent, divergent steps.  The head flags leave as one 64-bit ballot word per
// wave plus its population count: the scan that numbers the leaves then runs over n / 64 words
// instead of n flags, and write_leaves finds a leaf's end by counting trailing zeros.
// (10^7 primitives, max_per_leaf 32: leaf_heads 199 -> 62 us, scan + write_leaves 215 -> 40 us;
// same leaves -- the tests compare them with the oracle's literal restatement.)
constexpr int LEAF_FAST_MAX_MPL = 64;
constexpr int LEAF_FAST_LEVELS = 6;                       // windows of 1 .. 32 deltas
constexpr int LEAF_FAST_MARGIN = 1 << LEAF_FAST_LEVELS;   // >= the longest reach of a search
constexpr int LEAF_FAST_TILE = 256 + 2 * LEAF_FAST_MARGIN;

template <typename D>
__global__ __launch_bounds__(256) void leaf_heads_bits_kernel(const D* __restrict__ ds, int n, int mpl,
                                                              int levels,
                                                              unsigned long long* __restrict__ bits,
                                                              uint32_t* __restrict__ counts)
{
    // m[l][t] = max of node deltas lo + t .. lo + t + 2^l - 1 (node k's delta is ds[k + 1])
    __shared__ D m[LEAF_FAST_LEVELS][LEAF_FAST_TILE];
    const int b0 = blockIdx.x * blockDim.x;
    const int lo = b0 - 1 - LEAF_FAST_MARGIN;             // node index held by m[.][0]
    for (int t = threadIdx.x; t < LEAF_FAST_TILE; t += 256) {
        const int k = min(max(lo + t, 0), n - 2);         // clamped: never read as such (bounds below)
        m[0][t] = ds[k + 1];
    }
    __syncthreads();
    for (int l = 1; l < levels; ++l) {
        const int half = 1 << (l - 1);
        for (int t = threadIdx.x; t + half < LEAF_FAST_TILE; t += 256) {
            const D x = m[l - 1][t], y = m[l - 1][t + half];
            m[l][t] = (x < y) ? y : x;
        }
        __syncthreads();
    }
    const int i = b0 + threadIdx.x;
    bool head = i < n;                                    // primitive 0 starts the first leaf
    if (i > 0 && i < n) {
        const int j = i - 1;                              // the node between primitives i-1 and i
        const D dj = m[0][j - lo];
        // run_left: nodes j-1, j-2 ... while delta < d(j), never past node 0
        int pos = j;                                      // the run is [pos, j - 1]
        for (int l = levels - 1; l >= 0; --l) {
            const int s = 1 << l;
            if (pos - s >= 0 && m[l][pos - s - lo] < dj) pos -= s;
        }
        // run_right: nodes j+1, j+2 ... while !(d(j) < delta), never past node n - 2
        int end = j + 1;                                  // the run is [j + 1, end - 1]
        for (int l = levels - 1; l >= 0; --l) {
            const int s = 1 << l;
            if (end + s - 1 <= n - 2 && !(dj < m[l][end - lo])) end += s;
        }
        // (each run is found exactly up to 2^levels - 1 >= max_per_leaf - 1 members, which is
        // all the bounded scans of the reference rule can tell apart)
        head = 2 + (j - pos) + (end - (j + 1)) > mpl;
    }
    const unsigned long long word = __builtin_amdgcn_ballot_w64(head);
    if ((threadIdx.x & 63) == 0) {
        const int w = i >> 6;
        bits[w] = word;
        counts[w] = uint32_t(__builtin_popcountll(word));
    }
}

// Leaf records and per-leaf deltas from the ballot words: leaf index = leaves before this
// wave's word + heads below this lane; its end = the next set bit (in this word or a later one).
template <typename D>
__global__ __launch_bounds__(256) void write_leaves_bits_kernel(const unsigned long long* __restrict__ bits,
                                                                const uint32_t* __restrict__ wpos,
                                                                const D* __restrict__ ds, int n,
                                                                int4* __restrict__ leaves,
                                                                D* __restrict__ leaf_ds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0) leaf_ds[0] = ds[0];
    const int w = i >> 6, lane = i & 63;
    const unsigned long long word = bits[w];
    if ((word >> lane) & 1ull) {
        const int k = int(wpos[w]) + __builtin_popcountll(word & ((1ull << lane) - 1ull));
        const unsigned long long rest = lane == 63 ? 0ull : word >> (lane + 1);
        int c;
        if (rest) {
            c = __builtin_ctzll(rest) + 1;
        } else {
            c = 64 - lane;
            const int n_words = (n + 63) >> 6;
            for (int v = w + 1; v < n_words; ++v) {
                const unsigned long long nxt = bits[v];
                if (nxt) { c += __builtin_ctzll(nxt); break; }
                c += 64;
            }
            c = min(c, n - i);
        }
        leaves[k] = make_int4(i, c, 0, 0);
        leaf_ds[k + 1] = ds[i + c]; // delta(last primitive of the leaf)
    }
}

__device__ __forceinline__ void sphere_box(const float4 s, float* bot, float* top)
{
    // AABBSphere, include/grace/generic/functors/aabb.h:9-26
    bot[0] = s.x - s.w; top[0] = s.x + s.w;
    bot[1] = s.y - s.w; top[1] = s.y + s.w;
    bot[2] = s.z - s.w; top[2] = s.z + s.w;
}

// TriangleAABB (tests/profile_trace_triangle/triangle.cu:3-35): min/max over v, v+e1, v+e2,
// zero-extent axes inflated by AABB_EPSILON * |coordinate|.
__device__ __forceinline__ void triangle_box(const float* __restrict__ t, float* bot, float* top)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v0 = t[k], v1 = t[k] + t[3 + k], v2 = t[k] + t[6 + k];
        bot[k] = fminf(v0, fminf(v1, v2));
        top[k] = fmaxf(v0, fmaxf(v1, v2));
        if (bot[k] == top[k]) {
            const float scale = fabsf(bot[k]);
            bot[k] -= 0.000001f * scale;
            top[k] += 0.000001f * scale;
        }
    }
}

// PRIM_BOX: the caller's own AABB functor already evaluated per primitive (the generic
// build_ALBVH(..., AABBFunc) of grace/cuda/kernels/albvh.cuh: the functor runs in a header kernel
// of the caller's translation unit, the tree builder here): 6 floats {bot xyz, top xyz}.
enum { PRIM_SPHERE = GRACE_PRIM_SPHERE_F4, PRIM_TRIANGLE = GRACE_PRIM_TRIANGLE,
       PRIM_SPHERE_D4 = GRACE_PRIM_SPHERE_D4, PRIM_BOX = GRACE_PRIM_BOX };

// Leaf AABBs (albvh.cuh:402-424): eight lanes per leaf stride over its primitives, so a
// wave reads eight runs of consecutive primitives (128 B each for spheres), then an
// 8-lane shuffle reduction; min/max only, so the boxes are exactly the reference's.
template <int PRIM>
__global__ __launch_bounds__(256) void leaf_boxes_kernel(const float4* __restrict__ prims,
                                                         const int4* __restrict__ leaves,
                                                         int n_leaves, float* __restrict__ boxes)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = t >> 3, sub = t & 7;
    float bot[3] = { INFINITY, INFINITY, INFINITY };
    float top[3] = { -INFINITY, -INFINITY, -INFINITY };
    if (k < n_leaves) {
        const int4 leaf = leaves[k];
        for (int i = sub; i < leaf.y; i += 8) {
            float b[3], tp[3];
            if (PRIM == PRIM_SPHERE) sphere_box(prims[leaf.x + i], b, tp);
            else if (PRIM == PRIM_BOX) {
                const float* q = reinterpret_cast<const float*>(prims) + 6 * size_t(leaf.x + i);
#pragma unroll
                for (int c = 0; c < 3; ++c) { b[c] = q[c]; tp[c] = q[3 + c]; }
            } else if (PRIM == PRIM_SPHERE_D4) {
                // AABBSphere with Real4 = double4 (generic/functors/aabb.h:9-26): centre -+ radius
                // in double, narrowed to the float corner
                const double* s = reinterpret_cast<const double*>(prims) + 4 * size_t(leaf.x + i);
#pragma unroll
                for (int c = 0; c < 3; ++c) { b[c] = float(s[c] - s[3]); tp[c] = float(s[c] + s[3]); }
            } else triangle_box(reinterpret_cast<const float*>(prims) + 9 * size_t(leaf.x + i), b, tp);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                bot[c] = fminf(bot[c], b[c]);
                top[c] = fmaxf(top[c], tp[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
            bot[c] = fminf(bot[c], __shfl_xor(bot[c], off));
            top[c] = fmaxf(top[c], __shfl_xor(top[c], off));
        }
    }
    if (k < n_leaves && sub < 6)
        boxes[6 * size_t(k) + sub] = sub < 3 ? bot[sub] : top[sub - 3];
}

// ---- node stage without any inter-thread hand-off ------------------------------------------
// Node j (between leaves j and j+1) owns the leaves [first, last] bounded by its nearest
// "greater" neighbours in the order (delta, -index):
//   first - 1 = the nearest i < j with !(d(i) < d(j)),   last = the nearest i > j with d(j) < d(i)
// (-1 and n_leaves - 1, the sentinels, if there is none).  Both predicates are threshold tests
// on d, so block maxima of d (a 32-ary pyramid) locate them in O(32 log32 n) steps; the child
// boxes are range unions of leaf boxes, answered by a second 32-ary pyramid of box unions
// (min/max only: the same values as the reference's child-to-parent propagation,
// albvh.cuh:387-400).  Each node then writes its own record and plugs its index into its
// parent's child slot; each leaf does the same.  Every word of the node array has exactly one
// writer: no atomics, no fences, bit-reproducible.
constexpr int PYR = 32;
constexpr int PYR_SHIFT = 5;
constexpr int PYR_MAX_LEVELS = 8;

template <typename D>
struct MaxPyramid {
    const D* level[PYR_MAX_LEVELS]; // level[0][j] = d(j), j in [0, n_nodes); level[k+1][b] = max of 32
    int size[PYR_MAX_LEVELS];
    int levels;
};

struct BoxPyramid {
    const float* level[PYR_MAX_LEVELS]; // 6 floats per entry {bot xyz, top xyz}
    int size[PYR_MAX_LEVELS];
    int levels;
};

template <typename D>
__global__ __launch_bounds__(256) void pyr_max_kernel(const D* __restrict__ in, int n_in,
                                                      D* __restrict__ out, int n_out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_out) return;
    const int lo = b << PYR_SHIFT, hi = min(lo + PYR, n_in);
    D m = in[lo];
    for (int i = lo + 1; i < hi; ++i) { const D v = in[i]; m = (m < v) ? v : m; }
    out[b] = m;
}

__global__ __launch_bounds__(256) void pyr_box_kernel(const float* __restrict__ in, int n_in,
                                                      float* __restrict__ out, int n_out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = t / 6, c = t % 6;
    if (b >= n_out) return;
    const int lo = b << PYR_SHIFT, hi = min(lo + PYR, n_in);
    float m = in[6 * size_t(lo) + c];
    for (int i = lo + 1; i < hi; ++i) {
        const float v = in[6 * size_t(i) + c];
        m = c < 3 ? fminf(m, v) : fmaxf(m, v);
    }
    out[6 * size_t(b) + c] = m;
}

// Largest i < j with !(d(i) < x), or -1.
template <typename D>
__device__ __forceinline__ int nearest_ge_left(const MaxPyramid<D>& p, int j, const D x)
{
    int pos = j - 1, lvl = 0;
    while (pos >= 0) {
        const int block0 = pos & ~(PYR - 1);
        int e = pos;
        for (; e >= block0; --e)
            if (!(p.level[lvl][e] < x)) break;
        if (e >= block0) { // found at this level: walk down to the element
            while (lvl > 0) {
                --lvl;
                const int c0 = e << PYR_SHIFT;
                int c = min(c0 + PYR - 1, p.size[lvl] - 1);
                while (c > c0 && p.level[lvl][c] < x) --c; // bounded even for NaN input
                e = c;
            }
            return e;
        }
        if (lvl + 1 >= p.levels) return -1;
        pos = (block0 >> PYR_SHIFT) - 1;
        ++lvl;
    }
    return -1;
}

// Smallest i > j with x < d(i), or n (= n_nodes, the right sentinel's index).
template <typename D>
__device__ __forceinline__ int nearest_gt_right(const MaxPyramid<D>& p, int j, const D x, int n)
{
    int pos = j + 1, lvl = 0;
    while (pos < p.size[lvl]) {
        const int block1 = min((pos | (PYR - 1)) + 1, p.size[lvl]); // end of pos's block
        int e = pos;
        for (; e < block1; ++e)
            if (x < p.level[lvl][e]) break;
        if (e < block1) {
            while (lvl > 0) {
                --lvl;
                int c = e << PYR_SHIFT;
                const int c1 = min(c + PYR, p.size[lvl]) - 1;
                while (c < c1 && !(x < p.level[lvl][c])) ++c;
                e = c;
            }
            return e;
        }
        if (lvl + 1 >= p.levels) return n;
        pos = ((pos | (PYR - 1)) + 1) >> PYR_SHIFT;
        ++lvl;
    }
    return n;
}

// Union of the leaf boxes lo .. hi (inclusive).
__device__ __forceinline__ void box_union(const BoxPyramid& p, int lo, int hi, float* bot, float* top)
{
#pragma unroll
    for (int c = 0; c < 3; ++c) { bot[c] = INFINITY; top[c] = -INFINITY; }
    int a = lo, b = hi + 1, lvl = 0;
    auto take = [&](int e) {
        const float* q = p.level[lvl] + 6 * size_t(e);
#pragma unroll
        for (int c = 0; c < 3; ++c) { bot[c] = fminf(bot[c], q[c]); top[c] = fmaxf(top[c], q[3 + c]); }
    };
    while (a < b) {
        while (a < b && (a & (PYR - 1))) take(a++);
        while (a < b && (b & (PYR - 1))) take(--b);
        if (a >= b) break;
        if (lvl + 1 >= p.levels) { // top level: plain scan
            while (a < b) take(a++);
            break;
        }
        a >>= PYR_SHIFT; b >>= PYR_SHIFT; ++lvl;
    }
}

// lds: leaf deltas with the reference's +1 shift (lds[k + 1] = d(k); lds[0], lds[n_leaves]
// are the sentinels).  One thread per node and one per leaf.
template <typename D>
__global__ __launch_bounds__(256) void nodes_direct_kernel(const MaxPyramid<D> mp, const BoxPyramid bp,
                                                           const D* __restrict__ lds, int n_leaves,
                                                           int* __restrict__ nodes, int* __restrict__ root)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_nodes = n_leaves - 1;
    if (t < n_nodes) {
        const int j = t;
        const D x = lds[j + 1];
        const int first = nearest_ge_left(mp, j, x) + 1;
        const int last = nearest_gt_right(mp, j, x, n_nodes);
        int* np = nodes + 16 * size_t(j);
        float* nf = reinterpret_cast<float*>(np);
        np[2] = first;
        np[3] = last;
        float bot[3], top[3];
        box_union(bp, first, j, bot, top);          // left child covers leaves first .. j
        *reinterpret_cast<float4*>(nf + 4) = make_float4(bot[0], top[0], bot[1], top[1]);
        *reinterpret_cast<float2*>(nf + 12) = make_float2(bot[2], top[2]);
        box_union(bp, j + 1, last, bot, top);       // right child covers leaves j+1 .. last
        *reinterpret_cast<float4*>(nf + 8) = make_float4(bot[0], top[0], bot[1], top[1]);
        *reinterpret_cast<float2*>(nf + 14) = make_float2(bot[2], top[2]);
        // d(first-1) < d(last): right child of node first-1, else left child of node `last`
        // (albvh.cuh:465-505); the sentinels at either end are never a parent.
        const bool right_child = first > 0 && (last == n_leaves - 1 || lds[first] < lds[last + 1]);
        if (first == 0 && last == n_leaves - 1) {
            *root = j;                               // albvh.cuh:572-573
        } else if (right_child) {
            nodes[16 * size_t(first - 1) + 1] = j;
        } else {
            nodes[16 * size_t(last)] = j;
        }
    } else if (t < n_nodes + n_leaves) {
        const int k = t - n_nodes;                   // leaf k: cluster [k, k]
        const bool right_child = k > 0 && (k == n_leaves - 1 || lds[k] < lds[k + 1]);
        if (right_child) nodes[16 * size_t(k - 1) + 1] = n_nodes + k;
        else nodes[16 * size_t(k)] = n_nodes + k;
    }
}

// DeltaComp = thrust::greater (albvh.cuh:1029-1045 lets the caller choose the comparator; it is
// only ever applied as delta_comp(delta_L, delta_R), albvh.cuh:129,194,465,607): the build below is
// written for `<`, and a > b  <=>  flip(a) < flip(b) with flip = negation for floating-point deltas
// (exact, order-reversing, NaN stays unordered) and bitwise NOT for unsigned ones.
template <typename D>
__device__ __forceinline__ D flip_order(const D d)
{
    if constexpr (std::is_floating_point<D>::value) return -d;
    else return ~d;
}

template <typename D>
__global__ __launch_bounds__(256) void flip_deltas_kernel(const D* __restrict__ in, size_t n,
                                                          D* __restrict__ out)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n;
         i += size_t(gridDim.x) * blockDim.x)
        out[i] = flip_order(in[i]);
}

// Measurement hook (profile_tree's per-phase lines): HIP events around the leaf stage and the
// node stage of the last build on its stream.
// (they live in the calling thread's Context: phase_timing, phase_valid, phase_events)

template <typename D, int PRIM = PRIM_SPHERE>
grace_status albvh_build(const float* d_spheres, size_t n, const D* d_deltas, int mpl,
                         int* d_nodes, int* d_leaves, int* d_root, size_t* h_n_leaves,
                         hipStream_t stream, const bool greater = false)
{
    GRACE_REQUIRE(d_spheres && d_deltas && d_nodes && d_leaves && d_root && h_n_leaves,
                  "build_ALBVH: null pointer");
    GRACE_REQUIRE(mpl >= 1, "max_per_leaf must be at least 1.");
    // include/grace/cuda/kernels/albvh.cuh:795-799
    GRACE_REQUIRE(n > size_t(mpl),
                  "max_per_leaf must be less than the total number of primitives.");
    GRACE_REQUIRE(n < (size_t(1) << 31), "build_ALBVH: at most 2^31 - 1 primitives");
    const int ni = int(n);
    GRACE_TRY(scene_invalidate_if_written(d_nodes));   // a prepared trace scene over this tree is stale
    GRACE_TRY(scene_invalidate_if_written(d_leaves));

    const size_t ws = 3 * Workspace::aligned(n * 4) + Workspace::aligned(scan_ws_count(n) * 4)
        + Workspace::aligned((n + 1) * sizeof(D)) + Workspace::aligned(n * 4)
        + Workspace::aligned(n * 24) + 1024
        + Workspace::aligned((n / 64 + 8) * 8) + 2 * Workspace::aligned((n / 64 + 8) * 4)
        + (greater ? Workspace::aligned((n + 1) * sizeof(D)) : 0);
    FrameGuard frame;
    GRACE_TRY(frame.begin(ws, stream));
    if (greater) {
        D* flipped = Workspace::take<D>(n + 1);
        flip_deltas_kernel<D><<<stream_grid(n + 1, 256), 256, 0, stream>>>(d_deltas, n + 1, flipped);
        GRACE_CHECK_LAUNCH();
        d_deltas = flipped;
    }
    uint32_t* flags = Workspace::take<uint32_t>(n);
    uint32_t* counts = Workspace::take<uint32_t>(n); // scratch of the pyramids below
    (void)counts;
    uint32_t* pos = Workspace::take<uint32_t>(n);
    uint32_t* scan_ws = Workspace::take<uint32_t>(scan_ws_count(n));
    D* leaf_ds = Workspace::take<D>(n + 1);
    uint32_t* arrivals = Workspace::take<uint32_t>(n);
    uint32_t* d_total = Workspace::take<uint32_t>(1);
    float* boxes = Workspace::take<float>(6 * n);

    Context& ctx = *Workspace::frame_context();
    if (ctx.phase_timing) {
        for (auto& e : ctx.phase_events)
            if (!e) GRACE_TRY_HIP(hipEventCreate(&e));
        ctx.phase_valid = false;
        GRACE_TRY_HIP(hipEventRecord(ctx.phase_events[0], stream));
    }
    const int grid = ceil_div(n, 256);
    if (mpl <= LEAF_FAST_MAX_MPL) {
        int levels = 0;                                   // 2^levels - 1 >= mpl - 1
        while ((1 << levels) < mpl) ++levels;
        if (levels < 1) levels = 1;
        const size_t n_words = size_t(grid) * 4;          // one ballot word per wave launched
        unsigned long long* bits = Workspace::take<unsigned long long>(n_words);
        uint32_t* wcount = Workspace::take<uint32_t>(n_words);
        uint32_t* wpos = Workspace::take<uint32_t>(n_words);
        leaf_heads_bits_kernel<D><<<grid, 256, 0, stream>>>(d_deltas, ni, mpl, levels, bits, wcount);
        GRACE_CHECK_LAUNCH();
        GRACE_TRY(exclusive_scan_u32(wcount, wpos, n_words, scan_ws, d_total, stream));
        write_leaves_bits_kernel<D><<<grid, 256, 0, stream>>>(bits, wpos, d_deltas, ni,
                                                              reinterpret_cast<int4*>(d_leaves), leaf_ds);
        GRACE_CHECK_LAUNCH();
    } else {
        if (mpl <= LEAF_TILE_MAX_MPL)
            leaf_heads_kernel<D, true><<<grid, 256, 0, stream>>>(d_deltas, ni, mpl, flags);
        else
            leaf_heads_kernel<D, false><<<grid, 256, 0, stream>>>(d_deltas, ni, mpl, flags);
        GRACE_CHECK_LAUNCH();
        GRACE_TRY(exclusive_scan_u32(flags, pos, n, scan_ws, d_total, stream));
        write_leaves_kernel<D><<<grid, 256, 0, stream>>>(flags, pos, d_deltas, ni,
                                                         reinterpret_cast<int4*>(d_leaves), leaf_ds);
        GRACE_CHECK_LAUNCH();
    }
    if (ctx.phase_timing) GRACE_TRY_HIP(hipEventRecord(ctx.phase_events[1], stream));
    uint32_t n_leaves = 0;
    GRACE_TRY_HIP(hipMemcpyAsync(&n_leaves, d_total, 4, hipMemcpyDeviceToHost, stream));
    GRACE_TRY_HIP(hipStreamSynchronize(stream));
    *h_n_leaves = n_leaves;
    if (n_leaves < 2)
        return set_error(GRACE_INVALID_ARGUMENT, __FILE__, __LINE__,
                         "build_ALBVH: fewer than two leaves (NaN deltas?)");

    leaf_boxes_kernel<PRIM><<<ceil_div(size_t(n_leaves) * 8, 256), 256, 0, stream>>>(
        reinterpret_cast<const float4*>(d_spheres), reinterpret_cast<const int4*>(d_leaves),
        int(n_leaves), boxes);
    GRACE_CHECK_LAUNCH();
    // Pyramids: maxima of the node deltas d(j) = leaf_ds[j + 1], and unions of the leaf boxes.
    const int n_nodes = int(n_leaves) - 1;
    MaxPyramid<D> mp;
    BoxPyramid bp;
    mp.levels = bp.levels = 1;
    mp.level[0] = leaf_ds + 1; mp.size[0] = n_nodes;
    bp.level[0] = boxes; bp.size[0] = int(n_leaves);
    D* pyr_d = reinterpret_cast<D*>(flags);         // flags / counts / pos are dead by now
    float* pyr_b = reinterpret_cast<float*>(arrivals);
    while (mp.size[mp.levels - 1] > PYR && mp.levels < PYR_MAX_LEVELS) {
        const int n_in = mp.size[mp.levels - 1], n_out = (n_in + PYR - 1) / PYR;
        pyr_max_kernel<D><<<ceil_div(n_out, 256), 256, 0, stream>>>(mp.level[mp.levels - 1], n_in,
                                                                    pyr_d, n_out);
        GRACE_CHECK_LAUNCH();
        mp.level[mp.levels] = pyr_d; mp.size[mp.levels] = n_out; ++mp.levels;
        pyr_d += (n_out + 63) & ~63;
    }
    while (bp.size[bp.levels - 1] > PYR && bp.levels < PYR_MAX_LEVELS) {
        const int n_in = bp.size[bp.levels - 1], n_out = (n_in + PYR - 1) / PYR;
        pyr_box_kernel<<<ceil_div(size_t(n_out) * 6, 256), 256, 0, stream>>>(bp.level[bp.levels - 1],
                                                                             n_in, pyr_b, n_out);
        GRACE_CHECK_LAUNCH();
        bp.level[bp.levels] = pyr_b; bp.size[bp.levels] = n_out; ++bp.levels;
        pyr_b += (size_t(n_out) * 6 + 63) & ~size_t(63);
    }
    nodes_direct_kernel<D><<<ceil_div(size_t(n_nodes) + n_leaves, 256), 256, 0, stream>>>(
        mp, bp, leaf_ds, int(n_leaves), d_nodes, d_root);
    GRACE_CHECK_LAUNCH();
    if (ctx.phase_timing) {
        GRACE_TRY_HIP(hipEventRecord(ctx.phase_events[2], stream));
        ctx.phase_valid = true;
    }
    return GRACE_OK;
}

} // namespace

extern "C" {

grace_status grace_albvh_build_f4(const float* d_spheres, size_t n, const float* d_deltas,
                                  int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                  size_t* h_n_leaves, grace_stream stream)
{
    return albvh_build<float>(d_spheres, n, d_deltas, max_per_leaf, d_nodes, d_leaves, d_root,
                              h_n_leaves, as_stream(stream));
}

grace_status grace_albvh_build_f4_u32(const float* d_spheres, size_t n, const uint32_t* d_deltas,
                                      int max_per_leaf, int* d_nodes, int* d_leaves,
                                      int* d_root, size_t* h_n_leaves, grace_stream stream)
{
    return albvh_build<uint32_t>(d_spheres, n, d_deltas, max_per_leaf, d_nodes, d_leaves,
                                 d_root, h_n_leaves, as_stream(stream));
}

grace_status grace_albvh_build_tri_u32(const float* d_tris, size_t n, const uint32_t* d_deltas,
                                       int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                       size_t* h_n_leaves, grace_stream stream)
{
    return albvh_build<uint32_t, PRIM_TRIANGLE>(d_tris, n, d_deltas, max_per_leaf, d_nodes,
                                                d_leaves, d_root, h_n_leaves, as_stream(stream));
}

grace_status grace_albvh_build_d4(const double* d_spheres, size_t n, const float* d_deltas,
                                  int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                  size_t* h_n_leaves, grace_stream stream)
{
    return albvh_build<float, PRIM_SPHERE_D4>(reinterpret_cast<const float*>(d_spheres), n, d_deltas,
                                              max_per_leaf, d_nodes, d_leaves, d_root, h_n_leaves,
                                              as_stream(stream));
}

// The remaining (Real4, DeltaType) instantiations of ALBVH_sph (build_sph.cuh:118-124).
#define GRACE_ALBVH(NAME, PRIM_T, DELTA_T, PRIM_KIND)                                            \
    grace_status NAME(const PRIM_T* d_spheres, size_t n, const DELTA_T* d_deltas,                \
                      int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,                \
                      size_t* h_n_leaves, grace_stream stream)                                   \
    {                                                                                            \
        return albvh_build<DELTA_T, PRIM_KIND>(reinterpret_cast<const float*>(d_spheres), n,     \
                                               d_deltas, max_per_leaf, d_nodes, d_leaves,        \
                                               d_root, h_n_leaves, as_stream(stream));           \
    }
GRACE_ALBVH(grace_albvh_build_f4_u64, float, uint64_t, PRIM_SPHERE)
GRACE_ALBVH(grace_albvh_build_f4_f64, float, double, PRIM_SPHERE)
GRACE_ALBVH(grace_albvh_build_d4_f64, double, double, PRIM_SPHERE_D4)
GRACE_ALBVH(grace_albvh_build_d4_u32, double, uint32_t, PRIM_SPHERE_D4)
GRACE_ALBVH(grace_albvh_build_d4_u64, double, uint64_t, PRIM_SPHERE_D4)
#undef GRACE_ALBVH

// One entry for every (primitive kind, delta type, comparator) of build_ALBVH
// (albvh.cuh:986-1072), incl. caller-evaluated boxes and DeltaComp = greater.
grace_status grace_albvh_build_ex(int prim_kind, const void* d_prims, size_t n, int delta_type,
                                  const void* d_deltas, int delta_comp, int max_per_leaf,
                                  int* d_nodes, int* d_leaves, int* d_root, size_t* h_n_leaves,
                                  grace_stream stream)
{
    GRACE_REQUIRE(delta_comp == GRACE_COMP_LESS || delta_comp == GRACE_COMP_GREATER,
                  "build_ALBVH: delta comparator must be GRACE_COMP_LESS or GRACE_COMP_GREATER");
    const bool gt = delta_comp == GRACE_COMP_GREATER;
    const float* p = static_cast<const float*>(d_prims);
    hipStream_t s = as_stream(stream);
#define GRACE_EX(K, DT, D)                                                                       \
    if (prim_kind == K && delta_type == DT)                                                      \
        return albvh_build<D, K>(p, n, static_cast<const D*>(d_deltas), max_per_leaf, d_nodes,   \
                                 d_leaves, d_root, h_n_leaves, s, gt);
#define GRACE_EX_KIND(K)                                                                         \
    GRACE_EX(K, GRACE_DELTA_F32, float) GRACE_EX(K, GRACE_DELTA_F64, double)                     \
    GRACE_EX(K, GRACE_DELTA_U32, uint32_t) GRACE_EX(K, GRACE_DELTA_U64, uint64_t)
    GRACE_EX_KIND(PRIM_SPHERE) GRACE_EX_KIND(PRIM_TRIANGLE) GRACE_EX_KIND(PRIM_SPHERE_D4)
    GRACE_EX_KIND(PRIM_BOX)
#undef GRACE_EX_KIND
#undef GRACE_EX
    return set_error(GRACE_INVALID_ARGUMENT, __FILE__, __LINE__,
                     "build_ALBVH: unknown primitive kind or delta type");
}

grace_status grace_albvh_enable_timing(int enabled)
{
    Context* ctx = nullptr;
    GRACE_TRY(current_context(&ctx));
    ctx->phase_timing = enabled != 0;
    if (!ctx->phase_timing) ctx->phase_valid = false;
    return GRACE_OK;
}

grace_status grace_albvh_last_phase_ms(float* h_leaves_ms, float* h_nodes_ms)
{
    GRACE_REQUIRE(h_leaves_ms && h_nodes_ms, "albvh_last_phase_ms: null pointer");
    Context* ctx = nullptr;
    GRACE_TRY(current_context(&ctx));
    GRACE_REQUIRE(ctx->phase_timing && ctx->phase_valid, "no timed ALBVH build recorded");
    GRACE_TRY_HIP(hipEventSynchronize(ctx->phase_events[2]));
    GRACE_TRY_HIP(hipEventElapsedTime(h_leaves_ms, ctx->phase_events[0], ctx->phase_events[1]));
    GRACE_TRY_HIP(hipEventElapsedTime(h_nodes_ms, ctx->phase_events[1], ctx->phase_events[2]));
    return GRACE_OK;
}

} // extern "C"
