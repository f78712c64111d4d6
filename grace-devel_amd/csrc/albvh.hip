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
//                     "greater" neighbours.  Each primitive grows its own cluster with
//                     bounded (<= max_per_leaf) neighbour scans -- no atomics, no
//                     inter-thread hand-off, fully deterministic.
//   2. scan         : leaf index = exclusive scan of the head flags (scan.hip).
//   3. write_leaves : compact leaf records + per-leaf deltas (fuses the reference's
//                     remove_if + copy_leaf_deltas_kernel, albvh.cuh:51-74,826-846).
//   4. nodes_climb  : ONE bottom-up pass over the leaves (one thread per leaf); the second
//                     thread to reach a node (agent-scope atomic counter) carries the union
//                     box upward.  Visibility across CUs/XCDs: release fence before the
//                     counter, acquire fence after it.
// Four launches + one scan instead of ~5 launches per level.  Traffic is HBM-bound and
// small: deltas 8 B, flags/counts 12 B, sphere 16 B per primitive, 64 B per node.
#include "common.hpp"

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
template <typename D>
__global__ __launch_bounds__(256) void leaf_heads_kernel(const D* __restrict__ ds, int n, int mpl,
                                                         uint32_t* __restrict__ flags,
                                                         uint32_t* __restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int l = i, r = i;
    for (;;) {
        if (l == 0 && r == n - 1) break;
        const D dl = ds[l];     // delta(l - 1)
        const D dr = ds[r + 1]; // delta(r)
        if (dl < dr) {
            // Parent is node l-1; it also owns every primitive to the left whose
            // separating node is lower in the order: delta(j) < delta(l-1), j < l-1.
            int nl = l - 1;
            while (r - nl + 1 <= mpl && nl >= 1 && ds[nl] < dl) --nl;
            if (opaque(r - nl + 1) > mpl) break;
            l = nl;
        } else {
            // Parent is node r; it owns primitives to the right while the separating
            // node j > r is lower in the order: !(delta(r) < delta(j)).
            int nr = r + 1;
            while (nr - l + 1 <= mpl && nr <= n - 2 && !(dr < ds[nr + 1])) ++nr;
            if (opaque(nr - l + 1) > mpl) break;
            r = nr;
        }
    }
    const bool head = (l == i);
    flags[i] = head ? 1u : 0u;
    counts[i] = head ? uint32_t(r - l + 1) : 0u;
}

template <typename D>
__global__ __launch_bounds__(256) void write_leaves_kernel(const uint32_t* __restrict__ flags,
                                                           const uint32_t* __restrict__ counts,
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
        const int c = int(counts[i]);
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

enum { PRIM_SPHERE = 0, PRIM_TRIANGLE = 1 };

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
            else triangle_box(reinterpret_cast<const float*>(prims) + 9 * size_t(leaf.x + i), b, tp);
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

template <typename D>
__global__ __launch_bounds__(256) void nodes_climb_kernel(const float* __restrict__ boxes,
                                                          int n_leaves,
                                                          const D* __restrict__ lds, int* nodes,
                                                          uint32_t* arrivals, int* root)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_leaves) return;
    const int n_nodes = n_leaves - 1;
    float bot[3], top[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        bot[c] = boxes[6 * size_t(k) + c];
        top[c] = boxes[6 * size_t(k) + 3 + c];
    }

    int gl = k, gr = k, cur = n_nodes + k;
    for (;;) {
        if (gl == 0 && gr == n_leaves - 1) { *root = cur; break; }
        const bool right_child = lds[gl] < lds[gr + 1]; // delta(gl-1) < delta(gr)
        const int p = right_child ? gl - 1 : gr;
        int* np = nodes + 16 * size_t(p);
        float* nf = reinterpret_cast<float*>(np);
        if (right_child) {
            np[1] = cur;
            np[3] = gr;
            *reinterpret_cast<float4*>(nf + 8) = make_float4(bot[0], top[0], bot[1], top[1]);
            *reinterpret_cast<float2*>(nf + 14) = make_float2(bot[2], top[2]);
        } else {
            np[0] = cur;
            np[2] = gl;
            *reinterpret_cast<float4*>(nf + 4) = make_float4(bot[0], top[0], bot[1], top[1]);
            *reinterpret_cast<float2*>(nf + 12) = make_float2(bot[2], top[2]);
        }
        // Publish this child's half of node p, then count the arrival.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t before =
            __hip_atomic_fetch_add(&arrivals[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == 0) break; // the sibling's thread will carry node p upward
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // Second arrival: read the sibling's half and continue as node p.
        if (right_child) {
            gl = np[2];
            const float4 q = *reinterpret_cast<const float4*>(nf + 4);
            const float2 z = *reinterpret_cast<const float2*>(nf + 12);
            bot[0] = fminf(bot[0], q.x); top[0] = fmaxf(top[0], q.y);
            bot[1] = fminf(bot[1], q.z); top[1] = fmaxf(top[1], q.w);
            bot[2] = fminf(bot[2], z.x); top[2] = fmaxf(top[2], z.y);
        } else {
            gr = np[3];
            const float4 q = *reinterpret_cast<const float4*>(nf + 8);
            const float2 z = *reinterpret_cast<const float2*>(nf + 14);
            bot[0] = fminf(bot[0], q.x); top[0] = fmaxf(top[0], q.y);
            bot[1] = fminf(bot[1], q.z); top[1] = fmaxf(top[1], q.w);
            bot[2] = fminf(bot[2], z.x); top[2] = fmaxf(top[2], z.y);
        }
        cur = p;
    }
}

template <typename D, int PRIM = PRIM_SPHERE>
grace_status albvh_build(const float* d_spheres, size_t n, const D* d_deltas, int mpl,
                         int* d_nodes, int* d_leaves, int* d_root, size_t* h_n_leaves,
                         hipStream_t stream)
{
    GRACE_REQUIRE(d_spheres && d_deltas && d_nodes && d_leaves && d_root && h_n_leaves,
                  "build_ALBVH: null pointer");
    GRACE_REQUIRE(mpl >= 1, "max_per_leaf must be at least 1.");
    // include/grace/cuda/kernels/albvh.cuh:795-799
    GRACE_REQUIRE(n > size_t(mpl),
                  "max_per_leaf must be less than the total number of primitives.");
    GRACE_REQUIRE(n < (size_t(1) << 31), "build_ALBVH: at most 2^31 - 1 primitives");
    const int ni = int(n);

    const size_t ws = 3 * Workspace::aligned(n * 4) + Workspace::aligned(scan_ws_count(n) * 4)
        + Workspace::aligned((n + 1) * sizeof(D)) + Workspace::aligned(n * 4)
        + Workspace::aligned(n * 24) + 1024;
    GRACE_TRY(Workspace::begin(ws));
    uint32_t* flags = Workspace::take<uint32_t>(n);
    uint32_t* counts = Workspace::take<uint32_t>(n);
    uint32_t* pos = Workspace::take<uint32_t>(n);
    uint32_t* scan_ws = Workspace::take<uint32_t>(scan_ws_count(n));
    D* leaf_ds = Workspace::take<D>(n + 1);
    uint32_t* arrivals = Workspace::take<uint32_t>(n);
    uint32_t* d_total = Workspace::take<uint32_t>(1);
    float* boxes = Workspace::take<float>(6 * n);

    const int grid = ceil_div(n, 256);
    leaf_heads_kernel<D><<<grid, 256, 0, stream>>>(d_deltas, ni, mpl, flags, counts);
    GRACE_CHECK_LAUNCH();
    GRACE_TRY(exclusive_scan_u32(flags, pos, n, scan_ws, d_total, stream));
    write_leaves_kernel<D><<<grid, 256, 0, stream>>>(flags, counts, pos, d_deltas, ni,
                                                     reinterpret_cast<int4*>(d_leaves), leaf_ds);
    GRACE_CHECK_LAUNCH();
    uint32_t n_leaves = 0;
    GRACE_TRY_HIP(hipMemcpyAsync(&n_leaves, d_total, 4, hipMemcpyDeviceToHost, stream));
    GRACE_TRY_HIP(hipStreamSynchronize(stream));
    *h_n_leaves = n_leaves;
    if (n_leaves < 2)
        return set_error(GRACE_INVALID_ARGUMENT, __FILE__, __LINE__,
                         "build_ALBVH: fewer than two leaves (NaN deltas?)");

    GRACE_TRY_HIP(hipMemsetAsync(arrivals, 0, size_t(n_leaves) * 4, stream));
    leaf_boxes_kernel<PRIM><<<ceil_div(size_t(n_leaves) * 8, 256), 256, 0, stream>>>(
        reinterpret_cast<const float4*>(d_spheres), reinterpret_cast<const int4*>(d_leaves),
        int(n_leaves), boxes);
    GRACE_CHECK_LAUNCH();
    nodes_climb_kernel<D><<<ceil_div(n_leaves, 256), 256, 0, stream>>>(
        boxes, int(n_leaves), leaf_ds, d_nodes, arrivals, d_root);
    GRACE_CHECK_LAUNCH();
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

} // extern "C"
