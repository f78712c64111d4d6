// Scene-constant records of the traversal, for gfx950: per-sphere constants A = {x, y, z, h^2},
// B = {1/h or 50/h, 1/h^2} (hoisted out of the ray x sphere loop), one box per cluster of 64
// consecutive primitives, every node's primitive span; bounding spheres + fp64 copies of
// triangles; float records that contain double4 spheres.  Streaming, HBM-bound passes over the
// primitives (40.5 B per sphere written, 16 B read).  See trace.hip for how the walk uses them.
#include "trace_state.hpp"

using namespace grace_hip;

namespace {

// One wave, one cluster: the box of the 64 records a(i), i = 64 c + lane (lanes with i >= n hold
// nothing); lane 0 writes the cluster record and returns the cluster's smallest r^2 in every lane.
__device__ __forceinline__ float cluster_box_of_wave(const float4 s, const bool have, const size_t c,
                                                     float4* __restrict__ C)
{
    const int lane = threadIdx.x & 63;
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    float r2_min = INFINITY;
    if (have) {
        r2_min = s.w;
        const float r = sqrtf(s.w) * 1.00001f;   // sqrt(fl(h h)) can round below h
        const float ctr[3] = { s.x, s.y, s.z };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float slack = (fabsf(ctr[k]) + r) * 4.76837158203125e-07f; // 2^-21
            lo[k] = (ctr[k] - r) - slack;
            hi[k] = (ctr[k] + r) + slack;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) r2_min = fminf(r2_min, __shfl_xor(r2_min, off));
    if (lane == 0) {
        // .w of the low corner: the smallest r^2 of the members (the origin-lattice cull of
        // axis-aligned packets looks only at clusters that hold spheres smaller than the
        // packet's ray spacing)
        C[2 * c] = make_float4(lo[0], lo[1], lo[2], r2_min);
        C[2 * c + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
    return r2_min;
}

// The scene's smallest r^2 (positive floats order like their bit patterns): one float4 past the
// last cluster record, pre-set to a huge value by scene_fill; one atomic per workgroup (one per
// cluster serialised 156 k atomics on one address at 10^7 primitives: +1.8 ms per unprepared call).
// Every thread of the workgroup must call it.
__device__ __forceinline__ void publish_r2_min(const float wave_r2_min, float4* __restrict__ C_tail)
{
    __shared__ float s_r2_min[4];
    if ((threadIdx.x & 63) == 0) s_r2_min[(threadIdx.x >> 6) & 3] = wave_r2_min;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = s_r2_min[0];
        for (unsigned w = 1; w < blockDim.x / 64 && w < 4; ++w) m = fminf(m, s_r2_min[w]);
        if (m < INFINITY) atomicMin(reinterpret_cast<unsigned int*>(C_tail), __float_as_uint(m));
    }
}

// C != null: the cluster boxes in the same pass (a wave's 64 consecutive records ARE a cluster:
// the loop stride is a multiple of the workgroup size) -- one launch and one read of A fewer.
// B1 = {1/h, 1/h^2} (the reference's per-hit arithmetic), B50 = {50/h, 1/h^2} (the fast integral's
// table position): either or both.
__global__ __launch_bounds__(256) void trace_prepass_kernel(const float4* __restrict__ spheres,
                                                            size_t n, float4* __restrict__ A,
                                                            float2* __restrict__ B1,
                                                            float2* __restrict__ B50,
                                                            float4* __restrict__ C,
                                                            const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
    float wave_r2_min = INFINITY;
    // (whole waves enter every iteration: the bound is rounded up to the wave's first record)
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; (i & ~size_t(63)) < n + 4;
         i += size_t(gridDim.x) * blockDim.x) {

        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 b1 = make_float2(0.f, 0.f), b50 = b1;
        if (i < n) {
            const float4 s = spheres[i];
            a = make_float4(s.x, s.y, s.z, s.w * s.w); // sphere.w * sphere.w, intersect.h:37
            const float ir = 1.f / s.w;                // functors/trace.cuh:181
            b1 = make_float2(ir, ir * ir);             // functors/trace.cuh:184
            // The fast integral's branch-free rounds add (table value 0) * (1/h^2) for a
            // candidate the ray misses: keep that product 0 when 1/h^2 overflows (h < 5e-20).
            b50 = make_float2(ir * float(N_TABLE - 1), fminf(ir * ir, 3.4028234664e38f));
        }
        if (i < n + 4) {
            A[i] = a;
            if (B1) B1[i] = b1;
            if (B50) B50[i] = b50;
        }
        // (wave-uniform condition: every lane of the wave takes part in the shuffles)
        if (C && (i & ~size_t(63)) < n) wave_r2_min = fminf(wave_r2_min, cluster_box_of_wave(a, i < n, i >> 6, C));
    }
    if (C) publish_r2_min(wave_r2_min, C + 2 * ((n + 63) / 64));
}

// double4 spheres: the float record {x, y, z, r^2} that drives the walk's culls must CONTAIN the
// double sphere -- the centre is narrowed (error <= half a float ulp per co-ordinate) and the
// double hit test is close to exact, so the radius is inflated by 2^-18 relative plus 2^-21 of
// the co-ordinate magnitudes before squaring (the float culls' own margins then cover their own
// rounding as for float spheres).
__global__ __launch_bounds__(256) void trace_prepass_d4_kernel(const double* __restrict__ spheres,
                                                               size_t n, float4* __restrict__ A)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n + 4;
         i += size_t(gridDim.x) * blockDim.x) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            const double* s = spheres + 4 * i;
            const double r = fabs(s[3]) * (1.0 + 3.814697265625e-06)
                + (fabs(s[0]) + fabs(s[1]) + fabs(s[2]) + fabs(s[3])) * 4.76837158203125e-07;
            a = make_float4(float(s[0]), float(s[1]), float(s[2]), float(r * r * (1.0 + 1e-6)));
        }
        A[i] = a;
    }
}

// ---- cluster boxes ---------------------------------------------------------------------------
// Primitives are Morton-sorted, so 64 consecutive ones (a CLUSTER: indices [64 c, 64 c + 64)) are
// a compact clump about as wide as a smoothing length.  One box per cluster -- the union of
// the member spheres' boxes [c - h, c + h] -- lets a sweep drop 64 candidates with one lane's
// test instead of 64 lanes' tests: a swept subtree first tests its clusters (lane j <-> cluster
// j), then runs culling rounds only over the clusters that survive.  Never a result: a cluster
// is dropped only if no ray of the packet can hit any member, so per-ray hit sets are unchanged.
// The half-width is inflated by 4 ulp of the co-ordinate magnitude: sphere_hit's own rounding
// (q = fl(s - o), b2 = fl(fl(q1^2) + fl(q2^2)) < h^2 admits |s - o| up to h (1 + 3 u) + u |s|).
// A = {x, y, z, r^2} as written by the pre-passes (spheres: r = h; triangles: bounding radius).
__global__ __launch_bounds__(256) void cluster_boxes_kernel(const float4* __restrict__ A, size_t n,
                                                            float4* __restrict__ C,
                                                            const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
    const size_t n_clusters = (n + 63) / 64;
    const int lane = threadIdx.x & 63;
    float wave_r2_min = INFINITY;   // over the clusters this wave handles
    for (size_t c = blockIdx.x * size_t(blockDim.x / 64) + (threadIdx.x >> 6); c < n_clusters;
         c += size_t(gridDim.x) * (blockDim.x / 64)) {
        const size_t i = c * 64 + lane;
        const bool have = i < n;
        const float4 s = have ? A[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        wave_r2_min = fminf(wave_r2_min, cluster_box_of_wave(s, have, c, C));
    }
    publish_r2_min(wave_r2_min, C + 2 * n_clusters);
}

// Group boxes: the union of the boxes of 2^(shift - 6) consecutive clusters, one wave per group
// (fminf / fmaxf like the cluster boxes: a NaN member is left out, as it is there).
__global__ __launch_bounds__(256) void group_boxes_kernel(const float4* __restrict__ C, size_t n_clusters,
                                                          int per_group, size_t n_groups,
                                                          float4* __restrict__ G,
                                                          const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
    const int lane = threadIdx.x & 63;
    for (size_t g = blockIdx.x * size_t(blockDim.x / 64) + (threadIdx.x >> 6); g < n_groups;
         g += size_t(gridDim.x) * (blockDim.x / 64)) {
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        const size_t c0 = g * size_t(per_group);
        for (size_t c = c0 + lane; c < c0 + per_group && c < n_clusters; c += 64) {
            const float4 l = C[2 * c], h = C[2 * c + 1];
            lo[0] = fminf(lo[0], l.x); lo[1] = fminf(lo[1], l.y); lo[2] = fminf(lo[2], l.z);
            hi[0] = fmaxf(hi[0], h.x); hi[1] = fmaxf(hi[1], h.y); hi[2] = fmaxf(hi[2], h.z);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
            }
        }
        if (lane == 0) {
            G[2 * g] = make_float4(lo[0], lo[1], lo[2], 0.f);
            G[2 * g + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        }
    }
}

// ---- triangle primitives (tests/profile_trace_triangle) -----------------------------------
// Pre-pass: a bounding sphere per triangle (for the beam culling: a ray that meets the
// triangle passes within r of the centre; r^2 is inflated by 2^-10 against fp32 rounding) and
// the triangle widened to fp64 (the reference's dot/cross products are fp64 products of
// float operands, tests/helper/vector_math.cu:27-52; widening once is exact).
__global__ __launch_bounds__(256) void tri_prepass_kernel(const float* __restrict__ tris, size_t n,
                                                          float4* __restrict__ A,
                                                          double* __restrict__ T64,
                                                          const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n + 4;
         i += size_t(gridDim.x) * blockDim.x) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) {
            const float* t = tris + 9 * i;
            float c[3], r2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) c[k] = t[k] + (t[3 + k] + t[6 + k]) * (1.0f / 3.0f);
#pragma unroll
            for (int vtx = 0; vtx < 3; ++vtx) {
                float d2 = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float p = (vtx == 0 ? t[k] : vtx == 1 ? t[k] + t[3 + k] : t[k] + t[6 + k]) - c[k];
                    d2 += p * p;
                }
                r2 = fmaxf(r2, d2);
            }
            a = make_float4(c[0], c[1], c[2], r2 * 1.0009765625f + 1e-37f);
#pragma unroll
            for (int k = 0; k < 9; ++k) T64[9 * i + k] = double(t[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) T64[9 * i + k] = 0.0;
        }
        A[i] = a;
    }
}

// Primitive range of every node: a node's leaves are consecutive (nodes.h:27-28) and so are
// their primitives, [leaves[first].x, leaves[last].x + leaves[last].y).
__global__ __launch_bounds__(256) void node_prims_kernel(const int4* __restrict__ nodes4,
                                                         const int4* __restrict__ leaves, int n_nodes,
                                                         int2* __restrict__ out,
                                                         uint32_t* __restrict__ r2_min_slot,
                                                         const uint32_t* __restrict__ run_if)
{
    if (run_if && *run_if == 0u) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // (the slot the cluster-box pass reduces the scene's smallest r^2 into starts out huge: this
    // launch precedes that pass on the stream)
    if (i == 0 && r2_min_slot) *r2_min_slot = 0x7f7f7f7fu;
    if (i >= n_nodes) return;
    const int4 n0 = nodes4[4 * size_t(i)];
    const int4 lf = leaves[n0.z], ll = leaves[n0.w];
    out[i] = make_int2(lf.x, ll.x + ll.y - lf.x);
}

} // namespace

namespace grace_hip {


static grace_status scene_free_buffers(Scene& sc)
{
    void* bufs[] = { sc.A, sc.B1, sc.B50, sc.T64, sc.node_prims, sc.C, sc.ctl };
    bool any = false;
    for (void* b : bufs) any = any || b;
    if (any) GRACE_TRY_HIP(hipDeviceSynchronize());
    for (void* b : bufs)
        if (b) GRACE_TRY_HIP(hipFree(b));
    sc.A = nullptr; sc.B1 = sc.B50 = nullptr; sc.T64 = nullptr; sc.node_prims = nullptr; sc.C = nullptr;
    sc.ctl = nullptr;
    sc.valid = false;
    sc.pinned = false;
    return GRACE_OK;
}

grace_status scene_release(TraceState& ts)
{
    GRACE_TRY(scene_free_buffers(ts.scene));
    ts.scene = Scene();
    return GRACE_OK;
}

grace_status scene_cache_alloc(TraceState& ts, const SceneKey& key)
{
    Scene& sc = ts.scene;
    GRACE_TRY(scene_free_buffers(sc));
    const bool tri = key.kind == 1;
    auto alloc = [&](void** ptr, size_t bytes) -> grace_status {
        hipError_t e = hipMalloc(ptr, bytes);
        if (e != hipSuccess)
            return set_error(GRACE_OUT_OF_MEMORY, __FILE__, __LINE__, hipGetErrorString(e));
        return GRACE_OK;
    };
    const size_t n_prims = key.n_prims;
    grace_status st = alloc(reinterpret_cast<void**>(&sc.A), (n_prims + 4) * sizeof(float4));
    if (st == GRACE_OK && !tri) st = alloc(reinterpret_cast<void**>(&sc.B1), (n_prims + 4) * sizeof(float2));
    if (st == GRACE_OK && !tri) st = alloc(reinterpret_cast<void**>(&sc.B50), (n_prims + 4) * sizeof(float2));
    if (st == GRACE_OK && tri) st = alloc(reinterpret_cast<void**>(&sc.T64), 72 * (n_prims + 4));
    if (st == GRACE_OK) st = alloc(reinterpret_cast<void**>(&sc.node_prims), key.n_nodes * sizeof(int2));
    if (st == GRACE_OK) st = alloc(reinterpret_cast<void**>(&sc.C), cluster_record_count(n_prims) * sizeof(float4));
    if (st == GRACE_OK) st = alloc(reinterpret_cast<void**>(&sc.ctl), sizeof(CacheCtl));
    if (st != GRACE_OK) { (void)scene_free_buffers(sc); return st; }
    sc.key = key;
    return GRACE_OK;
}

grace_status scene_fill(int kind, const void* prims, size_t n_prims, const float4* nodes,
                        size_t n_nodes, const int4* leaves, float4* A, float2* B1, float2* B50,
                        double* T64, int2* node_prims, float4* C, hipStream_t stream,
                        const uint32_t* run_if)
{
    node_prims_kernel<<<ceil_div(n_nodes, 256), 256, 0, stream>>>(
        reinterpret_cast<const int4*>(nodes), leaves, int(n_nodes), node_prims,
        reinterpret_cast<uint32_t*>(C + 2 * ((n_prims + 63) / 64)), run_if);
    GRACE_CHECK_LAUNCH();
    if (kind == 1) {
        tri_prepass_kernel<<<stream_grid(n_prims + 4, 256), 256, 0, stream>>>(
            static_cast<const float*>(prims), n_prims, A, T64, run_if);
        GRACE_CHECK_LAUNCH();
    } else if (kind == 2) {
        trace_prepass_d4_kernel<<<stream_grid(n_prims + 4, 256), 256, 0, stream>>>(
            static_cast<const double*>(prims), n_prims, A);
        GRACE_CHECK_LAUNCH();
    } else {
        // (+ the cluster boxes: fused)
        trace_prepass_kernel<<<stream_grid(n_prims + 4, 256), 256, 0, stream>>>(
            static_cast<const float4*>(prims), n_prims, A, B1, B50, C, run_if);
        GRACE_CHECK_LAUNCH();
    }
    if (kind != 0) {
        cluster_boxes_kernel<<<stream_grid((n_prims + 63) / 64, 4), 256, 0, stream>>>(A, n_prims, C, run_if);
        GRACE_CHECK_LAUNCH();
    }
    {
        const size_t n_clusters = (n_prims + 63) / 64, n_groups = group_count(n_prims);
        group_boxes_kernel<<<stream_grid(n_groups, 4), 256, 0, stream>>>(
            C, n_clusters, 1 << (group_shift(n_prims) - 6), n_groups, C + 2 * n_clusters + 1, run_if);
        GRACE_CHECK_LAUNCH();
    }
    return GRACE_OK;
}

// grace_trace_prepare_f4 / _tri: the cache filled NOW (and kept until released or replaced by
// another prepare), instead of at the second call on the same arrays.
grace_status scene_prepare(TraceState& ts, bool tri, const void* prims, size_t n_prims,
                           const int* d_nodes, size_t n_nodes, const int* d_leaves, hipStream_t stream)
{
    GRACE_REQUIRE(prims && d_nodes && d_leaves, "trace_prepare: null pointer");
    GRACE_REQUIRE(n_prims > 0 && n_nodes >= 1, "trace_prepare: empty scene");
    SceneKey key;
    key.kind = tri ? 1 : 0; key.prims = prims; key.nodes = d_nodes; key.leaves = d_leaves;
    key.n_prims = n_prims; key.n_nodes = n_nodes;
    GRACE_TRY(scene_cache_alloc(ts, key));
    Scene& sc = ts.scene;
    FrameGuard frame;
    grace_status st = frame.begin(Workspace::aligned(sig_partial_words() * 8) + 1024, stream);
    if (st == GRACE_OK) {
        unsigned long long* partial = Workspace::take<unsigned long long>(sig_partial_words());
        SigRequest rq;
        rq.prims = prims; rq.prims_bytes = n_prims * (tri ? 36 : 16);
        rq.nodes = d_nodes; rq.nodes_bytes = n_nodes * 64;
        rq.leaves = d_leaves; rq.leaves_bytes = (n_nodes + 1) * 16;
        rq.scene_ctl = sc.ctl; rq.scene_force = true;
        st = launch_signatures(rq, partial, stream);
    }
    if (st == GRACE_OK)
        st = scene_fill(tri ? 1 : 0, prims, n_prims, reinterpret_cast<const float4*>(d_nodes), n_nodes,
                        reinterpret_cast<const int4*>(d_leaves), sc.A, sc.B1, sc.B50, sc.T64,
                        sc.node_prims, sc.C, stream);
    if (st != GRACE_OK) { (void)scene_release(ts); return st; }
    sc.valid = true;
    sc.pinned = true;
    ts.scene.seen = key;
    return GRACE_OK;
}

} // namespace grace_hip
