// BVH traversal with ray-sphere tests and SPH-kernel line integrals, for gfx950.
//
// Replaces trace_kernel (reference include/grace/cuda/kernels/bintree_trace.cuh:52-197) as
// instantiated by trace_hitcounts_sph / trace_cumulative_sph / trace_sph
// (include/grace/cuda/trace_sph.cuh:58-168) with the functors of
// include/grace/cuda/functors/trace.cuh, AABBs_hit (include/grace/cuda/device/intersect.cuh:
// 10-40) and sphere_hit (include/grace/generic/intersect.h:10-55).
//
// Same traversal semantics: a packet of rays shares one stack; a child is pushed if ANY
// ray of the packet hits its box (right first, so the left subtree is walked first and
// every ray meets its hits in ascending primitive index); every ray of the packet is
// tested against every sphere of every leaf the packet enters.  Results per ray therefore
// equal the brute-force loop over all spheres (the reference's own criterion,
// tests/tree_traversal/tree_traversal.cu:65-100).
//
// CDNA4 design (not the reference's; measurements in DESIGN.md section 4/6):
//   * packet = one 64-lane wavefront (the reference: a 32-thread warp); for the per-hit and
//     triangle walks of small batches, 32 or 16 rays per wave;
//   * the packet's stack lives in TWO VGPRs indexed by lane (pop = v_readlane, push =
//     lane-select, with a scalar stack pointer): 128 entries, no LDS traffic;
//   * node records are wave-uniform: child indices / primitive spans are consumed as scalars,
//     the per-ray slab test keeps the reference's arithmetic (~100 node tests per packet);
//   * no FMA contraction (-ffp-contract=off), IEEE 1/x and sqrt in everything that decides a
//     hit or feeds a bit-exact output: the reference's CPU/GPU equality test is built with
//     -fmad=false (tests/tree_traversal/Makefile:5-8).  Explicit FMAs appear only where they
//     provably return the unfused bits (axis dot product), in culls, and in the fast integral;
//   * a streaming pre-pass hoists per-sphere work out of the (ray x sphere) loop:
//     A[i] = {x, y, z, h*h}, B[i] = {1/h or 50/h, (1/h)^2};
//   * ray coherence order: 64 consecutive rays of a space-filling order of the ray
//     co-ordinates that vary form a packet (15-bit round-to-nearest keys; a Hilbert curve for
//     two varying co-ordinates and, over the octahedral map of the direction, for batches from
//     one origin -- its runs never join distant patches; Z-order for power-of-two pixel grids,
//     whose tiles are the same and whose tile order suits the dispatcher better, and for
//     general batches); results do not depend on it; a batch traced repeatedly can have its
//     order prepared once (grace_trace_prepare_rays);
//   * hit counts and column densities do not walk the tree: primitives are Morton-sorted, so
//     groups of 4096 consecutive ones are compact cells; a packet tests all group boxes up
//     front, 64 per lane-parallel pass (boxes behind the cluster records, masks kept in LDS),
//     and sweeps the surviving groups in ascending order (trace_kernel.hpp);
//   * the per-hit traces, triangles and the stats walk descend the tree to subtrees of <= T
//     primitives (contiguous indices; T = 8192, 512 for triangles) and sweep those;
//   * a sweep (of a group or a subtree) has a cluster level: one box per 64 consecutive
//     primitives (a pre-pass) is tested first, lane j for cluster j, and only the surviving
//     clusters go through culling rounds of 64;
//   * scenes with spheres smaller than the ray spacing (dense cores of clustered SPH data) run
//     a separate instantiation (LAT), selected by a device flag: an exact cull against the
//     packet's origin lattice, and four waves per packet for big batches (their packets are
//     very unequal);
//   * beam culling per round, lane j deciding for candidate j whether ANY ray of the packet
//     can hit it: exact lower bound of b^2 for axis-aligned packets (rounding is monotone),
//     cone + four side planes for packets from one origin, interval arithmetic otherwise;
//   * survivors of a round are written compacted (slot = v_mbcnt) to the wave's LDS tile as
//     three 8-byte planes and read back two survivors ahead into three rotating register sets
//     (unconditional reads pinned by sched_barrier; immediates address the slots);
//   * axis-aligned packets (orthographic projections): sphere_hit collapses exactly under IEEE
//     rules to dot = fma(s_k, d_k, -o_k d_k), b2 = q1^2 + q2^2; rounds whose candidates are
//     provably inside every ray's [0, length) skip the range tests (6 instructions per test);
//   * kernel integral of the column-density trace: hardware sqrt + fp32 lerp of an fp32
//     (y, dy) table in LDS (7 instructions, tolerance parity) by default; the reference's
//     arithmetic bit for bit (correctly rounded sqrt, fp64 lerp) on request and always for
//     the per-hit outputs;
//   * class-ordered sums and packet splitting: primitives are dealt to 8 CLASSES in granules
//     of 1024 consecutive indices (class = (index >> 10) & 7).  A ray's column density is
//     defined as the balanced pairwise (binary-tree) fp32 sum of its 8 class sums, each class
//     summed in ascending primitive order.  The value is a function of the ray and the scene
//     only -- not of how rays are batched, ordered or sharded -- and differs from the
//     reference's single running sum only in the last bits (both within 1e-6 of the exact
//     sum; tolerance 1e-5).  What it buys: a packet can be walked by K = 1, 2, 4 or 8
//     waves, wave k owning the 8/K classes [k 8/K, (k+1) 8/K) -- a subtree of the summation
//     tree, and, because classes interleave along the Morton order, an even share of the
//     work for any beam -- with NO change of the result: each wave reduces its classes, a
//     tiny kernel finishes the tree.  K is the smallest power of two that puts >= 16384 waves
//     in flight; for batches of one direction (light packets) a device-side choice lowers it
//     (choose_split).  A wave keeps its class accumulators in LDS (2 KiB) and switches at
//     granule boundaries, once per culling round at most.  Hit counts split the same way
//     (integers);
//   * per-hit outputs (ordered per ray): large batches stage hits per lane in LDS and drain
//     them eight entries per ray; small batches split a packet over K waves by contiguous
//     chunk ranges with per-(ray, chunk) output offsets from a counting walk (hits_plan_kernel);
//   * packets are dealt to workgroups so that the workgroups sharing an XCD (blockIdx % 8)
//     walk a contiguous range of packets: neighbouring packets touch the same subtree and
//     each XCD's 4 MiB L2 keeps it.
//
// Roofline: divergent tree walk, integer/fp32 scalar-operand work -- no MFMA.  Algorithmic
// bytes per ray: 28 + 64 * nodes + 16 * leaves + 16 * spheres tested + 4,
// counted per ray by the `stats` instantiation below.
#include "trace_kernel.hpp"
#include "trace_plan.hpp"

#include <cstdlib>

using namespace grace_hip;

namespace grace_hip {

static grace_status trace_state_destroy(Context& c)
{
    if (!c.trace) return GRACE_OK;
    TraceState& ts = *c.trace;
    GRACE_TRY(scene_release(ts));
    GRACE_TRY(rays_release(ts));
    if (ts.hits.chunk_counts) GRACE_TRY_HIP(hipFree(ts.hits.chunk_counts));
    if (ts.status) GRACE_TRY_HIP(hipFree(ts.status));
    if (ts.ev0) GRACE_TRY_HIP(hipEventDestroy(ts.ev0));
    if (ts.ev1) GRACE_TRY_HIP(hipEventDestroy(ts.ev1));
    delete c.trace;
    c.trace = nullptr;
    return GRACE_OK;
}

grace_status trace_state(TraceState** out)
{
    Context* c = nullptr;
    GRACE_TRY(current_context(&c));
    if (!c->trace) {
        c->trace = new TraceState();
        g_trace_state_destroy = trace_state_destroy;
    }
    *out = c->trace;
    return GRACE_OK;
}

} // namespace grace_hip

namespace {

grace_status ensure_status(TraceState& ts, hipStream_t stream)
{
    if (!ts.status) {
        GRACE_TRY_HIP(hipMalloc(reinterpret_cast<void**>(&ts.status), sizeof(int)));
        GRACE_TRY_HIP(hipMemsetAsync(ts.status, 0, sizeof(int), stream));
    }
    return GRACE_OK;
}

template <int MODE>
grace_status launch_trace(TraceArgs a, size_t n_rays, size_t n_spheres, size_t n_nodes,
                          hipStream_t stream)
{
    GRACE_REQUIRE(a.rays && a.spheres && a.nodes && a.leaves && a.root, "trace: null pointer");
    GRACE_REQUIRE(n_rays < (size_t(1) << 31), "trace: bad ray count");
    GRACE_REQUIRE(n_nodes >= 1 && n_nodes < (size_t(1) << 30), "trace: bad node count");
    GRACE_REQUIRE(n_spheres > 0 && n_spheres < (size_t(1) << 31), "trace: bad primitive count");
    TraceState* ts_ptr = nullptr;
    GRACE_TRY(trace_state(&ts_ptr));
    TraceState& ts = *ts_ptr;
    GRACE_TRY(ensure_status(ts, stream));
    FrameGuard frame;
    // Split per-hit trace for small batches (see TraceArgs / hits_plan_kernel): chunk size =
    // a power of two >= one granule giving at most MAX_HIT_CHUNKS chunks.
    int hit_chunk_shift = GRANULE_SHIFT;
    while ((((n_spheres - 1) >> hit_chunk_shift) + 1) > size_t(MAX_HIT_CHUNKS)) ++hit_chunk_shift;
    const int hit_chunks = int(((n_spheres - 1) >> hit_chunk_shift) + 1);
    const size_t hit_packets = ceil_div(n_rays, size_t(64));
    int hit_split = 1;
    if (MODE == MODE_HITS && ts.width <= 0 && hit_packets < 4096 && hit_chunks >= 8) {
        if (ts.split > 0) hit_split = ts.split;
        else while (hit_split < 8 && hit_packets * hit_split < 16384) hit_split *= 2;
    }
    const bool hits_split = hit_split > 1;
    int* chunk_counts = nullptr; int* chunk_off = nullptr;
    int* scratch_counts = nullptr;
    int4* wave_map = nullptr; int* n_wave_map = nullptr;
    uint32_t* pk_prefix = nullptr; uint32_t* pk_total = nullptr; int* pk_first = nullptr; int* pk_parts = nullptr;
    // Would the per-hit trace of this batch use the split path?  (the same rule, for the hit-count
    // call that is asked to keep its chunk counts)
    const bool keep_chunks = MODE == MODE_COUNT && a.keep_chunks && ts.width <= 0 && hit_packets < 4096
        && hit_chunks >= 8 && ts.split != 1;
    const bool reuse_chunks = MODE == MODE_HITS && hits_split && ts.hits.valid && ts.hits.rays == a.rays
        && ts.hits.n_rays == n_rays && ts.hits.prims == static_cast<const void*>(a.spheres)
        && ts.hits.n_prims == n_spheres && ts.hits.n_chunks == hit_chunks;
    if (MODE == MODE_HITS || MODE == MODE_COUNT) ts.hits.valid = false;   // consumed, or stale from here on
    if (keep_chunks) {
        const size_t need = n_rays * size_t(hit_chunks);
        if (ts.hits.capacity < need) {
            if (ts.hits.chunk_counts) { GRACE_TRY_HIP(hipDeviceSynchronize()); GRACE_TRY_HIP(hipFree(ts.hits.chunk_counts)); }
            ts.hits.chunk_counts = nullptr; ts.hits.capacity = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&ts.hits.chunk_counts), need * sizeof(int));
            if (e != hipSuccess) return set_error(GRACE_OUT_OF_MEMORY, __FILE__, __LINE__, hipGetErrorString(e));
            ts.hits.capacity = need;
        }
    }
    // Per-hit and triangle traces cannot split a packet among waves (their outputs are ordered
    // / reduced per ray inside one wave); with few rays they use narrower packets instead:
    // 2-4x the waves, each with a tighter beam, on a chip that would otherwise sit idle.
    int width = 64;
    if (ts.width > 0) width = ts.width;
    else if ((MODE == MODE_HITS && !hits_split) || MODE == MODE_TRI || MODE == MODE_HITS_D4)
        while (width > 16 && ceil_div(n_rays, size_t(width)) < 4096) width /= 2;
    // Hit counts and column densities split packets eight ways at most; a batch too small to fill
    // the chip even then (< 512 packets) also gets narrower packets (10^7 particles, 12288 HEALPix
    // rays: 3.5 -> 2.0 ms at 16 rays per packet; from 49152 rays on it loses: config 3 0.87 -> 0.95 ms).
    else if ((MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_COUNT_D4 || MODE == MODE_CUM_D4)
             && ts.split <= 0)
        while (width > 16 && ceil_div(n_rays, size_t(width)) * SUM_CLASSES < 4096) width /= 2;
    a.width = width;
    const int n_packets = ceil_div(n_rays, size_t(width));
    // Waves per packet: two resident sets of waves (2 x 8192) for small ray batches.
    int split = 1;
    // (double4 hit counts and column densities split the same way: the same classes, summed in double)
    constexpr bool CLASS_SPLIT = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_COUNT_D4
                                  || MODE == MODE_CUM_D4);
    if (CLASS_SPLIT) {
        // (column densities: a batch of exactly 16384 packets -- the 1024^2 frame -- still gets a
        // second wave per packet, 32768 waves; see choose_split.  Larger batches run one.)
        const size_t wave_budget = (MODE == MODE_CUMULATIVE) ? 16385 : 16384;
        if (ts.split > 0) split = ts.split;
        else {
            while (split < SUM_CLASSES && size_t(n_packets) * split < wave_budget) split *= 2;
            // (the device picks the working waves per packet: scenes with spheres smaller than the ray
            // spacing want four -- see lat_split below --, one-direction batches two, others one)
            if (MODE == MODE_CUMULATIVE && split > 1 && split < ts.lat_split && ts.ray_reorder && n_rays > 64
                && width == 64)
                split = ts.lat_split;
        }
    }
    if (hits_split) split = hit_split;
    a.split = split;
    a.split_dev = nullptr;
    // (the working waves per packet are chosen on the device, by ray_keys_kernel)
    const bool dev_split = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE) && split > 1 && ts.split <= 0;
    {
        constexpr bool need_b = (MODE == MODE_CUMULATIVE || MODE == MODE_HITS);
        const bool fast_b = MODE == MODE_CUMULATIVE && !ts.exact_integrals;
        const bool reorder = ts.ray_reorder && n_rays > 64;
        constexpr bool D4 = (MODE == MODE_COUNT_D4 || MODE == MODE_CUM_D4 || MODE == MODE_HITS_D4);
        // ---- which cached records does this call use?  (see trace_state.hpp) -------------------
        // NONE: derive into the workspace (a scene / batch seen for the first time);  FILL: the
        // same arrays as the previous call -- derive into the cache;  CHECK: cached -- validate by
        // signature on the device, the gated pre-pass recomputes if stale;  TRUST: cached, and the
        // caller has switched validation off.
        enum { USE_NONE, USE_FILL, USE_CHECK, USE_TRUST };
        auto decide = [&](const bool have, const bool pinned, const bool repeat) {
            if (have) return ts.cache_validation ? USE_CHECK : USE_TRUST;
            return (ts.cache_auto && repeat && !pinned) ? USE_FILL : USE_NONE;
        };
        int scene_use = USE_NONE, rays_use = USE_NONE;
        SceneKey skey;
        skey.kind = MODE == MODE_TRI ? 1 : 0;
        skey.prims = a.spheres; skey.nodes = a.nodes; skey.leaves = a.leaves;
        skey.n_prims = n_spheres; skey.n_nodes = n_nodes;
        if (!D4 && MODE != MODE_STATS) {
            scene_use = decide(ts.scene.valid && ts.scene.key == skey, ts.scene.valid && ts.scene.pinned,
                               ts.scene.seen == skey);
            ts.scene.seen = skey;
            if (scene_use == USE_FILL && scene_cache_alloc(ts, skey) != GRACE_OK) scene_use = USE_NONE;   // (no memory: no cache)
        }
        RayKey rkey;
        rkey.rays = a.rays; rkey.n = n_rays;
        if (reorder) {
            rays_use = decide(ts.rays.valid && ts.rays.key == rkey, ts.rays.valid && ts.rays.pinned,
                              ts.rays.seen == rkey);
            ts.rays.seen = rkey;
            if (rays_use == USE_FILL && rays_cache_alloc(ts, rkey) != GRACE_OK) rays_use = USE_NONE;
        }
        const bool scene_cached = scene_use != USE_NONE, rays_cached = rays_use != USE_NONE;
        const bool any_sig = scene_use == USE_FILL || scene_use == USE_CHECK || rays_use == USE_FILL || rays_use == USE_CHECK;
        const size_t n_clusters = (n_spheres + 63) / 64;
        GRACE_TRY(frame.begin((scene_cached ? 0 : Workspace::aligned((n_spheres + 4) * sizeof(float4))
                                               + Workspace::aligned((n_spheres + 4) * sizeof(float2))
                                               + Workspace::aligned(n_nodes * sizeof(int2))
                                               + Workspace::aligned(cluster_record_count(n_spheres) * sizeof(float4))
                                               + (MODE == MODE_TRI ? Workspace::aligned(72 * (n_spheres + 4)) : 0))
                                   + (hits_split ? 2 * Workspace::aligned(n_rays * size_t(hit_chunks) * 4)
                                                   + Workspace::aligned(hit_packets * hit_split * sizeof(int4))
                                                   + Workspace::aligned(hit_packets * size_t(hit_chunks) * 4)
                                                   + 4 * Workspace::aligned(hit_packets * 4 + 64)
                                                   + Workspace::aligned(n_rays * 4) : 0)
                                   + (MODE == MODE_CUMULATIVE ? Workspace::aligned(n_rays * SUM_CLASSES * 4) : 0)
                                   + (MODE == MODE_CUM_D4 ? Workspace::aligned(n_rays * SUM_CLASSES * 8) : 0)
                                   + (reorder ? 2 * Workspace::aligned(n_rays * 4)
                                                + sort_ws_bytes(n_rays, 4, 0) : 0)
                                   + (any_sig ? Workspace::aligned(sig_partial_words() * 8) : 0) + 1024, stream));
        if (any_sig) {
            SigRequest rq;
            if (scene_use == USE_FILL || scene_use == USE_CHECK) {
                rq.prims = a.spheres; rq.prims_bytes = n_spheres * (MODE == MODE_TRI ? 36 : 16);
                rq.nodes = a.nodes; rq.nodes_bytes = n_nodes * 64;
                rq.leaves = a.leaves; rq.leaves_bytes = (n_nodes + 1) * 16;
                rq.scene_ctl = ts.scene.ctl; rq.scene_force = scene_use == USE_FILL;
            }
            if (rays_use == USE_FILL || rays_use == USE_CHECK) {
                rq.rays = a.rays; rq.rays_bytes = n_rays * 28;
                rq.rays_ctl = ts.rays.ctl; rq.rays_force = rays_use == USE_FILL; rq.rays_ext = ts.rays.ext;
            }
            GRACE_TRY(launch_signatures(rq, Workspace::take<unsigned long long>(sig_partial_words()), stream));
        }
        if (scene_cached) {
            Scene& sc = ts.scene;
            if (scene_use != USE_TRUST) {
                // (gated by the cache's stale flag: a first fill is forced stale)
                GRACE_TRY(scene_fill(skey.kind, a.spheres, n_spheres, a.nodes, n_nodes, a.leaves, sc.A, sc.B1, sc.B50,
                                     sc.T64, sc.node_prims, sc.C, stream, &sc.ctl->stale));
                sc.valid = true;
            }
            a.A = sc.A;
            a.B = need_b ? (fast_b ? sc.B50 : sc.B1) : nullptr;
            a.T64 = sc.T64;
            a.node_prims = sc.node_prims;
            a.C = sc.C;
        } else {
            float4* A = Workspace::take<float4>(n_spheres + 4);
            float2* B = need_b ? Workspace::take<float2>(n_spheres + 4) : nullptr;
            double* T64 = (MODE == MODE_TRI) ? Workspace::take<double>(9 * (n_spheres + 4)) : nullptr;
            int2* node_prims = Workspace::take<int2>(n_nodes);
            float4* C = Workspace::take<float4>(cluster_record_count(n_spheres));
            GRACE_TRY(scene_fill(MODE == MODE_TRI ? 1 : D4 ? 2 : 0,
                                 D4 ? static_cast<const void*>(a.spheres_d) : a.spheres, n_spheres, a.nodes, n_nodes, a.leaves, A,
                                 fast_b ? nullptr : B, fast_b ? B : nullptr, T64, node_prims, C, stream));
            a.A = A; a.B = B; a.T64 = T64; a.node_prims = node_prims; a.C = C;
        }
        a.partial = (MODE == MODE_CUMULATIVE) ? Workspace::take<float>(n_rays * SUM_CLASSES) : nullptr;
        a.partial_d = (MODE == MODE_CUM_D4) ? Workspace::take<double>(n_rays * SUM_CLASSES) : nullptr;
        if (hits_split) {
            chunk_counts = Workspace::take<int>(n_rays * size_t(hit_chunks));
            chunk_off = Workspace::take<int>(n_rays * size_t(hit_chunks));
            wave_map = Workspace::take<int4>(hit_packets * hit_split);
            pk_prefix = Workspace::take<uint32_t>(hit_packets * size_t(hit_chunks));
            pk_total = Workspace::take<uint32_t>(hit_packets + 16);
            pk_first = Workspace::take<int>(hit_packets + 16);
            pk_parts = Workspace::take<int>(hit_packets + 16);
            n_wave_map = Workspace::take<int>(16);
            scratch_counts = Workspace::take<int>(n_rays);
        }
        // Subtrees of up to this many primitives are swept -- cluster tests, then culling rounds
        // over the surviving clusters -- rather than descended.
        // Axis-aligned packets test a cluster's box against their origin rectangle (sharp: large
        // subtrees pay; 16384 measured best in round 2); pencil packets test it
        // against the bundle's side planes, general packets its circumscribed sphere.
        // (re-measured after the pencil cluster test became a box-against-side-planes test: sphere
        // scenes now prefer 8192 there too -- config 2 column densities 1.74 -> 1.46 ms, hit counts
        // 1.54 -> 1.19, config 3 0.94 -> 0.87 --; triangles, culled through bounding spheres, keep 512:
        // 5.3 / 4.6 / 2.8 ms for the three cameras against 6.9 / 5.8 / 3.0 at 8192)
        // (round 3, after the test-free 16-byte survivor rounds made the sweeps cheaper relative to
        // the walk: axis-aligned packets prefer 32768 -- 1024^2 frame 2.82 -> 2.78 ms, its 1/2, 1/4,
        // 1/8 shards 1.53 -> 1.46, 0.84 -> 0.80, 0.51 -> 0.47 ms (their split waves each repeat the
        // walk); 65536 the same, 131072 worse; clustered scenes +3 %.  Pencil / general packets stay
        // at 8192: config 2 1.47 / 1.50 / 1.50 ms at 8192 / 16384 / 32768, config 3 0.87 / 0.83 / 0.86.)
        const int auto_treelet = (MODE == MODE_TRI) ? 512 : 8192, auto_treelet_axis = 32768;
#ifdef GRACE_PACKET_STATS
        a.treelet = ts.treelet < 0 ? auto_treelet : ts.treelet;
        a.treelet_axis = ts.treelet < 0 ? auto_treelet_axis : ts.treelet;
#else
        a.treelet = (MODE == MODE_STATS) ? 0 : (ts.treelet < 0 ? auto_treelet : ts.treelet);
        a.treelet_axis = (MODE == MODE_STATS) ? 0 : (ts.treelet < 0 ? auto_treelet_axis : ts.treelet);
#endif
        if (reorder) {
            // ext: 12 extents + [12] the device-side split + [13] the lattice flag (per call)
            uint32_t* ext = Workspace::take<uint32_t>(16);
            uint32_t* keys = Workspace::take<uint32_t>(n_rays);
            uint32_t* perm = Workspace::take<uint32_t>(n_rays);
            constexpr bool lat_mode = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE || MODE == MODE_HITS);
            uint32_t* lat_flag = lat_mode ? ext + 13 : nullptr;
            int* split_dev = dev_split ? reinterpret_cast<int*>(ext + 12) : nullptr;
            const int split_flags = (MODE == MODE_CUMULATIVE) ? SPLIT_WIDE_BUDGET : 0;
            if (rays_cached) {
                RayOrder& ro = ts.rays;
                if (rays_use != USE_TRUST) {
                    // (gated by the cache's stale flag; the order lands in the cache's own buffers)
                    GRACE_TRY(ray_order(a.rays, n_rays, ro.ext, keys, ro.perm, nullptr, nullptr, 0, 0, nullptr,
                                        stream, &ro.ctl->stale));
                    ro.valid = true;
                }
                // cached order: only this call's device-side choices remain
                if (lat_flag || split_dev) {
                    GRACE_TRY(launch_choose_variants(ro.ext, int(n_rays), a.C + 2 * n_clusters, lat_flag,
                                                     n_packets, split | split_flags, split_dev, stream));
                }
                a.perm = ro.perm;
            } else {
                GRACE_TRY(ray_order(a.rays, n_rays, ext, keys, perm, a.C + 2 * n_clusters, lat_flag, n_packets,
                                    split | split_flags, split_dev, stream));
                a.perm = perm;
            }
            if (lat_mode) a.lat_dev = reinterpret_cast<const int*>(ext + 13);
            if (dev_split) a.split_dev = reinterpret_cast<const int*>(ext + 12);
        }
    }
    a.n_rays = int(n_rays);
    a.n_nodes = int(n_nodes);
    a.status = ts.status;
    a.n_prims = int(n_spheres);
    a.group_shift = group_shift(n_spheres);
    a.chunk_shift = hit_chunk_shift;
    a.n_chunks = hit_chunks;
    a.chunk_counts = nullptr;
    a.chunk_off = chunk_off;
    a.wave_map = wave_map;
    a.n_wave_map = n_wave_map;
    if (split > 1 && (MODE == MODE_COUNT || MODE == MODE_COUNT_D4))
        GRACE_TRY_HIP(hipMemsetAsync(a.out_counts, 0, n_rays * sizeof(int), stream));
    if (keep_chunks && split > 1) {
        a.chunk_counts = ts.hits.chunk_counts;
        GRACE_TRY_HIP(hipMemsetAsync(ts.hits.chunk_counts, 0, n_rays * size_t(hit_chunks) * 4, stream));
        ts.hits.valid = true;
        ts.hits.rays = a.rays; ts.hits.n_rays = n_rays;
        ts.hits.prims = a.spheres; ts.hits.n_prims = n_spheres; ts.hits.n_chunks = hit_chunks;
    }
    if (ts.timing) {
        if (!ts.ev0) {
            GRACE_TRY_HIP(hipEventCreate(&ts.ev0));
            GRACE_TRY_HIP(hipEventCreate(&ts.ev1));
        }
        GRACE_TRY_HIP(hipEventRecord(ts.ev0, stream));
    }
    const int grid = ceil_div(size_t(n_packets) * split, TRACE_BLOCK / 64);
    ts.last_lat_dev = a.lat_dev; ts.last_lat_stream = stream;
    // Both variants of a kernel with a lattice instantiation (the device flag lets one run).
    auto both = [&](auto mode_tag, auto split_tag, auto alt_tag, const TraceArgs& args) {
        constexpr int M = decltype(mode_tag)::value;
        constexpr bool S = decltype(split_tag)::value, A = decltype(alt_tag)::value;
        trace_kernel<M, S, A, false><<<grid, TRACE_BLOCK, 0, stream>>>(args);
        if (args.lat_dev) trace_kernel<M, S, A, true><<<grid, TRACE_BLOCK, 0, stream>>>(args);
    };
    using T = std::true_type; using F = std::false_type;
    using M_ = std::integral_constant<int, MODE>;
    // A batch of >= 16384 packets runs one wave per packet -- unless the device flag says the scene
    // holds spheres smaller than the ray spacing (clustered SPH data: dense cores).  Such scenes
    // have packets dozens of times heavier than the median (10^7 particles, 90 % of them in 50
    // clumps: with the lattice cull the heaviest of 16384 waves still lived 14x the mean and set
    // the kernel time), so the lattice instantiation of these batches is the class-split kernel
    // with four waves per packet: the heaviest packets' work is spread over four SIMDs (measured
    // on two clustered scenes: K = 2 / 4 / 8 -> 3.62 / 3.47 / 4.24 ms and 4.24 / 3.42 / 3.71 ms;
    // one wave: 4.66 and 6.78 ms).  Same class sums, same bits.
    const bool lat_split = (MODE == MODE_COUNT || MODE == MODE_CUMULATIVE) && split == 1 && a.lat_dev && width == 64
        && ts.lat_split > 0 && ts.split <= 0;   // (an explicit grace_trace_set_packet_split is obeyed)
    auto one_or_split = [&](auto alt_tag) -> grace_status {
        constexpr bool A = decltype(alt_tag)::value;
        constexpr int M = (MODE == MODE_COUNT) ? MODE_COUNT : MODE_CUMULATIVE;
        if (MODE == MODE_COUNT) GRACE_TRY_HIP(hipMemsetAsync(a.out_counts, 0, n_rays * sizeof(int), stream));
        trace_kernel<M, false, A, false><<<grid, TRACE_BLOCK, 0, stream>>>(a);
        GRACE_CHECK_LAUNCH();
        TraceArgs a8 = a;
        a8.split = ts.lat_split; a8.split_dev = nullptr;
        trace_kernel<M, true, A, true><<<ceil_div(size_t(n_packets) * ts.lat_split, TRACE_BLOCK / 64), TRACE_BLOCK, 0, stream>>>(a8);
        GRACE_CHECK_LAUNCH();
        if (MODE == MODE_CUMULATIVE) {
            combine_classes_kernel<float><<<ceil_div(n_rays, 256), 256, 0, stream>>>(a.partial, int(n_rays), ts.lat_split,
                                                                              nullptr, a.out_sums, a.lat_dev);
            GRACE_CHECK_LAUNCH();
        }
        return GRACE_OK;
    };
    if constexpr (MODE == MODE_CUMULATIVE) {
        if (ts.exact_integrals) {
            if (split > 1) both(M_(), T(), F(), a);
            else if (lat_split) GRACE_TRY(one_or_split(F()));
            else both(M_(), F(), F(), a);
        } else {
            if (split > 1) both(M_(), T(), T(), a);
            else if (lat_split) GRACE_TRY(one_or_split(T()));
            else both(M_(), F(), T(), a);
        }
    } else if constexpr (MODE == MODE_HITS) {
        if (hits_split) {
            // 1. hits per (ray, chunk): the counting walk, split by summation class -- unless the
            //    hit-count call made for this trace_sph has kept them (grace_trace_hitcounts_keep_f4)
            const int* counts = ts.hits.chunk_counts;
            if (!reuse_chunks) {
                TraceArgs c = a;
                c.chunk_counts = chunk_counts;
                c.out_counts = scratch_counts;
                GRACE_TRY_HIP(hipMemsetAsync(chunk_counts, 0, n_rays * size_t(hit_chunks) * 4, stream));
                GRACE_TRY_HIP(hipMemsetAsync(scratch_counts, 0, n_rays * 4, stream));
                both(std::integral_constant<int, MODE_COUNT>(), T(), F(), c);
                GRACE_CHECK_LAUNCH();
                counts = chunk_counts;
            }
            // 2. output offsets per (ray, chunk); the launched waves dealt to the packets by hit
            //    totals; each packet's chunks cut into its waves' ranges
            hits_offsets_kernel<<<ceil_div(n_rays, 4), 256, 0, stream>>>(counts, a.offsets, int(n_rays),
                                                                         hit_chunks, chunk_off);
            GRACE_CHECK_LAUNCH();
            hits_plan_kernel<<<n_packets, MAX_HIT_CHUNKS, 0, stream>>>(counts, a.perm, int(n_rays),
                                                                       hit_chunks, pk_prefix, pk_total);
            GRACE_CHECK_LAUNCH();
            hits_assign_kernel<<<1, 1024, 0, stream>>>(pk_total, n_packets, n_packets * split, hit_chunks,
                                                       pk_first, pk_parts, n_wave_map,
                                                       ts.hits_stage_split ? 200000ull : 0ull);
            GRACE_CHECK_LAUNCH();
            hits_bounds_kernel<<<n_packets, 64, 0, stream>>>(pk_prefix, pk_total, pk_first, pk_parts,
                                                             hit_chunks, wave_map);
            GRACE_CHECK_LAUNCH();
            // 3. the per-hit walk, wave w owning wave_map[w]'s range of chunks
            //    Heavy packets (output-bandwidth-bound: 10^5 isotropic rays through 10^6 large spheres,
            //    410 k hits per packet: 18.0 -> 11.0 ms) stage their hits in LDS and store them eight
            //    per ray at a time; light ones (61 M hits over 768 packets: 3.6 ms direct, 4.5 staged)
            //    store directly.  The hit total is known on the device only: BOTH variants are
            //    launched and the plan's flag (hits_assign_kernel) lets one of them run -- no read-back,
            //    no host synchronisation inside the call.
            TraceArgs staged = a, direct = a;
            staged.stage_dev = direct.stage_dev = n_wave_map + 4;
            staged.stage_want = 1;
            direct.stage_want = 0;
            both(M_(), T(), T(), staged);
            GRACE_CHECK_LAUNCH();
            both(M_(), T(), F(), direct);
        } else if (n_packets >= 4096) {
            both(M_(), F(), T(), a);
        } else {
            both(M_(), F(), F(), a);
        }
    } else if constexpr (MODE == MODE_COUNT) {
        if (split > 1) both(M_(), T(), F(), a);
        else if (lat_split) GRACE_TRY(one_or_split(F()));
        else both(M_(), F(), F(), a);
    } else if constexpr (MODE == MODE_COUNT_D4 || MODE == MODE_CUM_D4) {
        if (split > 1) trace_kernel<MODE, true><<<grid, TRACE_BLOCK, 0, stream>>>(a);
        else trace_kernel<MODE, false><<<grid, TRACE_BLOCK, 0, stream>>>(a);
    } else {
        trace_kernel<MODE, false><<<grid, TRACE_BLOCK, 0, stream>>>(a);
    }
    GRACE_CHECK_LAUNCH();
    GRACE_TRY(stamps_report(MODE));
    if (MODE == MODE_CUM_D4 && split > 1) {
        combine_classes_kernel<double><<<ceil_div(n_rays, 256), 256, 0, stream>>>(a.partial_d, int(n_rays), split,
                                                                                   nullptr, a.out_sums_d);
        GRACE_CHECK_LAUNCH();
    }
    if (MODE == MODE_CUMULATIVE && split > 1) {
        combine_classes_kernel<float><<<ceil_div(n_rays, 256), 256, 0, stream>>>(a.partial, int(n_rays),
                                                                                 split, a.split_dev, a.out_sums);
        GRACE_CHECK_LAUNCH();
    }
    if (ts.timing) {
        GRACE_TRY_HIP(hipEventRecord(ts.ev1, stream));
        ts.ev_valid = true;
    }
    return GRACE_OK;
}

} // namespace

namespace grace_hip {
// Called by this library's entry points that WRITE caller arrays (sort payloads, tree builds):
// a prepared scene over that array is stale from here on.
grace_status rays_invalidate_if_written(const void* d_written)
{
    Context* c = nullptr;
    GRACE_TRY(current_context(&c));
    if (!c->trace) return GRACE_OK;
    TraceState& ts = *c->trace;
    if (ts.cache_validation) return GRACE_OK;     // validated before every use: nothing to drop eagerly
    if (ts.rays.valid && d_written && d_written == static_cast<const void*>(ts.rays.key.rays)) return rays_release(ts);
    return GRACE_OK;
}

grace_status scene_invalidate_if_written(const void* d_written)
{
    Context* c = nullptr;
    GRACE_TRY(current_context(&c));
    if (!c->trace) return GRACE_OK;
    TraceState& ts = *c->trace;
    if (ts.cache_validation) return GRACE_OK;
    if (ts.scene.valid && d_written
        && (d_written == ts.scene.key.prims || d_written == ts.scene.key.nodes || d_written == ts.scene.key.leaves))
        return scene_release(ts);
    return GRACE_OK;
}
} // namespace grace_hip

extern "C" {

grace_status grace_trace_hitcounts_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                      size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                      const int* d_leaves, const int* d_root,
                                      int* d_hit_counts, grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_hit_counts, "trace_hitcounts: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.out_counts = d_hit_counts;
    return launch_trace<MODE_COUNT>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_hitcounts_keep_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                           size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                           const int* d_leaves, const int* d_root,
                                           int* d_hit_counts, grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;
    GRACE_REQUIRE(d_hit_counts, "trace_hitcounts: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.out_counts = d_hit_counts;
    a.keep_chunks = true;
    return launch_trace<MODE_COUNT>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_cumulative_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                       size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                       const int* d_leaves, const int* d_root,
                                       float* d_cumulated, grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_cumulated, "trace_cumulative: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.out_sums = d_cumulated;
    return launch_trace<MODE_CUMULATIVE>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_hits_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                 size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                 const int* d_leaves, const int* d_root,
                                 const int* d_ray_offsets, int* d_hit_indices,
                                 float* d_hit_integrals, float* d_hit_distances,
                                 grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_ray_offsets && d_hit_indices && d_hit_integrals && d_hit_distances,
                  "trace_hits: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.offsets = d_ray_offsets;
    a.hit_idx = d_hit_indices;
    a.hit_integral = d_hit_integrals;
    a.hit_dist = d_hit_distances;
    return launch_trace<MODE_HITS>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_closest_tri(const void* d_rays, size_t n_rays, const float* d_tris,
                                     size_t n_tris, const int* d_nodes, size_t n_nodes,
                                     const int* d_leaves, const int* d_root, int* d_closest,
                                     grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_closest, "trace_closest_tri: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_tris); // 9 floats per triangle
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.out_counts = d_closest;
    return launch_trace<MODE_TRI>(a, n_rays, n_tris, n_nodes, as_stream(stream));
}

// ---- double4 spheres (Real4 = double4, Real = double) ----------------------------------------
static TraceArgs d4_args(const void* d_rays, const double* d_spheres, const int* d_nodes,
                         const int* d_leaves, const int* d_root)
{
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);   // (non-null check only)
    a.spheres_d = d_spheres;
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    return a;
}

grace_status grace_trace_hitcounts_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                      size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                      const int* d_leaves, const int* d_root, int* d_hit_counts,
                                      grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_hit_counts, "trace_hitcounts (double4): null output");
    TraceArgs a = d4_args(d_rays, d_spheres, d_nodes, d_leaves, d_root);
    a.out_counts = d_hit_counts;
    return launch_trace<MODE_COUNT_D4>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_cumulative_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                       size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                       const int* d_leaves, const int* d_root, double* d_sums,
                                       grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_sums, "trace_cumulative (double4): null output");
    TraceArgs a = d4_args(d_rays, d_spheres, d_nodes, d_leaves, d_root);
    a.out_sums_d = d_sums;
    return launch_trace<MODE_CUM_D4>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_hits_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                 size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                 const int* d_leaves, const int* d_root, const int* d_ray_offsets,
                                 int* d_hit_indices, double* d_hit_integrals,
                                 double* d_hit_distances, grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_ray_offsets && d_hit_indices && d_hit_integrals && d_hit_distances,
                  "trace_hits (double4): null output");
    TraceArgs a = d4_args(d_rays, d_spheres, d_nodes, d_leaves, d_root);
    a.offsets = d_ray_offsets;
    a.hit_idx = d_hit_indices;
    a.hit_integral_d = d_hit_integrals;
    a.hit_dist_d = d_hit_distances;
    return launch_trace<MODE_HITS_D4>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_trace_status_d4(grace_stream stream) { return grace_trace_status(stream); }

grace_status grace_trace_stats_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                  size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                  const int* d_leaves, const int* d_root,
                                  uint32_t* d_stats4, grace_stream stream)
{
    if (n_rays == 0) return GRACE_OK;   // an empty shard of a sharded batch: nothing to trace
    GRACE_REQUIRE(d_stats4, "trace_stats: null output");
    TraceArgs a = {};
    a.rays = static_cast<const float*>(d_rays);
    a.spheres = reinterpret_cast<const float4*>(d_spheres);
    a.nodes = reinterpret_cast<const float4*>(d_nodes);
    a.leaves = reinterpret_cast<const int4*>(d_leaves);
    a.root = d_root;
    a.stats = d_stats4;
    return launch_trace<MODE_STATS>(a, n_rays, n_spheres, n_nodes, as_stream(stream));
}

grace_status grace_hit_integrals_f32(const float* d_b2, const float* d_h, size_t n, float* d_out,
                                     grace_stream stream)
{
    GRACE_REQUIRE(n == 0 || (d_b2 && d_h && d_out), "hit_integrals: null pointer");
    if (n == 0) return GRACE_OK;
    hit_integrals_kernel<<<stream_grid(n, 256), 256, 0, as_stream(stream)>>>(d_b2, d_h, n, d_out);
    GRACE_CHECK_LAUNCH();
    return GRACE_OK;
}

// The calling thread's TraceState as `ts` (entry-point boilerplate).
#define GRACE_TRACE_STATE()                                                                      \
    TraceState* ts_ptr_ = nullptr;                                                               \
    GRACE_TRY(trace_state(&ts_ptr_));                                                            \
    TraceState& ts = *ts_ptr_

grace_status grace_trace_prepare_f4(const float* d_spheres, size_t n_spheres, const int* d_nodes,
                                    size_t n_nodes, const int* d_leaves, grace_stream stream)
{
    GRACE_TRACE_STATE();
    return scene_prepare(ts, false, d_spheres, n_spheres, d_nodes, n_nodes, d_leaves, as_stream(stream));
}

grace_status grace_trace_prepare_tri(const float* d_tris, size_t n_tris, const int* d_nodes,
                                     size_t n_nodes, const int* d_leaves, grace_stream stream)
{
    GRACE_TRACE_STATE();
    return scene_prepare(ts, true, d_tris, n_tris, d_nodes, n_nodes, d_leaves, as_stream(stream));
}

grace_status grace_trace_prepare_rays(const void* d_rays, size_t n_rays, grace_stream stream)
{
    GRACE_TRACE_STATE();
    return rays_prepare(ts, static_cast<const float*>(d_rays), n_rays, as_stream(stream));
}

grace_status grace_trace_release_rays(void)
{
    GRACE_TRACE_STATE();
    return rays_release(ts);
}

grace_status grace_trace_release(void)
{
    GRACE_TRACE_STATE();
    if (ts.hits.chunk_counts) {
        GRACE_TRY_HIP(hipDeviceSynchronize());
        GRACE_TRY_HIP(hipFree(ts.hits.chunk_counts));
    }
    ts.hits = HitsCache();
    return scene_release(ts);
}

grace_status grace_trace_enable_timing(int enabled)
{
    GRACE_TRACE_STATE();
    ts.timing = enabled != 0;
    ts.ev_valid = false;
    return GRACE_OK;
}

grace_status grace_trace_last_kernel_ms(float* h_ms)
{
    GRACE_REQUIRE(h_ms, "null output");
    GRACE_TRACE_STATE();
    GRACE_REQUIRE(ts.timing && ts.ev_valid, "no timed traversal launch recorded");
    GRACE_TRY_HIP(hipEventSynchronize(ts.ev1));
    GRACE_TRY_HIP(hipEventElapsedTime(h_ms, ts.ev0, ts.ev1));
    return GRACE_OK;
}

grace_status grace_trace_last_lattice(int* h_lattice)
{
    GRACE_REQUIRE(h_lattice, "null output");
    *h_lattice = 0;
    GRACE_TRACE_STATE();
    if (ts.last_lat_dev) {
        GRACE_TRY_HIP(hipStreamSynchronize(ts.last_lat_stream));
        GRACE_TRY_HIP(hipMemcpy(h_lattice, ts.last_lat_dev, sizeof(int), hipMemcpyDeviceToHost));
    }
    return GRACE_OK;
}

grace_status grace_trace_set_packet_split(int waves_per_packet)
{
    GRACE_REQUIRE(waves_per_packet == -1 || waves_per_packet == 1 || waves_per_packet == 2
                      || waves_per_packet == 4 || waves_per_packet == 8,
                  "packet split must be 1, 2, 4, 8 or -1 (automatic)");
    GRACE_TRACE_STATE();
    ts.split = waves_per_packet;
    return GRACE_OK;
}

grace_status grace_trace_set_packet_width(int rays_per_packet)
{
    GRACE_REQUIRE(rays_per_packet == -1 || rays_per_packet == 16 || rays_per_packet == 32
                      || rays_per_packet == 64,
                  "packet width must be 16, 32, 64 or -1 (automatic)");
    GRACE_TRACE_STATE();
    ts.width = rays_per_packet;
    return GRACE_OK;
}

grace_status grace_trace_set_exact_integrals(int enabled)
{
    GRACE_TRACE_STATE();
    ts.exact_integrals = enabled != 0;
    return GRACE_OK;
}

grace_status grace_trace_set_treelet_size(int max_primitives)
{
    GRACE_REQUIRE(max_primitives >= -1, "treelet size must be >= 0 (or -1 for automatic)");
    GRACE_TRACE_STATE();
    ts.treelet = max_primitives;
    return GRACE_OK;
}

grace_status grace_trace_set_ray_reorder(int enabled)
{
    GRACE_TRACE_STATE();
    ts.ray_reorder = enabled != 0;
    return GRACE_OK;
}

grace_status grace_trace_set_cache_validation(int enabled)
{
    GRACE_TRACE_STATE();
    ts.cache_validation = enabled != 0;
    return GRACE_OK;
}

grace_status grace_trace_set_cache_auto(int enabled)
{
    GRACE_TRACE_STATE();
    ts.cache_auto = enabled != 0;
    if (!ts.cache_auto) {     // what was cached automatically goes; pinned (prepared) records stay
        if (ts.scene.valid && !ts.scene.pinned) GRACE_TRY(scene_release(ts));
        if (ts.rays.valid && !ts.rays.pinned) GRACE_TRY(rays_release(ts));
        ts.scene.seen = SceneKey();
        ts.rays.seen = RayKey();
    }
    return GRACE_OK;
}

grace_status grace_trace_set_lattice_split(int waves_per_packet)
{
    GRACE_REQUIRE(waves_per_packet == 0 || waves_per_packet == 2 || waves_per_packet == 4
                      || waves_per_packet == 8,
                  "lattice split must be 0 (one wave per packet), 2, 4 or 8");
    GRACE_TRACE_STATE();
    ts.lat_split = waves_per_packet;
    return GRACE_OK;
}

grace_status grace_trace_set_hits_staging(int enabled)
{
    GRACE_TRACE_STATE();
    ts.hits_stage_split = enabled != 0;
    return GRACE_OK;
}

grace_status grace_trace_status(grace_stream stream)
{
    GRACE_TRACE_STATE();
    if (!ts.status) return GRACE_OK;
    int h = 0;
    GRACE_TRY_HIP(hipMemcpyAsync(&h, ts.status, sizeof(int), hipMemcpyDeviceToHost,
                                 as_stream(stream)));
    GRACE_TRY_HIP(hipStreamSynchronize(as_stream(stream)));
    if (h != 0) {
        GRACE_TRY_HIP(hipMemsetAsync(ts.status, 0, sizeof(int), as_stream(stream)));
        return set_error(GRACE_STACK_OVERFLOW, __FILE__, __LINE__,
                         "trace: packet stack (128 entries) exhausted");
    }
    return GRACE_OK;
}

} // extern "C"
