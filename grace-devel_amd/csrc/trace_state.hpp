// Shared declarations of the traversal's translation units -- trace.hip (the kernel's
// instantiations, launch logic, entry points), trace_prepass.hip (scene-constant records) and
// trace_coherence.hip (ray coherence order): modes, the kernel's argument block, and the state a
// Context keeps between trace calls (status word, knobs, timing events, cached scene records and
// ray order).
#pragma once

#include "common.hpp"

namespace grace_hip {

constexpr int TRACE_BLOCK = 256;
constexpr int N_TABLE = 51;
// Flag on the `launched` argument of the device-side split choice: column densities let a second
// wave per packet work up to 32768 waves (hit counts: 16384).
constexpr int SPLIT_WIDE_BUDGET = 0x100;

// Cluster records C of a scene of n primitives (float4 units): {lo, hi} per cluster of 64
// consecutive primitives, one tail record (the scene's smallest r^2 and the node_prims flag), then
// {lo, hi} per GROUP of 2^group_shift(n) consecutive primitives -- the union of its clusters'
// boxes; at most 4096 groups, of 4096 primitives each up to 16.7 M primitives.  Axis-aligned
// packets of the hit-count / column-density traces test the groups' boxes, 64 per lane-parallel
// pass, instead of walking the tree down to subtrees of that size (trace_kernel.hpp).
inline int group_shift(size_t n_prims)
{
    int s = 12;
    while (((n_prims + (size_t(1) << s) - 1) >> s) > 4096) ++s;
    return s;
}
inline size_t group_count(size_t n_prims)
{
    const int s = group_shift(n_prims);
    return (n_prims + (size_t(1) << s) - 1) >> s;
}
inline size_t cluster_record_count(size_t n_prims)
{
    return 2 * ((n_prims + 63) / 64) + 1 + 2 * group_count(n_prims);
}
constexpr int MAX_HIT_CHUNKS = 256; // chunk ranges of the split per-hit trace
constexpr int SUM_CLASSES = 8;   // summation classes (leaves of the pairwise sum tree)
constexpr int GRANULE_SHIFT = 10; // 1024 consecutive primitives share a class


enum { MODE_COUNT = 0, MODE_CUMULATIVE = 1, MODE_HITS = 2, MODE_STATS = 3, MODE_TRI = 4,
       // Real4 = double4, Real = double (trace_sph.cuh:57-241 instantiated in double): the walk and
       // every cull run on float records that CONTAIN the double spheres; each survivor is then
       // tested and integrated in double against the caller's double4 record.
       MODE_COUNT_D4 = 5, MODE_CUM_D4 = 6, MODE_HITS_D4 = 7 };

struct TraceArgs {
    const float* rays;      // 7 floats per ray
    const uint32_t* perm;   // packet slot -> ray index (coherence order), or null
    int n_rays;
    const float4* spheres;
    const int2* node_prims; // pre-pass: per node {first primitive, primitive count}
    int treelet;            // nodes with <= treelet primitives are swept as one leaf (0: off)
    int treelet_axis;       // the same for axis-aligned packets (whose cluster test is much sharper)
    const float4* A;        // pre-pass: {x, y, z, h*h}, padded by 4 entries
    const float2* B;        // pre-pass: {1/h, (1/h)^2}, padded by 4 entries
    int group_shift;        // primitives per group box = 2^group_shift (see cluster_record_count)
    const float4* C;        // pre-pass: per CLUSTER (64 consecutive primitives) {lo.xyz, -}, {hi.xyz, -}:
                            // the box of the member spheres, slightly inflated (cluster_boxes_kernel)
    const double* T64;      // MODE_TRI pre-pass: {v, e1, e2} widened to fp64, 9 per triangle
    const double* spheres_d; // *_D4 modes: the caller's double4 spheres
    double* out_sums_d;      // MODE_CUM_D4
    double* hit_integral_d;  // MODE_HITS_D4
    double* hit_dist_d;
    int split;              // waves per packet (1, 2, 4, 8); each owns SUM_CLASSES / split classes
    int n_prims;
    float* partial;         // split > 1, cumulative: [n_rays][split] subtree sums
    double* partial_d;      // ... of the double4 trace
    // Class split only: the number of waves per packet that actually work (a power of two <=
    // split, chosen on the device from the batch's coherence); waves beyond it exit at once.
    const int* split_dev;
    const int* lat_dev;     // which of the LAT = false / true instantiations runs (null: false)
    // Split per-hit walk: which of the direct (0) / LDS-staged (1) instantiations runs -- both are
    // launched, the device-side plan (hits_assign_kernel) has set *stage_dev (null: no gate).
    const int* stage_dev;
    int stage_want;
    // Split per-hit trace (small batches): primitives are cut into n_chunks ranges of
    // 2^chunk_shift consecutive indices.  The counting pass fills chunk_counts[ray][chunk];
    // the per-hit pass lets wave w own chunks [wave_map[w].y, wave_map[w].z) of packet
    // wave_map[w].x -- heavier packets get more waves (hits_assign_kernel) -- and writes a ray's
    // hits of a chunk from chunk_off[ray][chunk] on.
    int* chunk_counts;
    const int* chunk_off;
    const int4* wave_map;
    const int* n_wave_map;
    bool keep_chunks;       // host only: a hit-count trace whose chunk counts the per-hit trace will reuse
    int chunk_shift, n_chunks;
    int width;              // rays per packet: 64, or 32 / 16 for small batches of the modes that
                            // cannot split a packet (lanes >= width re-trace the packet's last ray)
    const float4* nodes;    // 4 x float4 per node
    int n_nodes;
    const int4* leaves;
    const int* root;
    int* out_counts;        // MODE_COUNT
    float* out_sums;        // MODE_CUMULATIVE
    const int* offsets;     // MODE_HITS
    int* hit_idx;
    float* hit_integral;
    float* hit_dist;
    uint32_t* stats;        // MODE_STATS, 4 per ray
    int* status;            // set to GRACE_STACK_OVERFLOW on stack exhaustion
};


// ---- cached records and their validation (trace_cache.hip) -----------------------------------
// A trace call is stateless for its caller -- trace(rays, spheres, tree) like the reference's --
// but most callers trace the same scene, or the same rays, again and again (frames of a camera
// path over one snapshot; one ray batch over an evolving simulation).  The records a call derives
// from its inputs alone are therefore kept per context: when the arrays of a call are those of
// the previous one (same pointers and sizes), the call after that finds their records cached.
// A cached record is NEVER trusted on pointer equality: every use first re-reads the inputs
// through a 128-bit signature kernel (one streaming pass, HBM-bound, ~1/5 of the cost of
// re-deriving the records) and compares it ON THE DEVICE with the signature the records were
// derived from; if they differ the (flag-gated) pre-pass kernels recompute the records in place.
// No host round trip either way.  grace_trace_set_cache_validation(0) lets a caller who promises
// not to modify cached arrays skip the signature pass.
struct CacheCtl {                 // device memory, one per cache
    unsigned long long sig[2];    // signature of the inputs the cached records were derived from
    uint32_t stale;               // set by the check kernel of the current call: recompute
    uint32_t pad;
};

struct SceneKey {
    int kind = -1;                // 0 float4 spheres, 1 triangles
    const void* prims = nullptr; const void* nodes = nullptr; const void* leaves = nullptr;
    size_t n_prims = 0, n_nodes = 0;
    bool operator==(const SceneKey& o) const
    {
        return kind == o.kind && prims == o.prims && nodes == o.nodes && leaves == o.leaves
            && n_prims == o.n_prims && n_nodes == o.n_nodes;
    }
};

// Scene-constant pre-pass data (trace_prepass.hip): A, B, the nodes' primitive spans and the
// cluster boxes depend on the primitives and the tree only.
struct Scene {
    bool valid = false;           // the buffers hold the records of `key`
    SceneKey key;
    SceneKey seen;                // the previous call's scene (a repeat is what gets cached)
    float4* A = nullptr; float2* B1 = nullptr; float2* B50 = nullptr; double* T64 = nullptr;
    int2* node_prims = nullptr; float4* C = nullptr;
    CacheCtl* ctl = nullptr;
    bool pinned = false;          // filled by grace_trace_prepare_*: kept until released / replaced
};

struct RayKey {
    const float* rays = nullptr;
    size_t n = 0;
    bool operator==(const RayKey& o) const { return rays == o.rays && n == o.n; }
};

// Ray coherence order (trace_coherence.hip).
struct RayOrder {
    bool valid = false;
    RayKey key, seen;
    uint32_t* perm = nullptr;     // n
    uint32_t* ext = nullptr;      // 16 words: 12 extents (order-preserving uints: minima then maxima
                                  // of d, o), [14] not-a-grid flag, [15] longest ray
    CacheCtl* ctl = nullptr;
    bool pinned = false;
};

// trace_sph walks twice -- hit counts (for the offsets), then the per-hit pass -- and the split
// per-hit pass of small batches needs hits per (ray, chunk), a third walk.  The hit-count call
// made on behalf of trace_sph (grace_trace_hitcounts_keep_f4) records them into this buffer of
// its own (the workspace is reset by the scan in between); the per-hit call that follows on the
// same rays and spheres consumes them.
struct HitsCache {
    int* chunk_counts = nullptr;
    size_t capacity = 0;      // ints
    bool valid = false;
    const void* rays = nullptr; const void* prims = nullptr;
    size_t n_rays = 0, n_prims = 0;
    int n_chunks = 0;
};

struct TraceState {
    int* status = nullptr;                 // one device int, allocated on first use
    // Measurement hook (grace_trace_last_lattice): the device flag of the last trace launch (lives
    // in the call's workspace frame: valid until the next library call on the context).
    const int* last_lat_dev = nullptr;
    hipStream_t last_lat_stream = nullptr;
    bool timing = false;                   // record HIP events around the traversal kernel itself
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool ev_valid = false;
    // knobs (grace_trace_set_*)
    bool ray_reorder = true;
    int treelet = -1;                      // -1: chosen per call
    int split = -1;                        // waves per packet; -1: automatic
    int lat_split = 4;                     // waves per packet of big batches in scenes with sub-spacing spheres (0: one)
    int width = -1;                        // rays per packet of the per-hit / triangle traces; -1: automatic
    bool exact_integrals = false;          // column-density trace: bit-reproducible per-hit arithmetic
    bool hits_stage_split = true;          // split per-hit walk: stage heavy packets' hits in LDS
    bool cache_validation = true;          // validate cached records by signature before every use
    bool cache_auto = true;                // cache the records of a scene / ray batch seen twice in a row
    Scene scene;
    RayOrder rays;
    HitsCache hits;
};

// The TraceState of the calling thread's context (created on first use).
grace_status trace_state(TraceState** out);

// trace_cache.hip: signatures of the inputs behind cached records.
// One launch reads up to four arrays -- the scene's primitives, nodes and leaves (group 0) and the
// rays (group 1); a null array is skipped -- and leaves per-workgroup partial sums in `partial`
// (sig_partial_words() 64-bit words of scratch); a second, single-workgroup launch folds them,
// compares each requested group with its cache's stored signature, stores the new one and sets the
// cache's stale flag (force: set it regardless -- a first fill).  A stale ray cache also gets its
// extents re-initialised.
size_t sig_partial_words();
struct SigRequest {
    const void* prims = nullptr; size_t prims_bytes = 0;
    const void* nodes = nullptr; size_t nodes_bytes = 0;
    const void* leaves = nullptr; size_t leaves_bytes = 0;
    CacheCtl* scene_ctl = nullptr; bool scene_force = false;
    const void* rays = nullptr; size_t rays_bytes = 0;
    CacheCtl* rays_ctl = nullptr; bool rays_force = false;
    uint32_t* rays_ext = nullptr;
};
grace_status launch_signatures(const SigRequest& rq, unsigned long long* partial, hipStream_t stream);

// trace_prepass.hip
grace_status scene_release(TraceState& ts);
// Buffers of the scene cache for `key` (replaces whatever is cached).
grace_status scene_cache_alloc(TraceState& ts, const SceneKey& key);
// Fills the scene-constant arrays (any of B1 / B50 / T64 may be null).  kind: 0 float4 spheres,
// 1 triangles, 2 double4 spheres.  run_if (device, optional): every kernel returns at once if
// *run_if == 0.
grace_status scene_fill(int kind, const void* prims, size_t n_prims, const float4* nodes,
                        size_t n_nodes, const int4* leaves, float4* A, float2* B1, float2* B50,
                        double* T64, int2* node_prims, float4* C, hipStream_t stream,
                        const uint32_t* run_if = nullptr);
grace_status scene_prepare(TraceState& ts, bool tri, const void* prims, size_t n_prims,
                           const int* d_nodes, size_t n_nodes, const int* d_leaves, hipStream_t stream);

// trace_coherence.hip
grace_status rays_release(TraceState& ts);
grace_status rays_cache_alloc(TraceState& ts, const RayKey& key);
// extents -> keys -> partial sort; keys: n words of scratch (the caller's workspace frame must
// include sort_ws_bytes(n_rays, 4, 0)).  run_if (device, optional): gates every kernel; the
// extents must then have been initialised already (launch_signatures does it for a stale cache).
grace_status ray_order(const float* d_rays, size_t n_rays, uint32_t* ext, uint32_t* keys, uint32_t* perm,
                       const float4* scene_min, uint32_t* lat_flag, int n_packets, int split, int* split_dev,
                       hipStream_t stream, const uint32_t* run_if = nullptr);
grace_status rays_prepare(TraceState& ts, const float* d_rays, size_t n_rays, hipStream_t stream);
// The device-side choices of a trace launch for a batch whose order is cached.
grace_status launch_choose_variants(const uint32_t* ext12, int n, const float4* scene_min, uint32_t* lat_flag,
                                    int split_packets, int split_launched, int* split_dev, hipStream_t stream);

} // namespace grace_hip
