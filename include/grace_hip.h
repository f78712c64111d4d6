/*
 * grace_hip.h -- C ABI of libgrace_hip.so: the MI355X (gfx950) implementation of the
 * GRACE BVH-build + SPH ray-traversal hot path.
 *
 * The reference (spthm/grace-devel) has no FFI layer: its boundary is the header-template
 * API of namespace grace, compiled by nvcc into the caller.  Each entry point below is one
 * concrete instantiation of that API over raw device pointers; the reference interface it
 * replaces is cited as file:line (paths relative to the reference root).  The C++ header
 * mirror (include/grace/grace.h) forwards to these and restores the reference's error
 * behaviour (throw std::invalid_argument / print + exit).
 *
 * Conventions
 *  - All pointers named d_* are device pointers on the current HIP device; h_* are host.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *    asynchronous on that stream unless they return a value to the host (documented).
 *  - Every function returns a grace_status; grace_last_error() describes the last failure
 *    of the calling thread.
 *  - Temporaries come from a grow-only device workspace owned by the library (the
 *    reference allocates and frees thrust temporaries inside every call).  Workspace, status
 *    word, cached trace scene / ray order, timing events and tuning knobs belong to a CONTEXT.
 *    Every device has a default context, used by threads that never ask for another: with one
 *    process per GPU, or one process that sets a device current and calls, nothing needs to be
 *    done.  A context serves one call at a time; host threads may call concurrently iff each has
 *    made a context of its own current (grace_context_create / grace_context_set_current) -- one
 *    thread per GPU of a node, or several threads sharing one GPU.  Calls may use any stream;
 *    consecutive calls of a context on different streams are ordered by the library (their
 *    temporaries share memory), and a stream may be destroyed as soon as the caller has
 *    synchronised with it.
 *  - float4 / int4 / Ray arrays are passed as float* / int* / void* with the reference's
 *    memory layout: sphere = {x, y, z, h}; Ray = {dx,dy,dz,ox,oy,oz,length} (28 B,
 *    include/grace/ray.h:5-10); node = 4 x 16 B, leaf = int4 (include/grace/cuda/nodes.h:22-42).
 */
#ifndef GRACE_HIP_H
#define GRACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum grace_status {
    GRACE_OK = 0,
    GRACE_INVALID_ARGUMENT = 1, /* the reference throws std::invalid_argument          */
    GRACE_HIP_ERROR = 2,        /* the reference prints and exit()s (error.h:40-56)     */
    GRACE_OUT_OF_MEMORY = 3,
    GRACE_STACK_OVERFLOW = 4    /* traversal stack exhausted (reference: GRACE_ASSERT)  */
} grace_status;

typedef void* grace_stream;

int grace_version(void);
const char* grace_last_error(void);

/* ---- device memory helpers: let HIP-free host code (include/grace) own device vectors,
 *      the role thrust::device_vector plays in the reference ---------------------------- */
grace_status grace_device_malloc(void** d_ptr, size_t bytes);
grace_status grace_device_free(void* d_ptr);
grace_status grace_memcpy_htod(void* d_dst, const void* h_src, size_t bytes, grace_stream stream);
grace_status grace_memcpy_dtoh(void* h_dst, const void* d_src, size_t bytes, grace_stream stream);
grace_status grace_memcpy_dtod(void* d_dst, const void* d_src, size_t bytes, grace_stream stream);
grace_status grace_memset(void* d_dst, int byte, size_t bytes, grace_stream stream);
grace_status grace_stream_synchronize(grace_stream stream);
/* Pre-size / drop the workspace of the calling thread's context (optional). */
grace_status grace_workspace_reserve(size_t bytes);
grace_status grace_workspace_release(void);

/* ---- contexts: everything the library keeps between calls (see Conventions).  The reference has
 *      one set of globals per process (texture references, bintree_trace.cuh:37-38) and drives
 *      several GPUs from one process only through ncclCommInitAll-style code of the caller's
 *      (SURVEY.md section 8e); a context per (thread, device) is what makes that form work here.
 *      grace_context_create: a fresh context on the CURRENT device.  grace_context_set_current:
 *      the calling thread's context from now on (NULL: back to the current device's default
 *      context).  grace_context_destroy frees the context's device memory and events (its device
 *      is made current for the duration); default contexts are never destroyed.
 *      grace_context_get_current returns the context the next call of this thread would use. */
typedef struct grace_context_s* grace_context;
grace_status grace_context_create(grace_context* ctx);
grace_status grace_context_destroy(grace_context ctx);
grace_status grace_context_set_current(grace_context ctx);
grace_status grace_context_get_current(grace_context* ctx);

/* ---- Morton keys ------------------------------------------------------------------- */
/* AABB of sphere centroids: compute_centroids + min_vec3/max_vec3
 * (include/grace/cuda/kernels/morton.cuh:153-164), fused into one pass.  Synchronises;
 * results in host arrays h_bot[3], h_top[3]. */
grace_status grace_centroid_bounds_f4(const float* d_spheres, size_t n,
                                      float* h_bot, float* h_top, grace_stream stream);
/* min_vec4 / max_vec4 over x,y,z,w (include/grace/cuda/util/extrema.cuh, as used by
 * tests/project_gadget/project_gadget.cu:66-68).  Synchronises. */
grace_status grace_minmax_f4(const float* d_v4, size_t n, float* h_mins4, float* h_maxs4,
                             grace_stream stream);
/* The rest of include/grace/cuda/util/extrema.cuh:190-772 (min_max_x/y/z/w, min/max_vec2/3/4 over
 * any vector type): component-wise minima and maxima of n records of n_comp (1..4) leading
 * components of elem_type, stride_bytes apart.  h_mins / h_maxs: host arrays of n_comp elements
 * of the same type.  Synchronises. */
enum { GRACE_ELEM_F32 = 0, GRACE_ELEM_F64 = 1, GRACE_ELEM_I32 = 2, GRACE_ELEM_U32 = 3 };
grace_status grace_minmax_components(const void* d_data, size_t n, int elem_type, int n_comp,
                                     size_t stride_bytes, void* h_mins, void* h_maxs,
                                     grace_stream stream);
/* grace::morton_keys(prims, N, bot, top, keys, CentroidSphere) with uinteger32 keys and
 * Real3 = float3 (include/grace/cuda/kernels/morton.cuh:97-119,30-55; build_sph.cuh:27-35). */
grace_status grace_morton_keys30_f4(const float* d_spheres, size_t n, const float* h_bot,
                                    const float* h_top, uint32_t* d_keys, grace_stream stream);
/* Same with uinteger64 keys (63 bits); float3 and double3 bounds. */
grace_status grace_morton_keys63_f4(const float* d_spheres, size_t n, const float* h_bot,
                                    const float* h_top, uint64_t* d_keys, grace_stream stream);
grace_status grace_morton_keys63_f4_d3(const float* d_spheres, size_t n, const double* h_bot,
                                       const double* h_top, uint64_t* d_keys,
                                       grace_stream stream);

/* morton_keys_sph / morton_keys over other point types (build_sph.cuh:16-33, Real4 = double4;
 * kernels/gen_rays.cuh:603-604, PointType = float3/float4): n records of elems_per_point
 * floats (is_double = 0) or doubles (1), x y z first.  Co-ordinates are narrowed to float
 * before the key arithmetic (CentroidSphere, generic/functors/centroid.h:33-40 --
 * tests/morton_key_kernel/63bit_keys.cu:52-58).  bot/top: host float[3]. */
grace_status grace_morton_keys30_points(const void* d_points, size_t n, int is_double,
                                        int elems_per_point, const float* h_bot,
                                        const float* h_top, uint32_t* d_keys, grace_stream stream);
grace_status grace_morton_keys63_points(const void* d_points, size_t n, int is_double,
                                        int elems_per_point, const float* h_bot,
                                        const float* h_top, uint64_t* d_keys, grace_stream stream);

/* The remaining (point type, bounds type, key type) instantiations of grace::morton_keys
 * (kernels/morton.cuh:97-189): float4 spheres with double3 bounds and 30-bit keys; generic
 * points with double3 bounds (scale and key arithmetic in double on the float-narrowed
 * co-ordinates); and the centroid bounds of generic points (the bounds-free overloads for
 * Real4 = double4: compute_centroids + min/max of the float3 centroids, morton.cuh:139-174).
 * grace_centroid_bounds_points synchronises. */
grace_status grace_morton_keys30_f4_d3(const float* d_spheres, size_t n, const double* h_bot,
                                       const double* h_top, uint32_t* d_keys,
                                       grace_stream stream);
grace_status grace_centroid_bounds_points(const void* d_points, size_t n, int is_double,
                                          int elems_per_point, float* h_bot, float* h_top,
                                          grace_stream stream);
grace_status grace_morton_keys30_points_d3(const void* d_points, size_t n, int is_double,
                                           int elems_per_point, const double* h_bot,
                                           const double* h_top, uint32_t* d_keys,
                                           grace_stream stream);
grace_status grace_morton_keys63_points_d3(const void* d_points, size_t n, int is_double,
                                           int elems_per_point, const double* h_bot,
                                           const double* h_top, uint64_t* d_keys,
                                           grace_stream stream);

/* ---- stable radix sort: the thrust::sort_by_key(keys, values) call sites
 *      (include/grace/cuda/build_sph.cuh:46,57,70,81; kernels/gen_rays.cuh:483,520,577,615).
 *      Keys ascending, equal keys keep their input order; values (value_bytes per element,
 *      a multiple of 4: 4/16/28/32/36 are the reference's payloads) are permuted in place.
 *      d_values may be NULL (keys only).  d_perm (optional, n uint32) receives the source
 *      index of every output element.  Asynchronous on `stream`, no host round trip; inputs
 *      of 2^18 elements or more also use the context's internal side stream, forked from
 *      and joined to `stream` by events (bucket sort with a device-gated fallback, see
 *      csrc/sort.hip). ---------------------------------------------------------------- */
grace_status grace_sort_pairs_u32(uint32_t* d_keys, void* d_values, size_t n, int value_bytes,
                                  int begin_bit, int end_bit, uint32_t* d_perm,
                                  grace_stream stream);
grace_status grace_sort_pairs_u64(uint64_t* d_keys, void* d_values, size_t n, int value_bytes,
                                  int begin_bit, int end_bit, uint32_t* d_perm,
                                  grace_stream stream);
/* Large sorts try the bucket sort first and fall back to the index sort on the device when a bucket
 * overflows (clustered keys).  The library remembers per context whether the last large sort
 * overflowed -- a word of pinned host memory written by the sort's own kernel, never waited for --
 * and then goes straight to the index sort, looking again every 8th time.  0 switches the memory
 * off (every large sort tries the buckets); the choice is between two paths with identical results. */
grace_status grace_sort_set_overflow_hint(int enabled);

/* ---- deltas: grace::compute_deltas (include/grace/cuda/kernels/albvh.cuh:33-47,949-978)
 *      with DeltaEuclidean / DeltaSurfaceArea / DeltaXOR
 *      (include/grace/generic/functors/albvh.h:17-126).  d_deltas has n + 1 entries,
 *      d_deltas[i] = delta(i - 1); sentinels +inf / all-ones. --------------------------- */
grace_status grace_deltas_euclid_f4(const float* d_spheres, size_t n, float* d_deltas,
                                    grace_stream stream);
grace_status grace_deltas_area_f4(const float* d_spheres, size_t n, float* d_deltas,
                                  grace_stream stream);
grace_status grace_deltas_xor_u32(const uint32_t* d_keys, size_t n, uint32_t* d_deltas,
                                  grace_stream stream);
grace_status grace_deltas_xor_u64(const uint64_t* d_keys, size_t n, uint64_t* d_deltas,
                                  grace_stream stream);

/* ---- ALBVH: grace::build_ALBVH (include/grace/cuda/kernels/albvh.cuh:986-1072) with
 *      DeltaComp = less and AABBSphere.  d_nodes: capacity 16 * (n - 1) ints; d_leaves:
 *      capacity 4 * n ints; d_root: one device int.  *h_n_leaves receives the leaf count
 *      (the reference resizes tree.nodes/leaves from it, albvh.cuh:842-845): synchronises.
 *      GRACE_INVALID_ARGUMENT if n <= max_per_leaf (albvh.cuh:795-799). ---------------- */
grace_status grace_albvh_build_f4(const float* d_spheres, size_t n, const float* d_deltas,
                                  int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                  size_t* h_n_leaves, grace_stream stream);
/* XOR (uint32) deltas variant. */
grace_status grace_albvh_build_f4_u32(const float* d_spheres, size_t n, const uint32_t* d_deltas,
                                      int max_per_leaf, int* d_nodes, int* d_leaves,
                                      int* d_root, size_t* h_n_leaves, grace_stream stream);

/* ---- triangle primitives: the alternate-primitive instantiation of the same templates
 *      (tests/profile_trace_triangle).  Triangle = {v, e1, e2}, 9 floats, 36 B
 *      (triangle.cuh:11-25). ------------------------------------------------------------- */
/* compute_centroids + min/max with TriangleCentroid (triangle.cuh:92-102;
 * include/grace/cuda/kernels/morton.cuh:139-174).  Synchronises. */
grace_status grace_centroid_bounds_tri(const float* d_tris, size_t n, float* h_bot, float* h_top,
                                       grace_stream stream);
/* grace::morton_keys(d_tris, ..., TriangleCentroid()) with 30-bit keys (tris_tree.cuh:27). */
grace_status grace_morton_keys30_tri(const float* d_tris, size_t n, const float* h_bot,
                                     const float* h_top, uint32_t* d_keys, grace_stream stream);
/* grace::build_ALBVH(d_tree, d_tris, d_deltas, TriangleAABB()) with XOR deltas
 * (tris_tree.cuh:28-29; TriangleAABB triangle.cu:3-35). */
grace_status grace_albvh_build_tri_u32(const float* d_tris, size_t n, const uint32_t* d_deltas,
                                       int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                       size_t* h_n_leaves, grace_stream stream);

/* ---- the generic forms of grace/cuda/kernels/albvh.cuh:986-1072 in one entry: any primitive kind
 *      (float4 / double4 spheres with AABBSphere, the shipped triangles, or GRACE_PRIM_BOX: the
 *      caller's own AABBFunc already evaluated per primitive -- 6 floats {bot xyz, top xyz} each,
 *      written by the header kernel of include/grace/cuda/kernels/albvh.cuh, which runs the functor
 *      in the caller's translation unit), any delta type, and DeltaComp = thrust::less or
 *      thrust::greater (albvh.cuh:1029-1045; the comparator is only ever applied as
 *      delta_comp(delta_L, delta_R), albvh.cuh:129,194,465,607). -------------------------------- */
enum { GRACE_PRIM_SPHERE_F4 = 0, GRACE_PRIM_TRIANGLE = 1, GRACE_PRIM_SPHERE_D4 = 2, GRACE_PRIM_BOX = 3 };
enum { GRACE_DELTA_F32 = 0, GRACE_DELTA_F64 = 1, GRACE_DELTA_U32 = 2, GRACE_DELTA_U64 = 3 };
enum { GRACE_COMP_LESS = 0, GRACE_COMP_GREATER = 1 };
grace_status grace_albvh_build_ex(int prim_kind, const void* d_prims, size_t n, int delta_type,
                                  const void* d_deltas, int delta_comp, int max_per_leaf,
                                  int* d_nodes, int* d_leaves, int* d_root, size_t* h_n_leaves,
                                  grace_stream stream);

/* Measurement hook for profile_tree-style harnesses (tests/profile_tree/profile_tree.cu prints
 * one line per build phase): when enabled, HIP events are recorded on the build's stream around
 * its leaf stage (leaf heads + scan + leaf records AND their deltas: the reference's
 * build_leaves, remove_empty_leaves and copy_leaf_deltas, fused here) and its node stage
 * (leaf boxes, pyramids, nodes: build_nodes). */
grace_status grace_albvh_enable_timing(int enabled);
grace_status grace_albvh_last_phase_ms(float* h_leaves_ms, float* h_nodes_ms);
/* trace_closest_tri (tris_trace.cu:43-62): RayEntry_tri / RayIntersect_tri / OnHit_tri
 * (tris_trace.cuh:11-73) over Moeller-Trumbore with back-face culling (triangle.cuh:54-88);
 * d_closest[ray] = index of the nearest triangle hit, or -1. */
grace_status grace_trace_closest_tri(const void* d_rays, size_t n_rays, const float* d_tris,
                                     size_t n_tris, const int* d_nodes, size_t n_nodes,
                                     const int* d_leaves, const int* d_root, int* d_closest,
                                     grace_stream stream);

/* ---- traversal: grace::trace_hitcounts_sph / trace_cumulative_sph / trace_sph pass 2
 *      (include/grace/cuda/trace_sph.cuh:58-168) over trace_kernel
 *      (include/grace/cuda/kernels/bintree_trace.cuh:52-197).  n_nodes = n_leaves - 1.
 *      Any n_rays is accepted here (0: nothing to do, e.g. the empty shard of a sharded
 *      batch); the header mirror enforces the reference's n_rays % 32 == 0
 *      (bintree_trace.cuh:231-238). ----------------------------------------------------- */
grace_status grace_trace_hitcounts_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                      size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                      const int* d_leaves, const int* d_root,
                                      int* d_hit_counts, grace_stream stream);
/* grace_trace_hitcounts_f4 for a caller that goes on to the per-hit pass (trace_sph,
 * trace_with_sentinels_sph: trace_sph.cuh:121-141): same output; for small batches it also keeps
 * the hits per (ray, primitive chunk) in a buffer of the library's, which the next
 * grace_trace_hits_f4 call on the same rays and spheres consumes instead of walking the tree a
 * third time.  Any trace call in between drops them. */
grace_status grace_trace_hitcounts_keep_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                           size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                           const int* d_leaves, const int* d_root,
                                           int* d_hit_counts, grace_stream stream);
grace_status grace_trace_cumulative_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                       size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                       const int* d_leaves, const int* d_root,
                                       float* d_cumulated, grace_stream stream);
/* Per-hit outputs written from d_ray_offsets[ray] (RayEntry_from_array +
 * OnHit_sphere_individual, include/grace/cuda/functors/trace.cuh:44-60,196-235). */
grace_status grace_trace_hits_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                 size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                 const int* d_leaves, const int* d_root,
                                 const int* d_ray_offsets, int* d_hit_indices,
                                 float* d_hit_integrals, float* d_hit_distances,
                                 grace_stream stream);
/* Instrumented walk: per ray {nodes visited, leaves visited, spheres tested, hits} for that
 * ray alone (4 x uint32 per ray) -- the counts SURVEY.md section 8d's algorithmic-bytes
 * formula is built from.  Not part of the reference API. */
grace_status grace_trace_stats_f4(const void* d_rays, size_t n_rays, const float* d_spheres,
                                  size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                  const int* d_leaves, const int* d_root,
                                  uint32_t* d_stats4, grace_stream stream);

/* The per-hit kernel integral of OnHit_sphere_cumulate / OnHit_sphere_individual
 * (include/grace/cuda/functors/trace.cuh:181-186, 221-224) evaluated on arrays:
 * out[i] = lerp(50 * sqrt(b2[i]) / h[i], table) / h[i]^2, bit-for-bit the traversal's value. */
grace_status grace_hit_integrals_f32(const float* d_b2, const float* d_h, size_t n, float* d_out,
                                     grace_stream stream);

/* Packets are formed from 64 consecutive rays of a coherence order computed inside every
 * trace call (Morton code over the varying ray coordinates + stable sort); results per ray
 * do not depend on it.  0 disables it: packets then follow the caller's ray order, as in
 * the reference.  Default 1.  Not part of the reference API. */
grace_status grace_trace_set_ray_reorder(int enabled);

/* Measurement hook: when enabled, HIP events are recorded on the call's stream directly
 * around the traversal kernel of every trace call (not around its pre-passes);
 * grace_trace_last_kernel_ms waits for the last one and returns its duration. */
grace_status grace_trace_enable_timing(int enabled);
grace_status grace_trace_last_kernel_ms(float* h_ms);

/* Measurement hook: 1 if the last hit-count / cumulative / per-hit trace ran the kernel
 * instantiation with the origin-lattice cull (chosen on the device: all rays share one
 * axis-aligned direction and the scene holds spheres smaller than the mean ray cell), else 0.
 * Valid until the next library call on the device.  Results never depend on the choice. */
grace_status grace_trace_last_lattice(int* h_lattice);

/* Waves per 64-ray packet for the hit-count and cumulative traces: 1, 2, 4 or 8 (each wave
 * owns 8/K of the 8 interleaved primitive classes over which the column density is summed), or
 * -1 (default) = as many as it takes to put >= 16384 waves on the chip (>= 4096 if all rays of
 * the batch share one direction: decided on the device from the ray extents).  The results do not
 * depend on it (see csrc/trace.hip, "class-ordered sums and packet splitting"). */
grace_status grace_trace_set_packet_split(int waves_per_packet);

/* Rays per packet of the per-hit and triangle traces (which cannot split a packet among waves):
 * 64, 32 or 16, or -1 (default) = halve while the call has fewer than 4096 packets.  Results do
 * not depend on it. */
grace_status grace_trace_set_packet_width(int rays_per_packet);

/* Column-density trace (grace_trace_cumulative_f4) only.  0 (default): each hit's kernel
 * integral is evaluated with the hardware sqrt (1 ulp) and an fp32 table lerp -- within a few
 * ulp of the reference arithmetic per term, column densities within 1e-6 of the fp64 sum
 * (stated tolerance 1e-5).  1: the reference's arithmetic bit for bit (correctly rounded
 * sqrt, fp64 lerp of the fp64 table; functors/trace.cuh:181-186, interpolate.h:11-39) --
 * the result is then bit-identical to the CPU oracle's class-ordered sum, ~20 % slower.
 * The per-hit outputs (grace_trace_hits_f4, grace_hit_integrals_f32) always use the latter. */
grace_status grace_trace_set_exact_integrals(int enabled);

/* Subtrees with at most this many primitives are swept -- one test per cluster of 64 consecutive
 * primitives, then culling rounds over the surviving clusters -- instead of being descended
 * (results per ray unchanged).  0 disables; -1 (default) = 16384 for axis-aligned packets (whose
 * cluster test is a sharp box-rectangle overlap), 512 for the others. */
grace_status grace_trace_set_treelet_size(int max_primitives);

/* Measurement switches (results never depend on them).  Lattice split: waves per packet that a
 * batch of >= 16384 packets gets when the device finds spheres smaller than the ray spacing in the
 * scene (clustered SPH data): 0 = one wave per packet, 2, 4 (default) or 8.  Hits staging: 0 = the
 * split per-hit walk of small batches always stores hits directly (default 1: heavy packets stage
 * them in LDS). */
grace_status grace_trace_set_lattice_split(int waves_per_packet);
grace_status grace_trace_set_hits_staging(int enabled);

/* Cached trace records.  Every trace call derives, from the primitives and the tree alone,
 * per-sphere records ({x, y, z, h^2}, {1/h, 1/h^2}), every node's primitive span and one box per
 * cluster of 64 consecutive primitives; and from the rays alone the coherence order in which
 * packets are formed (see csrc/trace.hip).  The reference's traces are stateless
 * (trace_sph.cuh:58-241 rebuild even the 51-entry table per call) and so is every call here for its
 * caller -- but a context KEEPS the records of the scene and of the ray batch it was last given:
 * when a call names the same arrays (pointers and sizes) as the call before it, the records are
 * derived into buffers of the context's own, and later calls on those arrays reuse them.
 *
 * A cached record is never trusted on pointer equality.  Before every use, one streaming pass
 * reduces the arrays the call was given to a 128-bit signature and compares it -- on the device,
 * no host round trip -- with the signature the cached records were derived from; if the caller
 * (or an allocator that handed out the same address again) changed the contents, the records are
 * recomputed in place by the same call.  Results are therefore always those of a fresh derivation;
 * the signature pass costs about a fifth of one (0.06 ms against 0.35 ms at 10^7 particles and
 * 10^6 rays).
 *
 * grace_trace_prepare_f4 / _tri / _rays fill the cache NOW instead of at the second call, and pin
 * it: it is kept until released or replaced by another prepare.  One scene and one ray batch per
 * context.  grace_trace_set_cache_validation(0) switches the signature pass off for a caller who
 * promises not to modify cached arrays until grace_trace_release() / _release_rays() (this
 * library's own sort, build and ray-generator entry points then drop the cache themselves when they
 * write to one of its arrays); default 1.  grace_trace_set_cache_auto(0): cache on
 * grace_trace_prepare_* only; default 1.  Not part of the reference API. */
grace_status grace_trace_prepare_f4(const float* d_spheres, size_t n_spheres, const int* d_nodes,
                                    size_t n_nodes, const int* d_leaves, grace_stream stream);
grace_status grace_trace_prepare_tri(const float* d_tris, size_t n_tris, const int* d_nodes,
                                     size_t n_nodes, const int* d_leaves, grace_stream stream);
grace_status grace_trace_release(void);
grace_status grace_trace_prepare_rays(const void* d_rays, size_t n_rays, grace_stream stream);
grace_status grace_trace_release_rays(void);
grace_status grace_trace_set_cache_validation(int enabled);
grace_status grace_trace_set_cache_auto(int enabled);

/* Reads (and clears) the traversal status word: GRACE_STACK_OVERFLOW if any packet ran out
 * of its 128-entry stack since the last check (the reference only asserts this in
 * GRACE_DEBUG builds, bintree_trace.cuh:164).  Synchronises. */
grace_status grace_trace_status(grace_stream stream);

/* ---- scans ---------------------------------------------------------------------------
 * thrust::exclusive_scan of hit counts (include/grace/cuda/trace_sph.cuh:135-137).
 * In place allowed.  *h_total (optional) receives the grand total, computed in 64 bits:
 * synchronises if given.  The offsets themselves wrap modulo 2^32 like the reference's int scan;
 * a caller that turns them into array positions (trace_sph, trace_with_sentinels_sph) must
 * refuse a total above INT32_MAX -- the host-side mirrors do, with std::invalid_argument /
 * ValueError. */
grace_status grace_scan_exclusive_i32(const int* d_in, size_t n, int* d_out, long long* h_total,
                                      grace_stream stream);
/* grace::exclusive_segmented_scan (include/grace/cuda/scan.cuh:15-37): per-segment
 * exclusive prefix sums; segment s covers [offsets[s], offsets[s+1]) (last: to n); empty
 * segments allowed.  d_data and d_results may alias. */
grace_status grace_segscan_exclusive_f32(const int* d_segment_offsets, size_t n_segments,
                                         const float* d_data, size_t n, float* d_results,
                                         grace_stream stream);
grace_status grace_segscan_exclusive_f64(const int* d_segment_offsets, size_t n_segments,
                                         const double* d_data, size_t n, double* d_results,
                                         grace_stream stream);
/* Pieces of trace_with_sentinels_sph (include/grace/cuda/trace_sph.cuh:171-241):
 * offsets[i] += i (thrust::transform with a counting iterator, :205-208) and the sentinel
 * fill of the per-hit arrays (:212-214; 32-bit pattern, so int and float sentinels alike). */
grace_status grace_add_iota_i32(int* d_values, size_t n, grace_stream stream);
grace_status grace_fill_u32(void* d_values, size_t n, uint32_t bits, grace_stream stream);
/* detail::multiply_by_weights (include/grace/cuda/kernels/weights.cuh:13-27). */
grace_status grace_multiply_by_weights_f32(const float* d_unweighted, size_t n,
                                           const float* d_weights, const uint32_t* d_weight_map,
                                           float* d_weighted, grace_stream stream);
/* ... with Real = double (scan.cuh:43-58 is a template on Real). */
grace_status grace_multiply_by_weights_f64(const double* d_unweighted, size_t n,
                                           const double* d_weights, const uint32_t* d_weight_map,
                                           double* d_weighted, grace_stream stream);

/* ---- per-ray sort of hits by distance: grace::sort_by_distance
 *      (include/grace/cuda/sort.cuh:100-131): within each ray's segment distances become
 *      non-decreasing (equal distances keep their order); hit_indices and hit_data (either
 *      may be NULL) are permuted by the same map. ---------------------------------------- */
grace_status grace_sort_by_distance_f32(float* d_distances, const int* d_ray_offsets,
                                        size_t n_rays, size_t n_hits, int* d_hit_indices,
                                        float* d_hit_data, grace_stream stream);

/* sort_by_distance<double, int, double>: the double outputs of grace_trace_hits_d4. */
grace_status grace_sort_by_distance_f64(double* d_distances, const int* d_ray_offsets,
                                        size_t n_rays, size_t n_hits, int* d_hit_indices,
                                        double* d_hit_data, grace_stream stream);

/* ---- ray inputs (deterministic generators; the reference's cuRAND streams are
 *      device-specific by its own account, include/grace/cuda/kernels/gen_rays.cuh:21-24) */
/* orthographic_projection_rays specialised as orthogonal_rays_z
 * (tests/helper/rays.cuh:55-79; kernels/gen_rays.cuh:319-360,667-725). mins4/maxs4 host. */
grace_status grace_rays_orthogonal_z(int n_side, const float* h_mins4, const float* h_maxs4,
                                     void* d_rays, float* h_area, grace_stream stream);
/* pinhole_camera_rays (kernels/gen_rays.cuh:362-395,727-789), Real = float; fovy in radians. */
grace_status grace_rays_pinhole(int res_x, int res_y, const float* h_camera, const float* h_look_at,
                                const float* h_view_up, float fovy, float length, void* d_rays,
                                grace_stream stream);
/* One source, HEALPix nested pixel centres (RayVectorGeneration/src/generateRays.c:57-59). */
grace_status grace_rays_healpix(int nside, float ox, float oy, float oz, float length,
                                void* d_rays, grace_stream stream);
/* Isotropic rays from one origin, sorted by ray_dir_morton_key
 * (kernels/gen_rays.cuh:38-43,104-170 uniform_random_rays); own counter-based generator. */
grace_status grace_rays_isotropic(size_t n_rays, float ox, float oy, float oz, float length,
                                  uint64_t seed, void* d_rays, grace_stream stream);

/* uniform_random_rays_single_octant (gen_rays.cuh:62-97; kernels/gen_rays.cuh:161-204,484-517):
 * octant 0 (MMM) .. 7 (PPP), bit 2 = x, bit 1 = y, bit 0 = z, set = positive component. */
grace_status grace_rays_isotropic_octant(size_t n_rays, float ox, float oy, float oz, float length,
                                         int octant, uint64_t seed, void* d_rays,
                                         grace_stream stream);
/* one_to_many_rays (gen_rays.cuh:99-208; kernels/gen_rays.cuh:206-243,519-611): ray i goes from
 * the origin to point i (direction normalised, length = distance).  Points: elems_per_point
 * floats/doubles each, x y z first.  sort_type: 0 NoSort, 1 DirectionSort (ray_dir_morton_key),
 * 2 EndPointSort (30-bit Morton key of the end point within h_bot/h_top; the reference's
 * bounds-free overload passes AABB_bot twice, gen_rays.cuh:121-122 -- not reproduced: pass the
 * real bounds).  Anything else: GRACE_INVALID_ARGUMENT (std::invalid_argument there). */
grace_status grace_rays_one_to_many(size_t n_rays, float ox, float oy, float oz,
                                    const void* d_points, int is_double, int elems_per_point,
                                    int sort_type, const float* h_bot, const float* h_top,
                                    void* d_rays, grace_stream stream);
/* plane_parallel_random_rays (gen_rays.cuh:210-262; kernels/gen_rays.cuh:245-317,613-665): a
 * width x height grid of cells spanned by w and h from base, one ray per cell from a random
 * point of the cell, direction normalize(cross(w, h)).  base, w, h: host float[3]. */
grace_status grace_rays_plane_parallel_random(int width, int height, const float* h_base,
                                              const float* h_w, const float* h_h, float length,
                                              uint64_t seed, void* d_rays, grace_stream stream);
/* orthographic_projection_rays (gen_rays.cuh:264-329; kernels/gen_rays.cuh:319-360,667-725),
 * Real = float: ray 0 is the top-left pixel, x fastest. */
grace_status grace_rays_orthographic_projection(int res_x, int res_y, const float* h_camera,
                                                const float* h_look_at, const float* h_view_up,
                                                float vertical_extent, float length,
                                                void* d_rays, grace_stream stream);

/* ---- double4 spheres: the reference's templates with Real4 = double4, Real = double
 *      (build_sph.cuh:84-126, trace_sph.cuh:57-110).  Keys: grace_morton_keys{30,63}_points;
 *      sort: grace_sort_pairs_u32/u64 with 32-byte records. -------------------------------- */
/* euclidean_deltas_sph<double4>: DeltaEuclidean forms the squared distance in double and returns
 * it as float (generic/functors/albvh.h:44-74); deltas[n + 1] floats, +inf at both ends. */
grace_status grace_deltas_euclid_d4(const double* d_spheres, size_t n, float* d_deltas,
                                    grace_stream stream);
/* The same with a device_vector<double> of deltas (build_tree<double4> declares
 * device_vector<Real> deltas, tests/helper/tree.cuh:20-24: the functor's float widened), and
 * surface_area_deltas_sph<double4> (build_sph.cuh:97-105; generic/functors/albvh.h:84-126 with
 * AABBSphere narrowing centre -+ radius to float3, generic/functors/aabb.h:9-26). */
grace_status grace_deltas_euclid_d4_f64(const double* d_spheres, size_t n, double* d_deltas,
                                        grace_stream stream);
grace_status grace_deltas_area_d4(const double* d_spheres, size_t n, float* d_deltas,
                                  grace_stream stream);
grace_status grace_deltas_area_d4_f64(const double* d_spheres, size_t n, double* d_deltas,
                                      grace_stream stream);
/* ALBVH_sph<Real4, DeltaType> (build_sph.cuh:118-124) for the remaining delta types: 64-bit XOR
 * deltas (morton_keys63_sort_sph -> XOR_deltas_sph -> ALBVH_sph), double deltas, and double4
 * spheres with XOR deltas.  DeltaComp is thrust::less in all of them (albvh.cuh:1045-1072); a
 * caller-defined comparator functor cannot cross a C ABI. */
grace_status grace_albvh_build_f4_u64(const float* d_spheres, size_t n, const uint64_t* d_deltas,
                                      int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                      size_t* h_n_leaves, grace_stream stream);
grace_status grace_albvh_build_f4_f64(const float* d_spheres, size_t n, const double* d_deltas,
                                      int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                      size_t* h_n_leaves, grace_stream stream);
grace_status grace_albvh_build_d4_f64(const double* d_spheres, size_t n, const double* d_deltas,
                                      int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                      size_t* h_n_leaves, grace_stream stream);
grace_status grace_albvh_build_d4_u32(const double* d_spheres, size_t n, const uint32_t* d_deltas,
                                      int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                      size_t* h_n_leaves, grace_stream stream);
grace_status grace_albvh_build_d4_u64(const double* d_spheres, size_t n, const uint64_t* d_deltas,
                                      int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                      size_t* h_n_leaves, grace_stream stream);
/* ALBVH_sph<double4>: same tree builder; leaf / node boxes are AABBSphere's float3 corners of the
 * double centre -+ radius (generic/functors/aabb.h:9-26).  Same output layout as _f4. */
grace_status grace_albvh_build_d4(const double* d_spheres, size_t n, const float* d_deltas,
                                  int max_per_leaf, int* d_nodes, int* d_leaves, int* d_root,
                                  size_t* h_n_leaves, grace_stream stream);
/* trace_hitcounts_sph / trace_cumulative_sph / trace_sph pass 2 <double4, (int,) double>
 * (trace_sph.cuh:57-168): sphere_hit and the kernel integral in double (ray members are float,
 * generic/intersect.h:9-55), one running double sum per ray in ascending primitive index; per-hit
 * integrals and distances are double.  Same kernel as the float path (ray coherence order,
 * cluster tests, culling rounds) walking float records that contain the double spheres; every
 * surviving candidate is then tested against the caller's double4 record. */
grace_status grace_trace_hitcounts_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                      size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                      const int* d_leaves, const int* d_root, int* d_hit_counts,
                                      grace_stream stream);
grace_status grace_trace_cumulative_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                       size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                       const int* d_leaves, const int* d_root, double* d_sums,
                                       grace_stream stream);
grace_status grace_trace_hits_d4(const void* d_rays, size_t n_rays, const double* d_spheres,
                                 size_t n_spheres, const int* d_nodes, size_t n_nodes,
                                 const int* d_leaves, const int* d_root, const int* d_ray_offsets,
                                 int* d_hit_indices, double* d_hit_integrals,
                                 double* d_hit_distances, grace_stream stream);
/* Same status word as grace_trace_status (kept for callers of the round-1 interface). */
grace_status grace_trace_status_d4(grace_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* GRACE_HIP_H */
