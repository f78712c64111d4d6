// grace/cuda/trace_sph.cuh -- the SPH trace API of the reference
// (include/grace/cuda/trace_sph.cuh:22-241) with its template signatures over
// thrust::device_vector; every body dispatches on Real4 / Real to the traversal kernel of
// libgrace_hip.so (csrc/trace.hip), which stands in for trace_texref<RayData>(... functors ...)
// with the functors the reference passes here:
//
//   trace_hitcounts_sph      Intersect_sphere_bool    + OnHit_increment          -> grace_trace_hitcounts_*
//   trace_cumulative_sph     Intersect_sphere_b2dist  + OnHit_sphere_cumulate    -> grace_trace_cumulative_*
//   trace_sph                hit counts, exclusive scan, resize, then
//                            RayEntry_from_array + OnHit_sphere_individual       -> grace_trace_hits_*
//   trace_with_sentinels_sph the same with one sentinel slot per ray
//
// (Real4, Real) is (float4, float) or (double4, double); IndexType is a 32-bit integer.  Per-ray
// results equal the brute-force loop over all spheres (the reference's own criterion,
// tests/tree_traversal); column densities are the class-ordered fp32 sum documented in
// grace_hip.h (within 1e-6 of the reference's single running sum).  As in the reference the
// number of rays must be a multiple of 32 (bintree_trace.cuh:231-238: std::invalid_argument).
#pragma once

#include "grace/cuda/nodes.h"
#include "grace/detail/raw.h"
#include "grace/ray.h"

#include <limits>

namespace grace {

// include/grace/cuda/trace_sph.cuh:22-50: the normalised cubic-spline column kernel,
// F(b / h) at 51 equidistant impact parameters; libgrace_hip.so holds the same table.
const int N_table = 51;

template <typename Real>
struct KernelIntegrals
{
    const static Real table[N_table];
};

template <typename Real>
const Real KernelIntegrals<Real>::table[N_table] = {
    Real(1.90986019771937), Real(1.90563449910964), Real(1.89304415940934), Real(1.87230928086763),
    Real(1.84374947679902), Real(1.80776276033034), Real(1.76481079856299), Real(1.71540816859939),
    Real(1.66011373131439), Real(1.59952322363667), Real(1.53426266082279), Real(1.46498233888091),
    Real(1.39235130929287), Real(1.31705223652377), Real(1.23977618317103), Real(1.16121278415369),
    Real(1.08201943664419), Real(1.00288866679720), Real(0.924475767210246), Real(0.847415371038733),
    Real(0.772316688105931), Real(0.699736940377312), Real(0.630211918937167), Real(0.564194562399538),
    Real(0.502076205853037), Real(0.444144023534733), Real(0.390518196140658), Real(0.341148855945766),
    Real(0.295941946237307), Real(0.254782896476983), Real(0.217538645099225), Real(0.184059547649710),
    Real(0.154181189781890), Real(0.127726122453554), Real(0.104505535066266),
    Real(8.432088120445191E-002), Real(6.696547102921641E-002), Real(5.222604427168923E-002),
    Real(3.988433820097490E-002), Real(2.971866601747601E-002), Real(2.150552303075515E-002),
    Real(1.502124104014533E-002), Real(1.004371608622562E-002), Real(6.354242122978656E-003),
    Real(3.739494884706115E-003), Real(1.993729589156428E-003), Real(9.212900163813992E-004),
    Real(3.395908945333921E-004), Real(8.287326418242995E-005), Real(7.387919939044624E-006),
    Real(0.000000000000000E+000)
};

namespace detail {

inline void check_ray_count(size_t n_rays)
{
    // bintree_trace.cuh:231-238
    if (n_rays % 32 != 0)
        throw std::invalid_argument("Number of rays must be a multiple of the warp size (32).");
}

struct TreeArgs { const int* nodes; size_t n_nodes; const int* leaves; const int* root; };
inline TreeArgs tree_args(const Tree& t)
{
    TreeArgs a = { reinterpret_cast<const int*>(raw(t.nodes)), t.leaves.size() - 1,
                   reinterpret_cast<const int*>(raw(t.leaves)), t.root_index_ptr };
    return a;
}

inline void hitcounts_dispatch(const Ray* r, size_t nr, const float4* s, size_t n, const TreeArgs& t, int* out)
{ GRACE_STATUS_CHECK(grace_trace_hitcounts_f4(r, nr, reinterpret_cast<const float*>(s), n, t.nodes, t.n_nodes, t.leaves, t.root, out, NULL)); }
inline void hitcounts_dispatch(const Ray* r, size_t nr, const double4* s, size_t n, const TreeArgs& t, int* out)
{ GRACE_STATUS_CHECK(grace_trace_hitcounts_d4(r, nr, reinterpret_cast<const double*>(s), n, t.nodes, t.n_nodes, t.leaves, t.root, out, NULL)); }

// The hit-count pass of trace_sph: the library keeps what the per-hit pass can reuse.
inline void hitcounts_keep_dispatch(const Ray* r, size_t nr, const float4* s, size_t n, const TreeArgs& t, int* out)
{ GRACE_STATUS_CHECK(grace_trace_hitcounts_keep_f4(r, nr, reinterpret_cast<const float*>(s), n, t.nodes, t.n_nodes, t.leaves, t.root, out, NULL)); }
inline void hitcounts_keep_dispatch(const Ray* r, size_t nr, const double4* s, size_t n, const TreeArgs& t, int* out)
{ hitcounts_dispatch(r, nr, s, n, t, out); }

inline void cumulative_dispatch(const Ray* r, size_t nr, const float4* s, size_t n, const TreeArgs& t, float* out)
{ GRACE_STATUS_CHECK(grace_trace_cumulative_f4(r, nr, reinterpret_cast<const float*>(s), n, t.nodes, t.n_nodes, t.leaves, t.root, out, NULL)); }
inline void cumulative_dispatch(const Ray* r, size_t nr, const double4* s, size_t n, const TreeArgs& t, double* out)
{ GRACE_STATUS_CHECK(grace_trace_cumulative_d4(r, nr, reinterpret_cast<const double*>(s), n, t.nodes, t.n_nodes, t.leaves, t.root, out, NULL)); }

inline void hits_dispatch(const Ray* r, size_t nr, const float4* s, size_t n, const TreeArgs& t,
                          const int* off, int* idx, float* integrals, float* dists)
{ GRACE_STATUS_CHECK(grace_trace_hits_f4(r, nr, reinterpret_cast<const float*>(s), n, t.nodes, t.n_nodes, t.leaves, t.root, off, idx, integrals, dists, NULL)); }
inline void hits_dispatch(const Ray* r, size_t nr, const double4* s, size_t n, const TreeArgs& t,
                          const int* off, int* idx, double* integrals, double* dists)
{ GRACE_STATUS_CHECK(grace_trace_hits_d4(r, nr, reinterpret_cast<const double*>(s), n, t.nodes, t.n_nodes, t.leaves, t.root, off, idx, integrals, dists, NULL)); }

// The traversal's status word: the reference asserts on stack exhaustion in GRACE_DEBUG builds
// (bintree_trace.cuh:164); here it is an error in every build.
inline void check_trace_status() { GRACE_STATUS_CHECK(grace_trace_status(NULL)); }

// Hit counts -> exclusive offsets; returns the total (trace_sph.cuh:126-137), refusing totals
// that int offsets cannot address.
inline size_t counts_to_offsets(thrust::device_vector<int>& d_ray_offsets, size_t extra)
{
    long long total = 0;
    GRACE_STATUS_CHECK(grace_scan_exclusive_i32(raw(d_ray_offsets), d_ray_offsets.size(),
                                                raw(d_ray_offsets), &total, NULL));
    if (total + (long long)extra > (long long)std::numeric_limits<int>::max())
        throw std::invalid_argument("trace_sph: more than INT_MAX hits; the int ray offsets cannot "
                                    "address the per-hit arrays. Trace fewer rays per call.");
    return size_t(total);
}

template <typename T>
inline void fill_bits(thrust::device_vector<T>& v, T value)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit elements");
    if (sizeof(T) == 4) {
        uint32_t bits;
        __builtin_memcpy(&bits, &value, 4);
        GRACE_STATUS_CHECK(grace_fill_u32(raw(v), v.size(), bits, NULL));
    } else {
        // 64-bit sentinels (double): the container's own fill (container behaviour, not an
        // algorithm on the hot path).
        v.assign(v.size(), value);
    }
}

} // namespace detail

template <typename Real4>
GRACE_HOST void trace_hitcounts_sph(
    const thrust::device_vector<Ray>& d_rays,
    const thrust::device_vector<Real4>& d_spheres,
    const Tree& d_tree,
    thrust::device_vector<int>& d_hit_counts)
{
    detail::check_ray_count(d_rays.size());
    detail::hitcounts_dispatch(detail::raw(d_rays), d_rays.size(), detail::raw(d_spheres),
                               d_spheres.size(), detail::tree_args(d_tree), detail::raw(d_hit_counts));
    detail::check_trace_status();
}

template <typename Real4, typename Real>
GRACE_HOST void trace_cumulative_sph(
    const thrust::device_vector<Ray>& d_rays,
    const thrust::device_vector<Real4>& d_spheres,
    const Tree& d_tree,
    thrust::device_vector<Real>& d_cumulated)
{
    detail::check_ray_count(d_rays.size());
    detail::cumulative_dispatch(detail::raw(d_rays), d_rays.size(), detail::raw(d_spheres),
                                d_spheres.size(), detail::tree_args(d_tree), detail::raw(d_cumulated));
    detail::check_trace_status();
}

template <typename Real4, typename IndexType, typename Real>
GRACE_HOST void trace_sph(
    const thrust::device_vector<Ray>& d_rays,
    const thrust::device_vector<Real4>& d_spheres,
    const Tree& d_tree,
    // The segmented scans and sorts require ray offsets to be int.
    thrust::device_vector<int>& d_ray_offsets,
    thrust::device_vector<IndexType>& d_hit_indices,
    thrust::device_vector<Real>& d_hit_integrals,
    thrust::device_vector<Real>& d_hit_distances)
{
    static_assert(sizeof(IndexType) == sizeof(int), "IndexType must be a 32-bit integer");
    // Initially, d_ray_offsets is actually per-ray *hit counts*.
    detail::check_ray_count(d_rays.size());
    detail::hitcounts_keep_dispatch(detail::raw(d_rays), d_rays.size(), detail::raw(d_spheres),
                                    d_spheres.size(), detail::tree_args(d_tree), detail::raw(d_ray_offsets));
    const size_t total_hits = detail::counts_to_offsets(d_ray_offsets, 0);

    d_hit_integrals.resize(total_hits);
    d_hit_indices.resize(total_hits);
    d_hit_distances.resize(total_hits);
    if (total_hits == 0) return;

    detail::hits_dispatch(detail::raw(d_rays), d_rays.size(), detail::raw(d_spheres),
                          d_spheres.size(), detail::tree_args(d_tree), detail::raw(d_ray_offsets),
                          reinterpret_cast<int*>(detail::raw(d_hit_indices)),
                          detail::raw(d_hit_integrals), detail::raw(d_hit_distances));
    detail::check_trace_status();
}

template <typename Real4, typename IndexType, typename Real>
GRACE_HOST void trace_with_sentinels_sph(
    const thrust::device_vector<Ray>& d_rays,
    const thrust::device_vector<Real4>& d_spheres,
    const Tree& d_tree,
    thrust::device_vector<int>& d_ray_offsets,
    thrust::device_vector<IndexType>& d_hit_indices,
    const int index_sentinel,
    thrust::device_vector<Real>& d_hit_integrals,
    const Real integral_sentinel,
    thrust::device_vector<Real>& d_hit_distances,
    const Real distance_sentinel)
{
    static_assert(sizeof(IndexType) == sizeof(int), "IndexType must be a 32-bit integer");
    const size_t n_rays = d_rays.size();
    detail::check_ray_count(n_rays);
    detail::hitcounts_keep_dispatch(detail::raw(d_rays), n_rays, detail::raw(d_spheres),
                                    d_spheres.size(), detail::tree_args(d_tree), detail::raw(d_ray_offsets));
    // Each ray segment in the output arrays ends with a sentinel value marking the end of the
    // ray; increase offsets accordingly (trace_sph.cuh:199-208).
    const size_t allocate_size = detail::counts_to_offsets(d_ray_offsets, n_rays) + n_rays;
    GRACE_STATUS_CHECK(grace_add_iota_i32(detail::raw(d_ray_offsets), n_rays, NULL));

    // Outputs start out as their sentinel values: these slots are not touched by the trace.
    d_hit_indices.resize(allocate_size);
    d_hit_integrals.resize(allocate_size);
    d_hit_distances.resize(allocate_size);
    detail::fill_bits(d_hit_indices, IndexType(index_sentinel));
    detail::fill_bits(d_hit_integrals, integral_sentinel);
    detail::fill_bits(d_hit_distances, distance_sentinel);

    detail::hits_dispatch(detail::raw(d_rays), n_rays, detail::raw(d_spheres), d_spheres.size(),
                          detail::tree_args(d_tree), detail::raw(d_ray_offsets),
                          reinterpret_cast<int*>(detail::raw(d_hit_indices)),
                          detail::raw(d_hit_integrals), detail::raw(d_hit_distances));
    detail::check_trace_status();
}

// ---- extensions (not in the reference) ------------------------------------------------------
// What every trace call derives from its arguments alone -- the scene's pre-pass records, the ray
// coherence order -- is cached by the library for arrays that are traced repeatedly (from the second
// consecutive call on; see "Cached trace records" in grace_hip.h).  prepare_trace_sph /
// prepare_trace_rays fill that cache NOW and pin it for as long as the returned handle lives.
// Cached records are validated against the arrays' current contents before every use, so modifying
// or reallocating d_spheres / d_tree / d_rays while a handle is alive is safe (it costs a
// re-derivation); only grace_trace_set_cache_validation(0) turns that into the caller's promise.
// Results never depend on any of this.
class PreparedTrace
{
public:
    PreparedTrace() : scene_(false), rays_(false) {}
    PreparedTrace(PreparedTrace&& o) : scene_(o.scene_), rays_(o.rays_) { o.scene_ = o.rays_ = false; }
    PreparedTrace& operator=(PreparedTrace&& o)
    {
        if (this != &o) { release(); scene_ = o.scene_; rays_ = o.rays_; o.scene_ = o.rays_ = false; }
        return *this;
    }
    ~PreparedTrace() { release(); }
    // Unpins and frees what this handle pinned (a later prepare_* may already have replaced it).
    void release()
    {
        if (scene_) GRACE_STATUS_CHECK(grace_trace_release());
        if (rays_) GRACE_STATUS_CHECK(grace_trace_release_rays());
        scene_ = rays_ = false;
    }

private:
    PreparedTrace(const PreparedTrace&);
    PreparedTrace& operator=(const PreparedTrace&);
    bool scene_, rays_;
    friend PreparedTrace prepare_trace_sph(const thrust::device_vector<float4>&, const Tree&);
    friend PreparedTrace prepare_trace_rays(const thrust::device_vector<Ray>&);
};

__attribute__((warn_unused_result))
GRACE_HOST PreparedTrace prepare_trace_sph(const thrust::device_vector<float4>& d_spheres, const Tree& d_tree)
{
    const detail::TreeArgs t = detail::tree_args(d_tree);
    GRACE_STATUS_CHECK(grace_trace_prepare_f4(reinterpret_cast<const float*>(detail::raw(d_spheres)),
                                              d_spheres.size(), t.nodes, t.n_nodes, t.leaves, NULL));
    PreparedTrace h;
    h.scene_ = true;
    return h;
}

__attribute__((warn_unused_result))
GRACE_HOST PreparedTrace prepare_trace_rays(const thrust::device_vector<Ray>& d_rays)
{
    GRACE_STATUS_CHECK(grace_trace_prepare_rays(detail::raw(d_rays), d_rays.size(), NULL));
    PreparedTrace h;
    h.rays_ = true;
    return h;
}

// Drops whatever the calling thread's context has cached or pinned.
GRACE_HOST void release_prepared_trace()
{
    GRACE_STATUS_CHECK(grace_trace_release());
    GRACE_STATUS_CHECK(grace_trace_release_rays());
}

} // namespace grace
